"""Wall-clock tic/toc (mirrors src/common/timer.py:3-19)."""
import time


class Timer:
    def __init__(self):
        self._t0 = None
        self.elapsed = 0.0

    def tic(self):
        self._t0 = time.perf_counter()

    def toc(self):
        self.elapsed = time.perf_counter() - self._t0
        return self.elapsed
