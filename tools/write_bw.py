#!/usr/bin/env python3
"""Write bandwidth of this GPU as the runtime's own fill kernel sees it (hipMemsetAsync of the assembly's 2.16 GB of
matrix values): the practical ceiling of a store-only kernel, next to which the assembly's store skeleton is read."""
import ctypes
import sys

hip = ctypes.CDLL("libamdhip64.so")
nbytes = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 269586136 * 8


def chk(rc):
    if rc:
        raise RuntimeError(f"hip error {rc}")


p = ctypes.c_void_p()
chk(hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)))
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
chk(hip.hipEventCreate(ctypes.byref(e0)))
chk(hip.hipEventCreate(ctypes.byref(e1)))
for rep in range(6):
    chk(hip.hipEventRecord(e0, None))
    chk(hip.hipMemsetAsync(p, 0, ctypes.c_size_t(nbytes), None))
    chk(hip.hipEventRecord(e1, None))
    chk(hip.hipEventSynchronize(e1))
    ms = ctypes.c_float()
    chk(hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1))
    print(f"memset {nbytes / 1e9:.2f} GB: {ms.value:.3f} ms -> {nbytes / ms.value / 1e9:.2f} TB/s")
chk(hip.hipFree(p))
