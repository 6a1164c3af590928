#!/usr/bin/env python3
"""Turn rocprofv3 result databases (rocpd sqlite, the default output of this ROCm) into the small text files
committed under profiles/.
  rocprof_summary.py stats  <results.db> <out.csv> [out.md]     --kernel-trace --stats run
  rocprof_summary.py pmc    <fetch.db> <write.db> <out.json> [per]   --pmc FETCH_SIZE / --pmc WRITE_SIZE runs (per: launches per logical launch)
  rocprof_summary.py counters <results.db> <out.json> [kernel-substring]   any --pmc run: per-kernel mean of every counter
  rocprof_summary.py timeline <results.db> <out.txt> [n]        last n dispatches of a --kernel-trace run: start, duration, gap"""
import json
import re
import sqlite3
import sys


def short(name, n=110):
    return name if len(name) <= n else name[:n] + "..."


def stats(db, out_csv, out_md=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                       "order by total_duration desc").fetchall()
    with open(out_csv, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for name, calls, tot, avg, pct in rows:
            f.write('"%s",%d,%.0f,%.1f,%.4f\n' % (short(name, 200).replace('"', "'"), calls, tot * 1e3, avg * 1e3, pct))
    if out_md:
        with open(out_md, "w") as f:
            f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
            for name, calls, tot, avg, pct in rows[:18]:
                f.write("| `%s` | %d | %.2f | %.1f | %.2f |\n" % (short(name, 90), calls, tot / 1e3, avg, pct))


def pmc(fetch_db, write_db, out_json, per=1):
    """per: launches that make up ONE logical launch (the row-run kernels run once per class of x-lines: 2 launches per matrix in 2-D,
    4 in 3-D) -- `hbm_bytes_per_launch` is then the sum over such a group"""
    per = int(per)
    res = {}
    for db, counter in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
        con = sqlite3.connect(db)
        q = ("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
             "group by kernel_name")
        for name, n, total in con.execute(q, (counter,)):
            key = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")
            key = re.split(r"[<(]", key)[0].strip()
            res.setdefault(key, {"full_name": short(name, 160)})[counter + "_KB"] = total / n
            res[key]["launches"] = n
    for k, v in res.items():
        if "FETCH_SIZE_KB" in v and "WRITE_SIZE_KB" in v:
            # gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section) -> x2
            v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0 * per
            if per > 1:
                v["launches_per_logical_launch"] = per
    with open(out_json, "w") as f:
        json.dump({"correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch (gfx950 FETCH_SIZE counts 64 B per "
                                 "128-B request; separate --pmc passes)",
                   "source_hash": _source_hash(),      # the kernel sources these counters belong to (bench.py checks it)
                   "kernels": res}, f, indent=1)


def _source_hash():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from pynama_amd import _lib
        return _lib.source_hash()
    except Exception:      # summarising on a machine without the library: bench.py will not quote these counters
        return "unknown"


def timeline(db, out_txt, n=60):
    """Back-to-back picture of the last `n` kernel dispatches (us): where an iteration's time goes between launches."""
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start desc limit ?", (int(n),)).fetchall()[::-1]
    with open(out_txt, "w") as f:
        f.write("start_us  dur_us  gap_before_us  kernel\n")
        t0, prev_end = rows[0][1], rows[0][1]
        for name, s, e in rows:
            f.write("%9.1f %7.1f %7.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(name, 70)))
            prev_end = e


def counters(db, out_json, needle=""):
    con = sqlite3.connect(db)
    res = {}
    q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
    for name, cn, n, mean in con.execute(q):
        if needle and needle not in name:
            continue
        key = re.split(r"[<(]", re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", ""))[0].strip()
        res.setdefault(key, {"launches": n})[cn] = mean
    with open(out_json, "w") as f:
        json.dump({"source_hash": _source_hash(), **res}, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(*sys.argv[2:5])
    elif sys.argv[1] == "timeline":
        timeline(*sys.argv[2:5])
    elif sys.argv[1] == "stats":
        stats(*sys.argv[2:5])
    else:
        pmc(*sys.argv[2:6])
