#!/bin/bash
# assembly time of the plan-free lattice kernel for every tile shape, and of the patch-plan kernel
# usage (GPU box): bash tools/lattice_sweep.sh [extra bench args]
for t in ${TILES:-0 1 2 3 4 5 6 7 8 9}; do
  PYNAMA_LATTICE_TILE=$t timeout -k 10 120 python bench.py --no-cpu-baseline --no-check --steps 5 --warmup 2 "$@" > gpurun_out/lat_tile$t.json 2> gpurun_out/lat_tile$t.err || exit 1
done
timeout -k 10 120 python bench.py --variant 2 --no-cpu-baseline --no-check --steps 5 --warmup 2 "$@" > gpurun_out/lat_plan.json 2> /dev/null || exit 1
for f in gpurun_out/lat_tile*.json gpurun_out/lat_plan.json; do
  python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], 'assembly ms %.3f' % d['breakdown_ms']['assembly'], 'G elem-DOF/s %.1f' % (d['value']/1e9), 'frac %.3f' % d['roofline_assembly']['frac'])" $f
done
