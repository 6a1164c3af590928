#!/bin/bash
# Round-3 profile pass on the GPU box (the r02 pass of profile_pass.sh + the second-order legs, the tetrahedra kernel's counters).
# Summaries land in gpurun_out/${TAG}_* (the rocpd databases are deleted: gpurun merges at most 64 MiB back).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
TAG=${TAG:-r03}
S="python3 tools/rocprof_summary.py"
db() { find $1 -name '*_results.db' | head -1; }
stats() {   # stats <tag> <command...>
  local t=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/p_s -o s -- "$@" > $O/${TAG}_${t}.log 2>&1
  $S stats $(db $O/p_s) $O/${TAG}_${t}_kernel_stats.csv $O/${TAG}_${t}_summary.md
  rm -rf $O/p_s
}
traffic() { # traffic <tag> <per> <command...>
  local t=$1 per=$2; shift; shift
  rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- "$@" > $O/${TAG}_${t}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- "$@" > $O/${TAG}_${t}_write.log 2>&1
  $S pmc $(db $O/p_f) $(db $O/p_w) $O/${TAG}_pmc_${t}.json $per
  rm -rf $O/p_f $O/p_w
}
sq() {      # sq <tag> <kernel substring> <command...>
  local t=$1 k=$2; shift; shift
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $O/p_c -o c -- "$@" > $O/${TAG}_${t}_sq.log 2>&1
  $S counters $(db $O/p_c) $O/${TAG}_${t}_sq_cycles.json $k
  rm -rf $O/p_c
}
PART=${PART:-all}     # gpurun calls are limited to 20 minutes: PART=a (bench, headline traffic), b (second-order legs), c (C3, C5)
if [ "$PART" = all ] || [ "$PART" = a ]; then
# headline alone, then the whole default run
rocprofv3 --kernel-trace --stats -d $O/p_s -o s -- python3 bench.py --no-cpu-baseline --no-extra > $O/${TAG}_bench_10Mdof.json 2> $O/${TAG}_bench.err
$S stats $(db $O/p_s) $O/${TAG}_bench_10Mdof_kernel_stats.csv $O/${TAG}_bench_10Mdof_summary.md; rm -rf $O/p_s
rocprofv3 --kernel-trace --stats -d $O/p_s -o s -- python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_all_configs.json 2> $O/${TAG}_bench_all.err
$S stats $(db $O/p_s) $O/${TAG}_bench_all_configs_kernel_stats.csv $O/${TAG}_bench_all_configs_summary.md; rm -rf $O/p_s
echo "bench stats done"
traffic traffic 1 python3 tools/prof_case.py cg 215 3
traffic assembly 1 python3 tools/prof_case.py asm 215 3
PYNAMA_JITTER=0.2 traffic assembly_general 1 python3 tools/prof_case.py asm 215 3
echo "headline traffic done"
fi
if [ "$PART" = all ] || [ "$PART" = b ]; then
# second-order legs: kernel statistics of the whole case, traffic per kernel family (one matrix shape per process)
stats ho3_2d python3 tools/ho3_case.py 2 1024 3
stats ho3_3d python3 tools/ho3_case.py 3 64 3
traffic ho3_2d_K 2 python3 tools/prof_case.py ho3k 2 1024 3
traffic ho3_2d_Rw 2 python3 tools/prof_case.py ho3rw 2 1024 3
traffic ho3_2d_cg 1 python3 tools/prof_case.py ho3cg 2 1024 3
traffic ho3_3d_K 4 python3 tools/prof_case.py ho3k 3 64 3
traffic ho3_3d_Rw 4 python3 tools/prof_case.py ho3rw 3 64 3
traffic ho3_3d_cg 1 python3 tools/prof_case.py ho3cg 3 64 3
sq ho3_3d_K assemble_ho3 python3 tools/prof_case.py ho3k 3 64 3
sq ho3_2d_K assemble_ho3 python3 tools/prof_case.py ho3k 2 1024 3
echo "ho3 done"
fi
if [ "$PART" = all ] || [ "$PART" = c ]; then
# C3 (128^3 KLE, compact Krhs)
stats kle128 python3 tools/prof_case.py kle 128 3
traffic kle128 1 python3 tools/prof_case.py kle 128 3
PYNAMA_JITTER=0.2 stats kle128_general python3 tools/prof_case.py kle 128 3
echo "kle done"
# C5: 5 M tetrahedra -- kernel statistics, traffic and SQ counters of the patch kernel (what bounds it)
stats tet5M python3 tools/tet_case.py
traffic tet5M 1 python3 tools/tet_case.py
sq tet5M assemble_p1_tet python3 tools/tet_case.py
fi
echo "all done"
