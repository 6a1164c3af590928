#!/bin/bash
# One profile pass on the GPU box: kernel statistics of bench.py, PMC HBM traffic of the CG / KLE kernels.
# Summaries land in gpurun_out/${TAG}_* (the rocpd databases are deleted: gpurun merges at most 64 MiB back).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
TAG=${TAG:-r01e}      # profile pass name (files land in gpurun_out/${TAG}_*, copied to profiles/)
rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_10Mdof.json 2> $O/${TAG}_bench.err
python3 tools/rocprof_summary.py stats $(find $O/p_bench -name '*_results.db' | head -1) $O/${TAG}_bench_10Mdof_kernel_stats.csv $O/${TAG}_bench_10Mdof_summary.md
rm -rf $O/p_bench
rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- python3 tools/prof_case.py cg 215 3 > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- python3 tools/prof_case.py cg 215 3 > $O/${TAG}_pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $(find $O/p_f -name '*_results.db' | head -1) $(find $O/p_w -name '*_results.db' | head -1) $O/${TAG}_pmc_traffic.json
rm -rf $O/p_f $O/p_w
if [ "${WITH_KLE:-1}" = "1" ]; then
rocprofv3 --kernel-trace --stats -d $O/p_k -o k -- python3 tools/prof_case.py kle 128 3 > $O/${TAG}_kle.log 2>&1
python3 tools/rocprof_summary.py stats $(find $O/p_k -name '*_results.db' | head -1) $O/${TAG}_kle128_kernel_stats.csv $O/${TAG}_kle128_summary.md
rm -rf $O/p_k
rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- python3 tools/prof_case.py kle 128 3 > $O/${TAG}_kle_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- python3 tools/prof_case.py kle 128 3 > $O/${TAG}_kle_pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $(find $O/p_f -name '*_results.db' | head -1) $(find $O/p_w -name '*_results.db' | head -1) $O/${TAG}_kle128_pmc_traffic.json
rm -rf $O/p_f $O/p_w
fi
# assembly kernel traffic (three launches of the assembly alone)
rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asm_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asm_pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $(find $O/p_f -name '*_results.db' | head -1) $(find $O/p_w -name '*_results.db' | head -1) $O/${TAG}_pmc_assembly.json
rm -rf $O/p_f $O/p_w
