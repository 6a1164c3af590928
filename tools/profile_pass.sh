#!/bin/bash
# One profile pass on the GPU box: kernel statistics of bench.py, PMC HBM traffic of the CG / KLE kernels.
# Summaries land in gpurun_out/r01d_* (the rocpd databases are deleted: gpurun merges at most 64 MiB back).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 bench.py --no-cpu-baseline > $O/r01d_bench_10Mdof.json 2> $O/r01d_bench.err
python3 tools/rocprof_summary.py stats $(find $O/p_bench -name '*_results.db' | head -1) $O/r01d_bench_10Mdof_kernel_stats.csv $O/r01d_bench_10Mdof_summary.md
rm -rf $O/p_bench
rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- python3 tools/prof_case.py cg 215 3 > $O/r01d_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- python3 tools/prof_case.py cg 215 3 > $O/r01d_pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $(find $O/p_f -name '*_results.db' | head -1) $(find $O/p_w -name '*_results.db' | head -1) $O/r01d_pmc_traffic.json
rm -rf $O/p_f $O/p_w
rocprofv3 --kernel-trace --stats -d $O/p_k -o k -- python3 tools/prof_case.py kle 128 3 > $O/r01d_kle.log 2>&1
python3 tools/rocprof_summary.py stats $(find $O/p_k -name '*_results.db' | head -1) $O/r01d_kle128_kernel_stats.csv $O/r01d_kle128_summary.md
rm -rf $O/p_k
rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- python3 tools/prof_case.py kle 128 3 > $O/r01d_kle_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- python3 tools/prof_case.py kle 128 3 > $O/r01d_kle_pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $(find $O/p_f -name '*_results.db' | head -1) $(find $O/p_w -name '*_results.db' | head -1) $O/r01d_kle128_pmc_traffic.json
rm -rf $O/p_f $O/p_w
# assembly kernel traffic (three launches of the assembly alone)
rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- python3 tools/prof_case.py asm 215 3 > $O/r01d_asm_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- python3 tools/prof_case.py asm 215 3 > $O/r01d_asm_pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $(find $O/p_f -name '*_results.db' | head -1) $(find $O/p_w -name '*_results.db' | head -1) $O/r01d_pmc_assembly.json
rm -rf $O/p_f $O/p_w
