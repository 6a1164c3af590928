#!/bin/bash
# One profile pass on the GPU box: kernel statistics of bench.py, PMC HBM traffic of the CG / KLE kernels.
# Summaries land in gpurun_out/${TAG}_* (the rocpd databases are deleted: gpurun merges at most 64 MiB back).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
TAG=${TAG:-r02}       # profile pass name (files land in gpurun_out/${TAG}_*, copied to profiles/)
# the headline workload alone (so that a kernel's average duration is the headline's), then the whole default run with the other configurations
rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 bench.py --no-cpu-baseline --no-extra > $O/${TAG}_bench_10Mdof.json 2> $O/${TAG}_bench.err
python3 tools/rocprof_summary.py stats $(find $O/p_bench -name '*_results.db' | head -1) $O/${TAG}_bench_10Mdof_kernel_stats.csv $O/${TAG}_bench_10Mdof_summary.md
rm -rf $O/p_bench
rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_all_configs.json 2> $O/${TAG}_bench_all.err
python3 tools/rocprof_summary.py stats $(find $O/p_bench -name '*_results.db' | head -1) $O/${TAG}_bench_all_configs_kernel_stats.csv $O/${TAG}_bench_all_configs_summary.md
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
# general geometry (jittered mesh): the z-marching kernel -- traffic, kernel statistics, where its wave cycles go
PYNAMA_JITTER=0.2 rocprofv3 --pmc FETCH_SIZE -d $O/p_f -o f -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asmg_pmc_fetch.log 2>&1
PYNAMA_JITTER=0.2 rocprofv3 --pmc WRITE_SIZE -d $O/p_w -o w -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asmg_pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $(find $O/p_f -name '*_results.db' | head -1) $(find $O/p_w -name '*_results.db' | head -1) $O/${TAG}_pmc_assembly_general.json
rm -rf $O/p_f $O/p_w
PYNAMA_JITTER=0.2 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $O/p_c1 -o c -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asmg_sq1.log 2>&1
python3 tools/rocprof_summary.py counters $(find $O/p_c1 -name '*_results.db' | head -1) $O/${TAG}_asm_general_sq_cycles.json assemble
rm -rf $O/p_c1
if [ "${WITH_KLE:-1}" = "1" ]; then
PYNAMA_JITTER=0.2 rocprofv3 --kernel-trace --stats -d $O/p_k -o k -- python3 tools/prof_case.py kle 128 3 > $O/${TAG}_kle_general.log 2>&1
python3 tools/rocprof_summary.py stats $(find $O/p_k -name '*_results.db' | head -1) $O/${TAG}_kle128_general_kernel_stats.csv $O/${TAG}_kle128_general_summary.md
rm -rf $O/p_k
fi
# C5: 5 M tetrahedra, GMRES(30): kernel statistics of the final tree
rocprofv3 --kernel-trace --stats -d $O/p_tet -o t -- python3 tools/tet_case.py > $O/${TAG}_tet_case.log 2>&1
python3 tools/rocprof_summary.py stats $(find $O/p_tet -name '*_results.db' | head -1) $O/${TAG}_tet5M_kernel_stats.csv $O/${TAG}_tet5M_summary.md
rm -rf $O/p_tet
