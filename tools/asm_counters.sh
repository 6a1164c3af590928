#!/bin/bash
# SQ counters of the assembly kernel (three launches of the 10M-DOF assembly): where its wave cycles go.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
TAG=${TAG:-r01e}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $O/p_c1 -o c -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asm_sq1.log 2>&1
python3 tools/rocprof_summary.py counters $(find $O/p_c1 -name '*_results.db' | head -1) $O/${TAG}_asm_sq_cycles.json assemble
rm -rf $O/p_c1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES -d $O/p_c2 -o c -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asm_sq2.log 2>&1
python3 tools/rocprof_summary.py counters $(find $O/p_c2 -name '*_results.db' | head -1) $O/${TAG}_asm_sq_lds.json assemble
rm -rf $O/p_c2
