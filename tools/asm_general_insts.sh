#!/bin/bash
# Instruction counts of the general-geometry assembly kernel (jittered 215^3): how much of its VALU work is FP64 arithmetic.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
TAG=${TAG:-r02}
PYNAMA_JITTER=0.2 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM -d $O/p_c2 -o c -- python3 tools/prof_case.py asm 215 3 > $O/${TAG}_asmg_sq2.log 2>&1
python3 tools/rocprof_summary.py counters $(find $O/p_c2 -name '*_results.db' | head -1) $O/${TAG}_asm_general_sq_insts.json assemble
rm -rf $O/p_c2
