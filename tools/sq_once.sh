set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $O/p_c -o c -- python3 tools/prof_case.py ho3k 3 64 3 > $O/sq_ho3.log 2>&1
python3 tools/rocprof_summary.py counters $(find $O/p_c -name '*_results.db' | head -1) $O/sq_ho3_3d_K.json assemble_ho3
rm -rf $O/p_c
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM -d $O/p_c -o c -- python3 tools/prof_case.py ho3k 3 64 3 > $O/sq_ho3b.log 2>&1 || true
python3 tools/rocprof_summary.py counters $(find $O/p_c -name '*_results.db' | head -1) $O/sq_ho3_3d_K_insts.json assemble_ho3 || true
rm -rf $O/p_c
cat $O/sq_ho3_3d_K.json $O/sq_ho3_3d_K_insts.json
