#!/bin/bash
# Per-kernel device times of the 128^3 KLE assembly (K+Krhs kernel, Rw kernel), parallelepipeds and general geometry.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for J in 0 0.2; do
  PYNAMA_JITTER=$J rocprofv3 --kernel-trace --stats -d $O/p_k -o k -- python3 tools/prof_case.py kle 128 3 > $O/kle_j$J.log 2>&1
  python3 tools/rocprof_summary.py stats $(find $O/p_k -name '*_results.db' | head -1) $O/kle_j$J.csv $O/kle_j$J.md
  rm -rf $O/p_k
  echo "jitter $J"; grep "assemble_q1_hex_kle" $O/kle_j$J.md
done
