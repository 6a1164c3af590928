#!/bin/bash
# Kernel statistics of the C5 case (tools/tet_case.py: 5 M tetrahedra, GMRES variants) on one GPU.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
TAG=${TAG:-r01e}
rocprofv3 --kernel-trace --stats -d $O/p_tet -o t -- python3 tools/tet_case.py > $O/${TAG}_tet_case.log 2>&1
DB=$(find $O/p_tet -name '*_results.db' | head -1)
python3 tools/rocprof_summary.py stats $DB $O/${TAG}_tet5M_kernel_stats.csv $O/${TAG}_tet5M_summary.md
rm -rf $O/p_tet
