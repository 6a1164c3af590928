#!/bin/bash
# Kernel timeline of a rank's 1/N share (tools/slab_case.py) on one GPU: where the per-iteration time goes at N ranks.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
N=${1:-8}
rocprofv3 --kernel-trace --stats -d $O/p_slab -o s -- python3 tools/slab_case.py $N 50 > $O/slab_case_$N.log 2>&1
DB=$(find $O/p_slab -name '*_results.db' | head -1)
python3 tools/rocprof_summary.py timeline $DB $O/slab_timeline_$N.txt 60
python3 tools/rocprof_summary.py stats $DB $O/slab_stats_$N.csv $O/slab_stats_$N.md
python3 -c "import sqlite3,sys; con=sqlite3.connect('$DB'); print([r[0] for r in con.execute(\"select name from sqlite_master where type in ('table','view')\")])" > $O/slab_db_tables.txt
rm -rf $O/p_slab
