#!/bin/bash
# rocprofv3 kernel table of the ResNet-18 inference forward (tools/exp/backbone_fwd.py: 160 frames of 480 x 640, 4 forwards) -> gpurun_out/$1
set -e
out=${1:-prof_bb}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$out -o bb -- python3 $GRAFT_REPO_ROOT/tools/exp/backbone_fwd.py > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_db_stats.py gpurun_out/$out 4 | cut -c1-150 > gpurun_out/$out.txt
python - <<PY
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/$out/*_results.db')[0])
for r in c.execute("select name, grid_x, grid_y, count(*), avg(end-start)/1e3 from kernels where name like '%conv3x3_k%' or name like '%conv_s2%' or name like '%stem_k%' group by name, grid_x, grid_y order by 1,2 desc"):
    print(r)
PY
head -9 gpurun_out/$out.txt; tail -1 gpurun_out/$out.txt
