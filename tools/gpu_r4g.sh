#!/bin/bash
# one rank of a fake 8-rank world under a kernel trace: per-kernel times and the idle gaps in front of every kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4m}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof_fw -- python $R/tools/fake_world_bench.py --workload c2 --worlds 8 --steps 24 --warmup 6 > $O/fake_world_c2_w8.txt 2> $O/fake_world_c2_w8.err
cd $R
DB=$(find $O/prof_fw -name "*.db" | head -1)
python tools/gap_analysis.py $DB 12 > $O/fake_world_w8_gaps.txt 2>&1
find $O -name "*.db" -size +8M -delete
cat $O/fake_world_w8_gaps.txt | cut -c1-110
cut -c1-300 $O/fake_world_c2_w8.txt | tail -2
