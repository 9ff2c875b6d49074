#!/bin/bash
# exchange kernels: parity tests, then one rank of a fake 8-rank world under a kernel trace + the scaling table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4j}
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_ranks.py -q -m gpu -x -p no:cacheprovider -k "exchange or local2j or two_ranks or rank" 2>&1 | tail -8 ) > $O/test.log 2>&1
grep -E "passed|failed|error" $O/test.log | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof_fw -- python $R/tools/fake_world_bench.py --workload c2 --worlds 8 --steps 16 --warmup 4 > $O/fake_world_c2_w8.txt 2> $O/fake_world_c2_w8.err
cd $R
DB=$(find $O/prof_fw -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 26 > $O/fake_world_c2_w8_kernels.txt 2>&1
[ -n "$DB" ] && python tools/gap_analysis.py $DB 10 --wide > $O/fake_world_c2_w8_gaps.txt 2>&1
find $O -name "*.db" -size +8M -delete
cut -c1-150 $O/fake_world_c2_w8_kernels.txt | head -30
tail -3 $O/fake_world_c2_w8_gaps.txt
timeout 300 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 > $O/fake_world_c2.txt 2> $O/fake_world_c2.err
cut -c1-330 $O/fake_world_c2.txt | tail -6
