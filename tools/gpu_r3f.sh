#!/bin/bash
# kernel trace + idle-gap analysis of ONE rank of a fake 8-rank (and 4-rank) c2 step
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for W in 8; do
  rm -rf $O/prof_fw$W
  timeout 600 rocprofv3 --kernel-trace -d $O/prof_fw$W -- python $R/tools/fake_world_bench.py --workload c2 --worlds $W --steps 20 --warmup 8 > $O/fw$W.txt 2> $O/fw$W.err
  DB=$(find $O/prof_fw$W -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB 40 > $O/fw${W}_kernels.txt 2>&1
  python $R/tools/gap_analysis.py $DB 12 > $O/fw${W}_gaps.txt 2>&1
  find $O -name "*.db" -size +20M -delete
done
cd $R
head -45 $O/fw8_kernels.txt | cut -c1-200
cat $O/fw8_gaps.txt | head -70
