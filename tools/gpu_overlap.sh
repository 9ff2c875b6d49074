#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-overlap}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python tools/overlap_trace.py --steps 10 > $O/steps.txt 2>&1; echo "plain exit $?"; tail -3 $O/steps.txt
cd /tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -- python $R/tools/overlap_trace.py --steps 3 > $O/prof.log 2>&1; echo "prof exit $?"
cd $R
python tools/overlap_summary.py $O/prof > $O/overlap.txt 2>&1; cat $O/overlap.txt
find $O -name "*.db" -size +20M -delete
