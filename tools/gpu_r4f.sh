#!/bin/bash
# PMC traffic of the configs[2] shape on one GPU (6 M Gaussians, 1080p): kernel trace + FETCH_SIZE + WRITE_SIZE passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4k}
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --workload c2 --no-cpu-baseline --no-extra --steps 8 --warmup 3 --repeats 1 --render-steps 4"
cd /tmp
for st in trace fetch write; do
  case $st in
    trace) PMC="";;
    fetch) PMC="--pmc FETCH_SIZE";;
    write) PMC="--pmc WRITE_SIZE";;
  esac
  EXTRA=""; [[ $st == fetch || $st == write ]] && EXTRA="--pmc-calib"
  timeout 300 rocprofv3 --kernel-trace $PMC -d $O/prof_$st -- $BENCH $EXTRA > $O/prof_$st.log 2>&1
  echo "rocprofv3 $st exit $?"
done
cd $R
python tools/pmc_collect.py --trace $O/prof_trace --fetch $O/prof_fetch --write $O/prof_write --out $O/pmc_c2.json \
    --command "rocprofv3 --kernel-trace [--pmc ...] -- $BENCH [--pmc-calib]" > $O/pmc_c2.txt 2> $O/pmc_c2.err
find $O -name "*.db" -size +20M -delete
tail -40 $O/pmc_c2.txt | cut -c1-180
