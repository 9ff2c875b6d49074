#!/bin/bash
# latency / LDS counters of the binning kernels on tools/binbench.py (view 0); usage: gpu_pmc_bin.sh TAG
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmcbin}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  rm -rf $O/set$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/set$i -- python $R/tools/binbench.py --iters 5 > $O/set$i.log 2>&1
  echo "set$i exit $?"
done
cd $R
python - <<PY
import sqlite3, glob
for d in sorted(glob.glob("$O/set*/")):
    dbs = glob.glob(d + "**/*.db", recursive=True)
    if not dbs: print(d, "no db"); continue
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%emit_scatter%' or kernel_name like '%radix_onesweep%' or kernel_name like '%touch_count%' or kernel_name like '%scan_gather%' group by kernel_name, counter_name").fetchall()
    for r in rows:
        print("$TAG", r[0].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:34], r[1], r[2], "%.4g" % r[3], "dur_us %.1f" % (r[4]/1e3))
PY
find $O -name "*.db" -delete
