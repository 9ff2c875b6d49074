#!/bin/bash
# latency / queue-level counters of the composite kernels on kbench (view 0); usage: gpu_pmc_k10.sh TAG [lib.so]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmck10}
LIB=${2:-}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
[ -n "$LIB" ] && export GSRASTER_LIB=$R/$LIB
cd /tmp
i=0
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LEVEL_WAVES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH_LEVEL" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS_STORE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_CYCLES"; do
  i=$((i+1))
  rm -rf $O/set$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/set$i -- python $R/tools/kbench.py --iters 5 --view 0 > $O/set$i.log 2>&1
  echo "set$i exit $?"
done
cd $R
python - <<PY
import sqlite3, glob, os
for d in sorted(glob.glob("$O/set*/")):
    dbs = glob.glob(d + "**/*.db", recursive=True)
    if not dbs: print(d, "no db"); continue
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%composite%' group by kernel_name, counter_name").fetchall()
    for r in rows:
        print("$TAG", r[0].replace("(anonymous namespace)::","").split("(")[0][:40], r[1], r[2], "%.4g" % r[3], "dur_us %.1f" % (r[4]/1e3))
PY
find $O -name "*.db" -delete
