#!/bin/bash
# parity tests touching the binning / preprocess / exchange kernels, then the c1 bench line with per-kernel times
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4f}
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_loss_and_step.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15 ) > $O/test.log 2>&1
grep -E "passed|failed|error" $O/test.log | tail -3
timeout 300 python bench.py --no-cpu-baseline ${EXTRA:---no-extra} --repeats 3 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["timing"]["ms_per_step_all"], "two-kernel", d["optimizer"]["ms_per_step_two_kernels"], "views/s", d["rendered_views_per_sec"])
print({k: v["avg_ms"] for k, v in d["kernels"].items()})
for e in d.get("extra_workloads", []):
    print("   ", e.get("workload", "")[:40], e.get("value"), e.get("ms_per_step"), "two-kernel", e.get("ms_per_step_two_kernels"), {k: v["avg_ms"] for k, v in e.get("dominant_kernels", {}).items()})
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-extra --steps 16 --warmup 4 --repeats 1 --render-steps 0 > $O/prof_bench.json 2> $O/prof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 30 > $O/kernel_stats.txt 2>&1
find $O -name "*.db" -size +8M -delete
cut -c1-160 $O/kernel_stats.txt | head -34
