#!/bin/bash
# quick A/B loop: targeted tests of the fused step + K11, the c1 bench line (fused and two-kernel legs), variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4b}
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_gpu_loss_and_step.py tests/test_gpu_golden.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15 ) > $O/test_step.log 2>&1
grep -E "passed|failed|error" $O/test_step.log | tail -3
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline ${EXTRA:---no-extra} --repeats 3 > $O/ab_$name.json 2> $O/ab_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/ab_$name.json"))
    print("$name", d["value"], d["timing"]["ms_per_step_all"], "two-kernel", d["optimizer"]["ms_per_step_two_kernels"], {k: v["avg_ms"] for k, v in d["kernels"].items() if "preprocess" in k or k == "adam"})
    for e in d.get("extra_workloads", []):
        print("   ", e.get("workload", "")[:40], e.get("value"), e.get("ms_per_step"), "two-kernel", e.get("ms_per_step_two_kernels"), {k: v["avg_ms"] for k, v in e.get("dominant_kernels", {}).items()})
except Exception as e:
    print("$name failed", e)
PY
}
EXTRA=" " run production X=1
for lib in $(ls variants/libgsraster_*.so 2>/dev/null); do
  n=$(basename $lib .so); n=${n#libgsraster_}
  run $n GSRASTER_LIB=$R/$lib
done
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-fuse-backward --repeats 3 > $O/ab_unfused.json 2> $O/ab_unfused.err
python - <<PY
import json
d = json.load(open("$O/ab_unfused.json"))
print("unfused", d["value"], d["timing"]["ms_per_step_all"], {k: v["avg_ms"] for k, v in d["kernels"].items() if "preprocess" in k or k == "adam"})
PY
