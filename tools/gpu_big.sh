#!/bin/bash
# large single-GPU workloads: the configs[2] / configs[4] scenes on one MI355X (the "same workload on 1 GPU" legs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-big}
mkdir -p $O
cd $R
for w in c2 c4; do
  timeout 600 python bench.py --workload $w --steps 16 --warmup 8 --repeats 1 --render-steps 8 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  echo "$w exit $?"; tail -c 300 $O/bench_$w.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$w.json"))
    print("$w", d["value"], d["ms_per_step"], d["rendered_views_per_sec"])
    for k,v in d["kernels"].items(): print("  ",k,v["launches"], v["avg_ms"], v.get("mean_pairs_D"))
except Exception as e: print("$w failed", e)
PY
done
rocm-smi --showmeminfo vram 2>/dev/null | head -8
