#!/bin/bash
# A/B of library variants on the c1 and c2 (6 M Gaussians, one GPU) workloads: per-kernel times of both
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4i}
mkdir -p $O
cd $R
run() {
  local name=$1 wl=$2; shift 2
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --workload $wl --repeats 2 --steps ${STEPS:-20} --render-steps 10 > $O/${wl}_$name.json 2> $O/${wl}_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/${wl}_$name.json"))
    print("$wl $name", d["value"], d["timing"]["ms_per_step_all"], "views/s", d["rendered_views_per_sec"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
except Exception as e:
    print("$wl $name failed", e)
PY
}
for wl in c1 c2; do
  run production $wl X=1
  for lib in $(ls variants/libgsraster_*.so 2>/dev/null); do
    n=$(basename $lib .so); n=${n#libgsraster_}
    run $n $wl GSRASTER_LIB=$R/$lib
  done
  run production2 $wl X=1
done
