#!/bin/bash
# densification leg of bench.py with the iteration as a hipGraph (GPU box): bash tools/dens_graph_check.sh <outdir> [bench args...]
O=${1:-gpurun_out/dens_graph}; shift || true
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-extra --repeats 1 --steps 30 --warmup 5 --render-steps 0 --densify-every 10 "$@" 2> $O/bench.err | tail -1 > $O/bench.json
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d.get("graph"))
x = d.get("densification") or {}
print({k: x[k] for k in x if k != "note"})
PY
tail -3 $O/bench.err
