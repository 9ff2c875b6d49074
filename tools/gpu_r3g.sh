#!/bin/bash
mkdir -p gpurun_out/r3g
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > gpurun_out/r3g/gputest.log 2>&1
tail -6 gpurun_out/r3g/gputest.log
timeout 900 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 > gpurun_out/r3g/fake_world_c2.txt 2> gpurun_out/r3g/fake_world_c2.err
cat gpurun_out/r3g/fake_world_c2.txt; tail -3 gpurun_out/r3g/fake_world_c2.err
timeout 300 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3g/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_all'], d['rendered_views_per_sec'])
print({k:v['avg_ms'] for k,v in d['kernels'].items()})
PY
