#!/bin/bash
# round-3 session B: the whole -m gpu suite (incl. the staged-reference live tests), then the N=1 bench line
mkdir -p gpurun_out/r3b
rm -f gpurun_out/reference_b1_report.txt
( time timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -60 ) > gpurun_out/r3b/gputest.log 2>&1
timeout 600 python bench.py > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
tail -25 gpurun_out/r3b/gputest.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print(d['value'], d['ms_per_step'], d['timing'])
for k,v in d['kernels'].items(): print(k, v['avg_ms'])
PY
