#!/bin/bash
# A/B session: parity tests on the production build, then kbench per library variant
set -u
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_loss_and_step.py -q -m gpu -s -p no:cacheprovider > $O/gputest.log 2>&1
echo "pytest exit $?" | tee -a $O/gputest.log
tail -4 $O/gputest.log
for lib in production $(ls variants/libgsraster_*.so 2>/dev/null); do
  [[ $lib == *stats* ]] && continue
  for view in 0 3; do
    if [[ $lib == production ]]; then unset GSRASTER_LIB; else export GSRASTER_LIB=$R/$lib; fi
    echo "== $lib view $view" >> $O/kbench.txt
    timeout 300 python tools/kbench.py --iters 10 --view $view >> $O/kbench.txt 2>&1
  done
  echo "== $lib low opacity view 0" >> $O/kbench.txt
  timeout 300 python tools/kbench.py --iters 10 --view 0 --opacity-logit-mean -2 --opacity-logit-std 1 >> $O/kbench.txt 2>&1
done
unset GSRASTER_LIB
grep -E "^==|composite|binning" $O/kbench.txt
if [ -f variants/libgsraster_stats.so ]; then
  for v in 0 3; do GSRASTER_LIB=$R/variants/libgsraster_stats.so timeout 300 python tools/kstats_bwd.py $v; done > $O/kstats_bwd.txt 2>&1
fi
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["timing"], d["rendered_views_per_sec"])
for k,v in d["kernels"].items(): print("  ",k,v["avg_ms"])
PY
