#!/bin/bash
# A/B session on whole-iteration kernel times: targeted parity tests on the production build, then bench.py's
# per-kernel HIP-event table per library variant (variants/libgsraster_*.so), then a kernel trace of the c4 shape.
set -u
TAG=${1:-r02h}; shift || true
STAGES=${*:-test ab c4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
cd $R
if has test; then
  timeout 1200 python -m pytest tests/test_gpu_loss_and_step.py tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x -p no:cacheprovider > $O/gputest.log 2>&1
  echo "pytest exit $?" | tee -a $O/gputest.log
  tail -4 $O/gputest.log
fi
if has ab; then
  run() {  # name, env assignments...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 6 --repeats 3 --render-steps 10 > $O/ab_$name.json 2> $O/ab_$name.err
    python - <<PY
import json
try:
    d = json.load(open("$O/ab_$name.json"))
    k = d["kernels"]
    print("%-12s %7.1f img/s  med %.3f ms  views %7.1f | " % ("$name", d["value"], d.get("ms_per_step_median") or d["ms_per_step"], d["rendered_views_per_sec"]) +
          "  ".join("%s %.4f" % (n[:14], v["avg_ms"]) for n, v in k.items()))
except Exception as e:
    print("$name FAILED", e)
PY
  }
  run production X=1 | tee -a $O/ab.txt
  for lib in $(ls variants/libgsraster_*.so 2>/dev/null); do
    [[ $lib == *stats* || $lib == *base* ]] && continue
    n=$(basename $lib .so); n=${n#libgsraster_}
    run $n GSRASTER_LIB=$R/$lib | tee -a $O/ab.txt
  done
  run production2 X=1 | tee -a $O/ab.txt
fi
if has c4; then
  cd /tmp
  rm -rf $O/prof_c4
  timeout 600 rocprofv3 --kernel-trace -d $O/prof_c4 -- python $R/bench.py --workload c4 --no-cpu-baseline --steps 6 --warmup 2 --repeats 1 --render-steps 0 > $O/c4.json 2> $O/c4.err
  echo "c4 exit $?"
  cd $R
  DB=$(find $O/prof_c4 -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB 30 > $O/c4_kernel_stats.txt 2>&1
  find $O -name "*.db" -size +20M -delete
  head -30 $O/c4_kernel_stats.txt
  python -c "
import json; d=json.load(open('$O/c4.json')); print(d['value'], d['ms_per_step']); print({k:v['avg_ms'] for k,v in d['kernels'].items()})"
fi
