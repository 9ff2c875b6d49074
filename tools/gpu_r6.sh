#!/bin/bash
# Round-6 GPU sessions: tools/gpu_r6.sh <tag> <stage> [<stage> ...]   (everything lands in gpurun_out/<tag>/)
#   k11test   the loss-and-step tests (fused K11 + Adam bit-equality, batched vs per camera)
#   k11ab     bench.py --bsz 4 (and --bsz 2) for the production library and every variants/libgsraster_*.so
#   any other stage name is handed to tools/gpu_r5.sh (and from there to tools/gpu_run.sh)
set -u
TAG=${1:-r06}; shift || true
STAGES=${*:-k11test k11ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
cd $R
line() {  # json -> one line with the K11 + Adam time
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["kernels"]
    print("%-16s %7.1f img/s  %.3f ms/step | " % (sys.argv[2], d["value"], d["ms_per_step"]) +
          "  ".join("%s %.4f" % (n[:18], v["avg_ms"]) for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
if has k11test; then
  timeout 900 python -m pytest tests/test_gpu_loss_and_step.py -q -m gpu -x -p no:cacheprovider > $O/k11test.log 2>&1
  echo "k11test pytest exit $?" | tee -a $O/k11test.log
  tail -5 $O/k11test.log | cut -c1-300
fi
if has k11ab; then
  for bsz in ${K11_BSZ:-4}; do
    for lib in "" $(ls variants/libgsraster_*.so 2>/dev/null); do
      n=production; [ -n "$lib" ] && { n=$(basename $lib .so); n=${n#libgsraster_}; }
      GSRASTER_LIB=${lib:+$R/$lib} timeout 300 python bench.py --bsz $bsz --no-cpu-baseline --no-extra --steps 20 --warmup 5 --repeats 2 --render-steps 2 > $O/k11ab_${n}_b$bsz.json 2> $O/k11ab_${n}_b$bsz.err
      line $O/k11ab_${n}_b$bsz.json "$n bsz$bsz" | tee -a $O/k11ab.txt
    done
  done
fi
rest=""
for st in $STAGES; do case $st in k11test|k11ab) ;; *) rest="$rest $st" ;; esac; done
[ -n "$rest" ] && bash tools/gpu_r5.sh $TAG $rest
exit 0
