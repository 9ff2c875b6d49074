#!/bin/bash
# Round-5 GPU sessions: tools/gpu_r5.sh <tag> <stage> [<stage> ...]   (everything lands in gpurun_out/<tag>/)
#   bintest   the binning tests (persistent vs look-back pipeline, ordered-subsequence with culling on / off, speculative sort)
#   binbench  tools/binbench.py: K3-K7 alone in the four gsr_set_bin_persistent modes (c1 whole frame with culling off / on,
#             tile rows 20-47, a 1/8 band); BINBENCH_LIBS="--lib variants/libgsraster_X.so ..." adds experiment builds
#   binbig    the same at configs[2]'s 6 M Gaussians and at a 5 M / 4K eighth band
#   cull      the exact-tile-culling tests, K3-K7 at 4K and the c1 / 4K training steps with GSR_TILE_CULL = 0 / 1
#   shapes    bench.py --workload c2 / c4 on one GPU, persistent binning on / off
#   fakeab    tools/fake_world_bench.py --workload c2 --worlds 1 8 with GSR_BIN_PERSIST = 1 / 0
#   fullsize  tests/test_gpu_fullsize.py
#   any other stage name is handed to tools/gpu_run.sh (test, quick, bench, benchq, pmc, fake, fakeg, ...)
set -u
TAG=${1:-r05}; shift || true
STAGES=${*:-bintest binbench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
cd $R
summ() {  # json file -> one line
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["kernels"]
    print("%-14s %7.1f img/s  %.3f ms  views %7.1f | " % (sys.argv[2], d["value"], d["ms_per_step"], d["rendered_views_per_sec"]) +
          "  ".join("%s %.4f" % (n[:16], v["avg_ms"]) for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
if has bintest; then
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "persistent or binning or speculative" > $O/bintest.log 2>&1
  echo "bintest pytest exit $?" | tee -a $O/bintest.log
  tail -15 $O/bintest.log | cut -c1-300
fi
if has binbench; then
  GSR_BIN_PERSIST_MAXD=100000000 timeout 300 python tools/binbench.py --iters 20 --cull 0 1 ${BINBENCH_LIBS:-} > $O/binbench_c1.txt 2> $O/binbench_c1.err
  echo "binbench c1 exit $?"; cat $O/binbench_c1.txt; tail -3 $O/binbench_c1.err | cut -c1-300
  timeout 300 python tools/binbench.py --iters 20 --band 30 39 > $O/binbench_band.txt 2> $O/binbench_band.err
  GSR_BIN_PERSIST_MAXD=100000000 timeout 300 python tools/binbench.py --iters 20 --band 20 48 > $O/binbench_band2.txt 2> $O/binbench_band2.err; cat $O/binbench_band2.txt
  echo "binbench band exit $?"; cat $O/binbench_band.txt; tail -3 $O/binbench_band.err | cut -c1-300
fi
if has binbig; then
  timeout 300 python tools/binbench.py --iters 10 --gaussians 6000000 > $O/binbench_c2.txt 2> $O/binbench_c2.err
  echo "binbench c2 exit $?"; cat $O/binbench_c2.txt; tail -3 $O/binbench_c2.err | cut -c1-300
  timeout 400 python tools/binbench.py --iters 5 --gaussians 5000000 --width 3840 --height 2160 --band 60 77 > $O/binbench_c4band.txt 2> $O/binbench_c4band.err
  echo "binbench c4 band exit $?"; cat $O/binbench_c4band.txt; tail -3 $O/binbench_c4band.err | cut -c1-300
fi
if has fullsize; then
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -p no:cacheprovider > $O/fullsize.log 2>&1
  echo "fullsize pytest exit $?" | tee -a $O/fullsize.log
  tail -6 $O/fullsize.log | cut -c1-300
fi
for st in $STAGES; do
  case $st in bintest|binbench|binbig|fullsize|shapes|fakeab|cull) ;; *) bash tools/gpu_run.sh $TAG $st ;; esac
done
if has shapes; then
  for wl in c2 c4; do
    timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 10 --warmup 4 --repeats 1 --render-steps 4 > $O/bench_${wl}_shape.json 2> $O/bench_${wl}_shape.err
    echo "bench $wl shape exit $?"; summ $O/bench_${wl}_shape.json $wl
    GSR_BIN_PERSIST=0 timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 10 --warmup 4 --repeats 1 --render-steps 4 > $O/bench_${wl}_shape_off.json 2> $O/bench_${wl}_shape_off.err
    summ $O/bench_${wl}_shape_off.json "$wl persist=0"
  done
fi
if has fakeab; then
  for v in 1 0; do
    GSR_BIN_PERSIST=$v timeout 400 python tools/fake_world_bench.py --workload c2 --worlds 1 8 --steps 20 > $O/fake_c2_persist$v.txt 2> $O/fake_c2_persist$v.err
    echo "persist=$v"; cut -c1-600 $O/fake_c2_persist$v.txt
  done
fi
if has cull; then
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -p no:cacheprovider -k "exact_tile_culling" > $O/culltest.log 2>&1
  echo "culltest exit $?"; grep -E "pairs|passed|failed|rror" $O/culltest.log | tail -8
  timeout 300 python tools/binbench.py --iters 10 --width 3840 --height 2160 --modes off both --cull 0 1 > $O/binbench_4k.txt 2> $O/binbench_4k.err
  echo "binbench 4k exit $?"; cat $O/binbench_4k.txt; tail -2 $O/binbench_4k.err | grep -v amdgpu
  for v in 0 1; do
    GSR_TILE_CULL=$v timeout 600 python bench.py --workload c1_4k --no-cpu-baseline --no-extra --steps 10 --warmup 4 --repeats 1 --render-steps 4 > $O/bench_4k_cull$v.json 2> $O/bench_4k_cull$v.err
    summ $O/bench_4k_cull$v.json "c1_4k cull=$v"
    GSR_TILE_CULL=$v timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 6 --repeats 2 --render-steps 10 > $O/bench_c1_cull$v.json 2> $O/bench_c1_cull$v.err
    summ $O/bench_c1_cull$v.json "c1 cull=$v"
  done
fi
