#!/bin/bash
# One GPU session, stages chosen on the command line:  tools/gpu_run.sh <tag> <stage> [<stage> ...]
#   fw        tests/test_gpu_fake_world.py (asynchronous multi-rank iteration on one device)
#   test      the whole -m gpu suite
#   quick     the parity / loss-and-step / golden tests only
#   bench     bench.py, default flags (the contract's N = 1 line)          -> bench.json
#   benchq    bench.py --no-extra --no-cpu-baseline (headline workload only) -> benchq.json
#   ab        benchq for the production library and every variants/libgsraster_*.so
#   fake      tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8      -> fake_world_c2.txt
#   fake8     the same, W = 1 and 8 only
#   gtest     tests/test_gpu_graphed_step.py;  fakeg: the fake world with --graph on
#   gaps8     kernel trace of one rank of the fake 8-rank world + idle gaps  -> fake_world_w8_gaps.txt
#   pmc       kernel trace + PMC passes of the bench command (tools/pmc_collect.py) -> pmc.json / pmc.txt
# Everything lands in gpurun_out/<tag>/.
set -u
TAG=${1:-r04}; shift || true
STAGES=${*:-fw benchq}
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
if has fw; then
  timeout 900 python -m pytest tests/test_gpu_fake_world.py -q -m gpu -x -s -p no:cacheprovider > $O/fw.log 2>&1
  echo "fw pytest exit $?" | tee -a $O/fw.log
  grep -E "fake world|passed|failed|Error|error" $O/fw.log | tail -15
fi
if has probe; then
  for w in a2a a2a_uneven allgather allreduce; do
    timeout 120 python tools/probes/rccl_capture_probe.py $w > $O/probe_$w.log 2>&1
    echo "probe $w exit $?"; grep -v amdgpu.ids $O/probe_$w.log | tail -3 | cut -c1-200
  done
fi
if has xprobe; then
  for cfg in "2 0" "2 1" "1 0"; do
    timeout 100 python tools/probes/graph_exchange_probe.py $cfg > "$O/xprobe_${cfg// /_}.log" 2>&1
    echo "xprobe $cfg exit $?"; grep -v amdgpu.ids "$O/xprobe_${cfg// /_}.log" | grep -E "stats|equals|Error|error|Fatal|File \"/(root|tmp)" | tail -6 | cut -c1-250
  done
fi
if has golden; then
  mkdir -p $O/golden
  timeout 900 python tests/golden/make_reference_b1_golden.py $O/golden w2b2 w4b2 w8b4 > $O/golden.log 2>&1
  echo "golden exit $?"; tail -5 $O/golden.log | cut -c1-300
fi
if has b1; then
  timeout 900 python -m pytest tests/test_gpu_reference_b1.py -q -m gpu -x -p no:cacheprovider > $O/b1.log 2>&1
  echo "b1 pytest exit $?"; tail -4 $O/b1.log | cut -c1-300
fi
if has seg; then
  timeout 900 python -m pytest tests/test_gpu_segments.py -q -m gpu -x -p no:cacheprovider > $O/seg.log 2>&1
  echo "seg pytest exit $?"; tail -25 $O/seg.log | cut -c1-300
fi
if has gtest; then
  timeout 900 python -m pytest tests/test_gpu_graphed_step.py -q -m gpu -x -s -p no:cacheprovider > $O/gtest.log 2>&1
  echo "gtest pytest exit $?" | tee -a $O/gtest.log
  tail -25 $O/gtest.log | cut -c1-400
fi
if has fakeg; then
  timeout 600 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 --steps 20 --graph on > $O/fake_world_c2_graph.txt 2> $O/fake_world_c2_graph.err
  cut -c1-700 $O/fake_world_c2_graph.txt; tail -5 $O/fake_world_c2_graph.err
fi
if has quick; then
  timeout 1200 python -m pytest tests/test_gpu_loss_and_step.py tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x -p no:cacheprovider > $O/quick.log 2>&1
  echo "quick pytest exit $?" | tee -a $O/quick.log
  tail -4 $O/quick.log
fi
if has test; then
  timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O/gputest.log 2>&1
  echo "pytest exit $?" | tee -a $O/gputest.log
  tail -6 $O/gputest.log
fi
if has bench; then
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench exit $?"
  summ $O/bench.json bench
fi
if has benchq; then
  timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 6 --repeats 3 --render-steps 10 > $O/benchq.json 2> $O/benchq.err
  summ $O/benchq.json benchq | tee -a $O/ab.txt
fi
if has benchg; then
  timeout 300 python bench.py --graph on --no-cpu-baseline --no-extra --steps 30 --warmup 6 --repeats 3 --render-steps 10 > $O/benchg.json 2> $O/benchg.err
  summ $O/benchg.json "graph=on" | tee -a $O/ab.txt
  python -c "import json; d=json.load(open('$O/benchg.json')); print(d.get('graph'), d['timing'])"
fi
if has ab; then
  for lib in $(ls variants/libgsraster_*.so 2>/dev/null); do
    [[ $lib == *stats* ]] && continue
    n=$(basename $lib .so); n=${n#libgsraster_}
    GSRASTER_LIB=$R/$lib timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 6 --repeats 3 --render-steps 10 > $O/ab_$n.json 2> $O/ab_$n.err
    summ $O/ab_$n.json $n | tee -a $O/ab.txt
  done
  timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 6 --repeats 3 --render-steps 10 > $O/ab_production2.json 2> $O/ab_production2.err
  summ $O/ab_production2.json production2 | tee -a $O/ab.txt
fi
if has abfake8; then
  for lib in "" $(ls variants/libgsraster_*.so 2>/dev/null); do
    [[ $lib == *stats* ]] && continue
    n=production; [ -n "$lib" ] && { n=$(basename $lib .so); n=${n#libgsraster_}; }
    GSRASTER_LIB=${lib:+$R/$lib} timeout 300 python tools/fake_world_bench.py --workload c2 --worlds 8 --steps 20 --graph on > $O/abfake8_$n.txt 2> $O/abfake8_$n.err
    echo "$n: $(cut -c1-420 $O/abfake8_$n.txt | head -1)"
  done
fi
if has abseg; then
  for v in 1 0 1 0; do
    GSR_SEGMENTS=$v timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 6 --repeats 3 --render-steps 10 > $O/abseg_$v.json 2> $O/abseg_$v.err
    summ $O/abseg_$v.json "segments=$v" | tee -a $O/abseg.txt
    GSR_SEGMENTS=$v timeout 300 python tools/fake_world_bench.py --workload c2 --worlds 8 --steps 20 --graph on > $O/abseg_fake8_$v.txt 2> $O/abseg_fake8_$v.err
    echo "segments=$v fake8: $(cut -c1-400 $O/abseg_fake8_$v.txt | head -1)" | tee -a $O/abseg.txt
  done
fi
if has ablow; then
  for v in "1 1" "0 1" "1 0" "0 0"; do
    set -- $v
    GSR_SEGMENTS=$1 GSR_BAND_GRID=$2 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 --repeats 2 --render-steps 5 --opacity-logit-mean -2 --opacity-logit-std 1 > $O/ablow_$1$2.json 2> $O/ablow_$1$2.err
    summ $O/ablow_$1$2.json "lowop seg=$1 band=$2" | tee -a $O/ablow.txt
    GSR_SEGMENTS=$1 GSR_BAND_GRID=$2 timeout 300 python tools/fake_world_bench.py --workload c2 --worlds 4 8 --steps 20 --graph on > $O/ablow_fake_$1$2.txt 2> $O/ablow_fake_$1$2.err
    cut -c1-330 $O/ablow_fake_$1$2.txt | sed "s/^/seg=$1 band=$2 /" | tee -a $O/ablow.txt
  done
fi
if has fake48; then
  timeout 400 python tools/fake_world_bench.py --workload c2 --worlds 4 8 --steps 20 --graph on > $O/fake_world_c2_48.txt 2> $O/fake_world_c2_48.err
  cut -c1-400 $O/fake_world_c2_48.txt
fi
if has fakebal; then
  timeout 900 python tools/fake_world_bench.py --workload c2 --worlds 8 --steps 20 --graph on --balanced 2 > $O/fake_world_c2_balanced.txt 2> $O/fake_world_c2_balanced.err
  cut -c1-900 $O/fake_world_c2_balanced.txt; grep -v amdgpu $O/fake_world_c2_balanced.err | tail -3
fi
if has fake; then
  timeout 600 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 --steps 20 > $O/fake_world_c2.txt 2> $O/fake_world_c2.err
  cut -c1-400 $O/fake_world_c2.txt
fi
if has fake8; then
  timeout 400 python tools/fake_world_bench.py --workload c2 --worlds 1 8 --steps 20 > $O/fake_world_c2.txt 2> $O/fake_world_c2.err
  cut -c1-400 $O/fake_world_c2.txt
fi
if has gaps8; then
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace -d $O/prof_fw -- python $R/tools/fake_world_bench.py --workload c2 --worlds 8 --steps 24 --warmup 6 > $O/fake_world_c2_w8.txt 2> $O/fake_world_c2_w8.err
  cd $R
  DB=$(find $O/prof_fw -name "*.db" | head -1)
  python tools/gap_analysis.py $DB 12 > $O/fake_world_w8_gaps.txt 2>&1
  find $O -name "*.db" -size +8M -delete
  cut -c1-110 $O/fake_world_w8_gaps.txt
fi
if has gapsg; then
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace -d $O/prof_fwg -- python $R/tools/fake_world_bench.py --workload c2 --worlds 8 --steps 24 --warmup 6 --graph on > $O/fake_world_c2_w8_graph.txt 2> $O/fake_world_c2_w8_graph.err
  cd $R
  DB=$(find $O/prof_fwg -name "*.db" | head -1)
  python tools/gap_analysis.py $DB 12 > $O/fake_world_w8_graph_gaps.txt 2>&1
  find $O -name "*.db" -size +8M -delete
  cut -c1-110 $O/fake_world_w8_graph_gaps.txt
fi
if has pmc; then
  bash tools/gpu_session.sh $TAG trace sq fetch write > $O/pmc_session.log 2>&1
  tail -5 $O/pmc_session.log
  # bench.py quotes the PMC figures of the build it runs on: put them where it looks (the session's copy comes home in
  # gpurun_out/<tag>/pmc.json and is committed as profiles/<round>_pmc.json)
  [ -s $O/pmc.json ] && cp $O/pmc.json $R/profiles/${PMC_NAME:-r04_pmc}.json && cp $O/pmc.txt $R/profiles/${PMC_NAME:-r04_pmc}.txt
fi
if has benchfull; then
  timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench exit $?"
  summ $O/bench.json bench
  python -c "import json; d=json.load(open('$O/bench.json')); print(json.dumps(d['roofline'])[:900]); print([ (w['workload'][:30], w.get('value')) for w in d.get('extra_workloads', [])]); print(d['cpu_baseline'])"
fi
