#!/bin/bash
# round-3 (second session) A: the fused K11 + Adam step and the restructured exchange kernels -- targeted GPU tests, the
# N = 1 bench line (fused and two-kernel legs, extras), a K11 workgroup-size variant, a bsz-4 run (camera-batched fused
# kernel), one rank of a fake 8-rank world (exchange kernels, fused step on a shard)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests/test_gpu_loss_and_step.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15 ) > $O/test_step.log 2>&1
grep -E "passed|failed|error" $O/test_step.log | tail -3
( time timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "exchange or local2j" 2>&1 | tail -15 ) > $O/test_exchange.log 2>&1
grep -E "passed|failed|error" $O/test_exchange.log | tail -3
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench exit $?"; tail -3 $O/bench.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print("value", d["value"], "ms", d["ms_per_step"], "timing", d["timing"]["ms_per_step_all"], "views/s", d["rendered_views_per_sec"])
    print("optimizer", d["optimizer"])
    print({k: v["avg_ms"] for k, v in d["kernels"].items()})
    print({k: v.get("frac_hbm_peak") for k, v in d["kernels"].items()})
    for e in d.get("extra_workloads", []):
        print(e.get("workload", "")[:60], e.get("value"), e.get("ms_per_step"), "two-kernel", e.get("ms_per_step_two_kernels"), e.get("dominant_kernels"))
except Exception as e:
    print("bench parse failed", e)
PY
if ls variants/libgsraster_*.so > /dev/null 2>&1; then
  for lib in variants/libgsraster_*.so; do
    n=$(basename $lib .so); n=${n#libgsraster_}
    GSRASTER_LIB=$R/$lib timeout 300 python bench.py --no-cpu-baseline --no-extra --repeats 3 > $O/ab_$n.json 2> $O/ab_$n.err
    python - <<PY
import json
try:
    d = json.load(open("$O/ab_$n.json"))
    print("variant $n", d["value"], d["timing"]["ms_per_step_all"], d["optimizer"], {k: v["avg_ms"] for k, v in d["kernels"].items() if "preprocess" in k or k == "adam"})
except Exception as e:
    print("variant $n failed", e)
PY
  done
fi
timeout 300 python bench.py --no-cpu-baseline --no-extra --workload weak --bsz 4 --steps 12 --warmup 4 --repeats 2 > $O/bsz4.json 2> $O/bsz4.err
python - <<PY
import json
try:
    d = json.load(open("$O/bsz4.json"))
    print("bsz4", d["value"], d["ms_per_step"], d["optimizer"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
except Exception as e:
    print("bsz4 failed", e)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof_fw -- python $R/tools/fake_world_bench.py --workload c2 --worlds 8 --steps 16 --warmup 4 > $O/fake_world_c2_w8.txt 2> $O/fake_world_c2_w8.err
cd $R
cut -c1-400 $O/fake_world_c2_w8.txt | tail -4
DB=$(find $O/prof_fw -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 24 > $O/fake_world_c2_w8_kernels.txt 2>&1
find $O -name "*.db" -size +8M -delete
cut -c1-150 $O/fake_world_c2_w8_kernels.txt | head -28
timeout 300 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 > $O/fake_world_c2.txt 2> $O/fake_world_c2.err
cut -c1-300 $O/fake_world_c2.txt | tail -6
