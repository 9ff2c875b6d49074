#!/bin/bash
# One gpurun call = tests + bench + rocprofv3 passes of the SAME bench command (GPU minutes are scarce: everything
# that needs the box goes into one script).  Usage on the box:  bash tools/gpu_session.sh <tag> [stages...]
# stages: test bench trace sq fetch write lowop multirank   (default: all)
set -u
TAG=${1:-r02}; shift || true
STAGES=${*:-test bench trace sq fetch write lowop multirank stats}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extra --steps 16 --warmup 4 --repeats 1 --render-steps 8 ${GSR_SESSION_BENCH_ARGS:-}"
has() { [[ " $STAGES " == *" $1 "* ]]; }
cd $R
if has test; then
  timeout 2400 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $O/gputest.log 2>&1
  echo "pytest exit $?" >> $O/gputest.log
  tail -5 $O/gputest.log
fi
if has bench; then
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -c 600 $O/bench.json
fi
cd /tmp
for st in trace sq fetch write; do
  has $st || continue
  case $st in
    trace) PMC="";;
    sq)    PMC="--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES";;
    fetch) PMC="--pmc FETCH_SIZE";;
    write) PMC="--pmc WRITE_SIZE";;
  esac
  EXTRA=""; [[ $st == fetch || $st == write ]] && EXTRA="--pmc-calib"
  rm -rf $O/prof_$st
  timeout -k 10 600 rocprofv3 --kernel-trace $PMC -d $O/prof_$st -- $BENCH $EXTRA > $O/prof_$st.log 2>&1 < /dev/null
  echo "rocprofv3 $st exit $?"
done
cd $R
if has trace || has sq || has fetch || has write; then
  python tools/pmc_collect.py --trace $O/prof_trace --sq $O/prof_sq --fetch $O/prof_fetch --write $O/prof_write \
      --out $O/pmc.json --command "rocprofv3 --kernel-trace [--pmc ...] -- $BENCH [--pmc-calib]" > $O/pmc.txt 2> $O/pmc.err
  tail -3 $O/pmc.err
  # the databases are large: keep only the summaries
  find $O -name "*.db" -size +20M -delete
fi
if has lowop; then
  timeout 600 python bench.py --no-cpu-baseline --no-extra --opacity-logit-mean -2 --opacity-logit-std 1 > $O/bench_lowopacity.json 2> $O/bench_lowopacity.err
  echo "lowop exit $?"
fi
if has multirank; then
  timeout 900 python tools/multirank_hostprof.py 2 > $O/multirank2.log 2>&1; echo "multirank2 exit $?"
  timeout 900 python tools/multirank_hostprof.py 4 > $O/multirank4.log 2>&1; echo "multirank4 exit $?"
fi
if has stats && [ -f variants/libgsraster_stats.so ]; then
  for v in 0 3; do GSRASTER_LIB=$R/variants/libgsraster_stats.so timeout 300 python tools/kstats_bwd.py $v; done > $O/kstats_bwd.txt 2>&1
  echo "stats exit $?"
fi
ls -la $O | head -40
