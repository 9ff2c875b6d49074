#!/bin/bash
# per-kernel durations of a short bench.py run: tools/kstats.sh <out.txt> [bench.py args...]   (GPU box)
# rocprofv3 --kernel-trace --stats under a hard timeout; never reads stdin (an empty result prints a note instead)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $(dirname $OUT)
D=$(mktemp -d /tmp/kstats.XXXXXX)
cd /tmp && export TMPDIR=/tmp
# (KSTATS_CMD: another command of this repo instead of bench.py, e.g. "python $R/tools/fake_world_bench.py")
CMD=${KSTATS_CMD:-"python $R/bench.py --no-cpu-baseline --no-extra --repeats 1"}
timeout -k 10 ${KSTATS_TIMEOUT:-240} rocprofv3 --kernel-trace --stats -d $D -- $CMD "$@" > $D/run.log 2>&1 < /dev/null
f=$(find $D -name "*.db" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -s "$f" ]; then
  python3 - "$f" > $OUT <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc").fetchall()
except Exception as e:
    print("query failed:", e); rows = []
print("%-64s %8s %12s %10s" % ("kernel", "calls", "total_us", "avg_us"))
for n, k, t, a in rows[:30]:
    print("%-64s %8d %12.1f %10.2f" % (n[:64], k, t / 1e3, a / 1e3))
PY
else
  echo "no rocpd database (rocprofv3 exit / timeout); tail of the run log:" > $OUT
  tail -5 $D/run.log >> $OUT
fi
rm -rf $D
cat $OUT
