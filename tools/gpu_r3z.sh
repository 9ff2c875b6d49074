#!/bin/bash
# round-3 closing session: the whole -m gpu suite, the N=1 bench line, rocprofv3 passes of the same command (PMC),
# the fake-world scaling instrument on the configs[2] and configs[4] shapes
mkdir -p gpurun_out/r3z
rm -f gpurun_out/reference_b1_report.txt
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r3z/gputest.log 2>&1
grep -E "passed|failed" gpurun_out/r3z/gputest.log
bash tools/gpu_session.sh r3z bench trace sq fetch write 2>&1 | grep -v "^total\|^d\|^-" | tail -12
timeout 900 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 > gpurun_out/r3z/fake_world_c2.txt 2> gpurun_out/r3z/fake_world_c2.err
cut -c1-330 gpurun_out/r3z/fake_world_c2.txt
timeout 900 python tools/fake_world_bench.py --workload c4 --worlds 1 8 --steps 8 --warmup 3 > gpurun_out/r3z/fake_world_c4.txt 2> gpurun_out/r3z/fake_world_c4.err
cut -c1-330 gpurun_out/r3z/fake_world_c4.txt; tail -2 gpurun_out/r3z/fake_world_c4.err
