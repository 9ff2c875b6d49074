#!/bin/bash
# round-3 session C: full -m gpu suite on the new build, then A/B of the library variants on bench.py's kernel table
mkdir -p gpurun_out/r3c
rm -f gpurun_out/reference_b1_report.txt
( time timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 ) > gpurun_out/r3c/gputest.log 2>&1
tail -12 gpurun_out/r3c/gputest.log
bash tools/gpu_session3.sh r3c ab
