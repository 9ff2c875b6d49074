#!/bin/bash
# round-3 session A: the reference's Python on the operator (live), fixtures, train.py unchanged
mkdir -p gpurun_out/r3a
rm -f gpurun_out/reference_b1_report.txt
timeout 1500 python -m pytest tests/test_gpu_reference_b1.py -q -k "${1:-live or train_py}" 2>&1 | tail -150 > gpurun_out/r3a/live.log
timeout 600 python tests/golden/make_reference_b1_golden.py gpurun_out/r3a/golden > gpurun_out/r3a/golden.log 2>&1
tail -30 gpurun_out/r3a/live.log
cat gpurun_out/r3a/golden.log | tail -8
