#!/bin/bash
mkdir -p gpurun_out/r3e
rm -f gpurun_out/reference_b1_report.txt
timeout 1500 python -m pytest tests/test_gpu_reference_b1.py tests/test_gpu_loss_and_step.py tests/test_gpu_two_ranks.py -q -x 2>&1 | tail -30 > gpurun_out/r3e/tests.log
tail -8 gpurun_out/r3e/tests.log
timeout 900 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 > gpurun_out/r3e/fake_world_c2.txt 2> gpurun_out/r3e/fake_world_c2.err
tail -6 gpurun_out/r3e/fake_world_c2.txt; tail -3 gpurun_out/r3e/fake_world_c2.err
