#!/bin/bash
mkdir -p gpurun_out/r3i
( timeout 1500 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_loss_and_step.py tests/test_gpu_reference_b1.py tests/test_gpu_parity.py -q -x 2>&1 | tail -8 ) > gpurun_out/r3i/tests.log 2>&1
tail -4 gpurun_out/r3i/tests.log
timeout 900 python tools/fake_world_bench.py --workload c2 --worlds 1 2 4 8 > gpurun_out/r3i/fake_world_c2.txt 2> gpurun_out/r3i/fake_world_c2.err
cat gpurun_out/r3i/fake_world_c2.txt | cut -c1-420; tail -2 gpurun_out/r3i/fake_world_c2.err
GSRASTER_LIB=$PWD/variants/libgsraster_xchunk256.so timeout 600 python tools/fake_world_bench.py --workload c2 --worlds 8 > gpurun_out/r3i/fake_world_c2_xchunk256.txt 2>&1
grep world gpurun_out/r3i/fake_world_c2_xchunk256.txt | cut -c1-420
