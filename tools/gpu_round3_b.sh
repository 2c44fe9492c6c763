#!/bin/bash
# round 3, call B: product-hook test, full-size cfg4 / cfg5a oracle tests, bench modes
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_distributed.py tests/test_round2_host.py -m gpu -q -x > gpurun_out/r3b/pytest.log 2>&1
tail -5 gpurun_out/r3b/pytest.log
timeout 600 python bench.py --steps 300 --warmup 50 > gpurun_out/r3b/bench_mel.json 2> gpurun_out/r3b/bench_mel.err
cat gpurun_out/r3b/bench_mel.json
timeout 600 python bench.py --op mfcc --scatter-gather --steps 200 --warmup 50 --clock-ramp 200 > gpurun_out/r3b/bench_mfcc.json 2> gpurun_out/r3b/bench_mfcc.err
cat gpurun_out/r3b/bench_mfcc.json; tail -3 gpurun_out/r3b/bench_mfcc.err
