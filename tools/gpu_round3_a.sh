#!/bin/bash
# round 3, call A: lfilter parity (new tests + the un-guarded round-2 test), general-order kernel timing
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_lfilter_orders.py tests/test_gpu_parity.py -m gpu -q -x -k "lfilter or more_than_160" > gpurun_out/r3a/pytest.log 2>&1
tail -5 gpurun_out/r3a/pytest.log
timeout 300 python tools/bench_lfilter_general.py > gpurun_out/r3a/lfilter_general.txt 2>&1
cat gpurun_out/r3a/lfilter_general.txt
