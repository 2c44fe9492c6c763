#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests -m gpu -q -x -k "mfcc or MFCC or hook or complex128 or interleaved" > gpurun_out/r3d/pytest.log 2>&1
tail -4 gpurun_out/r3d/pytest.log
timeout 600 python tools/bench_mfcc_paths.py > gpurun_out/r3d/mfcc_paths.txt 2>&1
cat gpurun_out/r3d/mfcc_paths.txt
