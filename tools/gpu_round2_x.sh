set -x
mkdir -p gpurun_out/r2x
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_transforms_gpu.py -m gpu -x -q -k "resampl or Resampl" 2>&1 | tail -2
for r in 1 2 3; do timeout 120 python tools/bench_resample_paths.py 2>/dev/null | head -2; done
AAMD_RSM_LAB=64 timeout 120 python tools/rsm_census.py > gpurun_out/r2x/census.txt 2>&1
timeout 300 python tools/bench_configs.py 2>/dev/null | grep -i resample | cut -c1-300
