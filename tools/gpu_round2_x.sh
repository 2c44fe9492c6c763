timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_autograd_f64.py -m gpu -x -q -k "resampl or Resampl or pitch or Pitch" 2>&1 | tail -2
timeout 300 python tools/bench_resample_rates.py 2>/dev/null
timeout 120 python tools/bench_resample_paths.py 2>/dev/null | head -1
