timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "fftconvolve or lfilter or resample" 2>&1 | grep -E "^E  *(Assert|assert)|passed|failed" | head
