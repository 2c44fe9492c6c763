set -x
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fftconv" 2>&1 | tail -3
for r in 1 2; do
timeout 300 python tools/bench_configs.py 2>/dev/null | grep -i "cfg5b" | cut -c97-230
AAMD_FFTCONV_NO_FDL=1 timeout 300 python tools/bench_configs.py 2>/dev/null | grep -i "cfg5b" | cut -c97-230
done
