timeout 600 python tools/bench_stft_shapes.py 2>&1 | grep "^{"
