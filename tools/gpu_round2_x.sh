timeout 600 python tools/bench_fftconv_plans.py 2>/dev/null
