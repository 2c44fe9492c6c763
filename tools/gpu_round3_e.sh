#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3e/pytest.log 2>&1
tail -4 gpurun_out/r3e/pytest.log
timeout 900 python tools/bench_configs.py > gpurun_out/r3e/configs.jsonl 2> gpurun_out/r3e/configs.err
cat gpurun_out/r3e/configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'][:60], round(r['ms_per_launch'],4), round(r['roofline_hbm']['frac'],3))
"
