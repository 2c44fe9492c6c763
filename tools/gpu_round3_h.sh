#!/bin/bash
# round 3, late: lfilter cascade change -- parity tests of the lfilter family, the lab A/B against the previous header, the
# config sweep (cfg5a row) and the VALU issue micro-benchmarks.  Output under gpurun_out/r3h_*.
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "lfilter or biquad or filtfilt or cascade" > gpurun_out/r3h_gpu_pytest_lfilter.log 2>&1
tail -3 gpurun_out/r3h_gpu_pytest_lfilter.log
timeout 300 python tools/lfw_ab.py run n1np head prod head prod --rounds 3 > gpurun_out/r3h_lfw_ab.txt 2>&1
grep variant gpurun_out/r3h_lfw_ab.txt
timeout 600 python tools/bench_configs.py > gpurun_out/r3h_configs.jsonl 2> gpurun_out/r3h_configs.err
python - <<'PY'
import json
for l in open('gpurun_out/r3h_configs.jsonl'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config'][:60], round(d['ms_per_launch']*1e3,1), 'us', round(d['roofline_hbm']['frac'],3))
PY
