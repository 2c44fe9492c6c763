#!/bin/bash
# round 3, late: MFCC instantiation with one twiddle batch in registers + row-wise group division -- MFCC parity tests and the
# config sweep twice (two passes over the rows: run-to-run spread on one box).  Output under gpurun_out/r3i_*.
set -x
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "mfcc or MFCC or db or top_db" > gpurun_out/r3i_gpu_pytest_mfcc.log 2>&1
tail -3 gpurun_out/r3i_gpu_pytest_mfcc.log
for i in 1 2; do
timeout 600 python tools/bench_configs.py > gpurun_out/r3i_configs_$i.jsonl 2> gpurun_out/r3i_configs_$i.err
python - <<PY
import json
for l in open('gpurun_out/r3i_configs_$i.jsonl'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config'][:60], round(d['ms_per_launch']*1e3,1), 'us', round(d['roofline_hbm']['frac'],3))
PY
done
