#!/bin/bash
mkdir -p gpurun_out/r3g
python tools/mel400_lab.py run base sig --launches 300 --rounds 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3g/lab_sig.txt
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3g/pytest.log 2>&1
tail -3 gpurun_out/r3g/pytest.log
timeout 600 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err
python -c "
import json; r = json.load(open('gpurun_out/r3g/bench.json')); print(r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('valu_issue_frac'), r['roofline'].get('lds_pipe_frac'))"
