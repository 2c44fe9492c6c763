#!/bin/bash
# round 3, call C: full GPU suite (new: interleaved PCM, product hook, full-size cfg4/cfg5a, compile, lfilter orders)
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3c/pytest.log 2>&1
tail -6 gpurun_out/r3c/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3c/smoke.log 2>&1; tail -2 gpurun_out/r3c/smoke.log
