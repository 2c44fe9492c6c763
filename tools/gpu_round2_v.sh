# round 2, call v: Resample on the f16 matrix pipe
set -x
O=gpurun_out/r2v
mkdir -p $O
timeout 120 python tools/bench_resample_paths.py > $O/resample_paths.jsonl 2> $O/resample_paths.err; echo "bench rc=$?"; tail -3 $O/resample_paths.err; cat $O/resample_paths.jsonl
timeout 600 python -m pytest tests/test_gpu_00_preflight.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_autograd_f64.py tests/test_torch_shim.py -m gpu -x -q -k "preflight or resampl or Resampl or pitch or speed" > $O/pytest_rs.log 2>&1; echo "rc=$?" >> $O/pytest_rs.log
tail -5 $O/pytest_rs.log
