# round 2, call u: the one-kernel MFCC
set -x
O=gpurun_out/r2u
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_00_preflight.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_distributed.py -m gpu -x -q -k "preflight or mfcc or MFCC or librosa" > $O/pytest_mfcc.log 2>&1; echo "rc=$?" >> $O/pytest_mfcc.log
tail -3 $O/pytest_mfcc.log
timeout 300 python tools/bench_mfcc_paths.py > $O/mfcc_paths.jsonl 2> $O/mfcc_paths.err; tail -3 $O/mfcc_paths.err; cat $O/mfcc_paths.jsonl
for lab in 1 2 3 4 8 15; do
  echo "AAMD_MFCC_LAB=$lab" >> $O/mfcc_lab.log
  AAMD_MFCC_LAB=$lab timeout 200 python tools/bench_mfcc_paths.py 2>/dev/null | head -1 >> $O/mfcc_lab.log
done
cat $O/mfcc_lab.log
