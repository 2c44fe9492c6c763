# Every sweep / A-B tool of round 2 in one GPU call (about 3 minutes of box time):  bash tools/gpu_sweeps.sh <tag>
# Results land in gpurun_out/<tag>/ as one .jsonl / .txt per tool; copy what is to be kept into profiles/.
O=gpurun_out/${1:-sweeps}
mkdir -p $O
timeout 300 python tools/bench_configs.py          > $O/configs.jsonl          2> $O/configs.err
timeout 300 python tools/bench_resample_rates.py   > $O/resample_rates.jsonl   2>/dev/null
timeout 200 python tools/bench_resample_paths.py   > $O/resample_paths.jsonl   2>/dev/null
timeout 300 python tools/bench_lfilter_shapes.py 2>/dev/null | grep "^{" > $O/lfilter_shapes.jsonl
timeout 300 python tools/bench_stft_shapes.py   2>/dev/null | grep "^{" > $O/stft_shapes.jsonl
timeout 300 python tools/bench_fftconv_plans.py    > $O/fftconv_plans.jsonl    2>/dev/null
timeout 200 python tools/bench_mfcc_paths.py       > $O/mfcc_paths.jsonl       2>/dev/null
AAMD_RSM_LAB=64 timeout 120 python tools/rsm_census.py > $O/rsm_census.txt 2>&1
for f in $O/*.jsonl; do echo "== $f"; cut -c1-200 $f; done
