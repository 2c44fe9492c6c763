set -x
mkdir -p gpurun_out/r2x
AAMD_RSM_LAB=64 timeout 120 python tools/rsm_census.py > gpurun_out/r2x/census.txt 2>&1; cat gpurun_out/r2x/census.txt
for l in 0 64 2 4 6 1 8; do echo "lab $l"; AAMD_RSM_LAB=$l timeout 120 python tools/bench_resample_paths.py 2>/dev/null | head -1 | cut -c1-90; done
