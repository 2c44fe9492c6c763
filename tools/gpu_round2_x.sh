set -x
O=gpurun_out/r2x
mkdir -p $O
rm -f $O/rsm_lab.log
for lab in 0 63; do
  echo "AAMD_RSM_LAB=$lab" >> $O/rsm_lab.log
  AAMD_RSM_LAB=$lab timeout 120 python tools/bench_resample_paths.py 2>/dev/null | head -1 >> $O/rsm_lab.log
done
cat $O/rsm_lab.log
