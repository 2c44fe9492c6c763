set -x
O=gpurun_out/r2s
mkdir -p $O
rm -f $O/lfw_lab2.log
for lab in 0 48 50 52 56 63; do
  echo "pipe (8 waves, two tiles)" >> $O/lfw_lab2.log
  AAMD_LFW_LAB=$lab timeout 120 python tools/lfw_lab.py 2>&1 | grep -v amdgpu.ids >> $O/lfw_lab2.log
done
cat $O/lfw_lab2.log
