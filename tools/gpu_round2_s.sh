# round 2, call s: Kaldi batch extension + the LDS-DMA / one-barrier lfilter wave kernels (one-tile, pipelined, mover)
set -x
O=gpurun_out/r2s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_00_preflight.py tests/test_kaldi.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_autograd_f64.py tests/test_torch_shim.py -m gpu -x -q -k "preflight or kaldi or lfilter or biquad or filt or iir or cascade or core_loop" > $O/pytest_lf.log 2>&1; echo "rc=$?" >> $O/pytest_lf.log
tail -6 $O/pytest_lf.log
rm -f $O/lfw_lab.log
for lab in 0 1 16 32 48; do
  echo "mover (8 + 4 waves, two tiles)" >> $O/lfw_lab.log
  AAMD_LFW_LAB=$lab timeout 120 python tools/lfw_lab.py 2>&1 | grep -v amdgpu.ids >> $O/lfw_lab.log
done
echo "pipe (8 waves, two tiles)" >> $O/lfw_lab.log
AAMD_LFW_PIPE=1 timeout 120 python tools/lfw_lab.py 2>&1 | grep -v amdgpu.ids >> $O/lfw_lab.log
echo "one tile, W=16" >> $O/lfw_lab.log
AAMD_LFW_PIPE=0 timeout 120 python tools/lfw_lab.py 2>&1 | grep -v amdgpu.ids >> $O/lfw_lab.log
cat $O/lfw_lab.log
