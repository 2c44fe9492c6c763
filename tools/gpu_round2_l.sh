set -x
O=gpurun_out/r2l
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 200 python tools/gpu_microbench.py mel > $O/micro_mel.log 2>&1; head -4 $O/micro_mel.log
timeout 300 python - > $O/pitch.log 2>&1 <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
import audio_amd.functional as F, audio_amd.transforms as T
x = (0.3*torch.randn(64, 160000, device='cuda')).clamp_(-1,1)
ps = T.PitchShift(16000, 4).cuda()
with torch.no_grad():
    for _ in range(3): y = ps(x)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): y = ps(x)
    torch.cuda.synchronize(); print("PitchShift 64 x 10 s: %.2f ms" % ((time.perf_counter()-t0)*100), tuple(y.shape))
    small = (0.3*torch.randn(8, 16000, device='cuda'))
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    for _ in range(50): mel(small)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(2000): mel(small)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("small-batch MelSpectrogram (8 x 1 s): host issue %.1f us/call, wall %.1f us/call" % ((t1-t0)/2000*1e6, (t2-t0)/2000*1e6))
PY
cat $O/pitch.log
