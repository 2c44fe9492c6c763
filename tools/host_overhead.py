import time, torch, cProfile, pstats, io, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import audio_amd.transforms as T, audio_amd.functional as F
dev = torch.device("cuda")
x = torch.randn(2, 1600, device=dev)
mods = {"mel": T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80).to(dev),
        "spec": T.Spectrogram(n_fft=400, hop_length=160).to(dev),
        "mfcc": T.MFCC(16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev),
        "resample": T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser").to(dev)}
with torch.no_grad():
    for name, m in mods.items():
        for _ in range(50): m(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(2000): m(x)
        host = (time.perf_counter() - t) / 2000 * 1e6
        torch.cuda.synchronize()
        print(name, "host us per call (tiny input, no sync): %.1f" % host, flush=True)
    m = mods["mel"]
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): m(x)
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
