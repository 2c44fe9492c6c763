"""Run a golden-manifest case through the PRODUCT (audio_amd on cuda) -- mirrors how the
reference's own tests call torchaudio."""
import torch

import audio_amd.functional as F
import audio_amd.transforms as T


def run(case, inputs, device="cuda"):
    op, kw = case["op"], dict(case["kwargs"])
    tin = [torch.from_numpy(a).to(device) for a in inputs]
    with torch.no_grad():
        if op == "Spectrogram":
            wf = torch.hamming_window if kw.pop("window", None) == "hamming" else torch.hann_window
            return T.Spectrogram(window_fn=wf, **kw).to(device)(tin[0])
        if op == "MelSpectrogram":
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = T.MelSpectrogram(**kw)
            return m.to(device)(tin[0])
        if op == "MFCC":
            return T.MFCC(**kw).to(device)(tin[0])
        if op == "AmplitudeToDB":
            return T.AmplitudeToDB(kw["stype"], kw["top_db"])(tin[0])
        if op == "MelScale":
            return T.MelScale(**kw).to(device)(tin[0])
        if op == "F.resample":
            return F.resample(tin[0], kw.pop("orig_freq"), kw.pop("new_freq"), **kw)
        if op == "T.Resample":
            return T.Resample(kw.pop("orig_freq"), kw.pop("new_freq"), **kw).to(device)(tin[0])
        if op == "lfilter":
            return F.lfilter(tin[0], tin[1], tin[2], **kw)
        if op == "biquad":
            return F.biquad(tin[0], **kw)
        if op == "filtfilt":
            return F.filtfilt(tin[0], tin[1], tin[2], **kw)
        if op == "lowpass_cascade":
            y = tin[0]
            for fc in kw["cutoffs"]:
                y = F.lowpass_biquad(y, kw["sample_rate"], fc, kw["Q"])
            return y
        if op.endswith("_biquad"):
            return getattr(F, op)(tin[0], **kw)
        if op == "fftconvolve":
            return F.fftconvolve(tin[0], tin[1], kw["mode"])
        if op == "T.FFTConvolve":
            return T.FFTConvolve(kw["mode"])(tin[0], tin[1])
    return None
