#!/usr/bin/env python
"""Regenerate profiles/INDEX.md: one row per kept file -- what it is and which claim of DESIGN.md / HISTORY.md it supports."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DESC = [
    (r'pmc_traffic_melspec400.json', 'HBM traffic per launch of the headline kernel (fallback of bench.py when rocprofv3 is absent)', 'DESIGN 5, `roofline.traffic`'),
    (r'r0\d_.*bench_driver_flags\.json', "bench.py with the DRIVER's flags (--gpus 1 --steps 20 --warmup 5)", 'the driver-run line reproduces on a builder box'),
    (r'r0\d_.*bench.*\.json', 'one JSON line of bench.py (headline metric, roofline, cpu_baseline; round 4: + `configs`, box probe)', 'DESIGN 0 / 5: the headline number of that build'),
    (r'r0\d_.*configs.*\.jsonl', 'tools/bench_configs.py: one line per BASELINE config (per-GPU shard)', 'DESIGN 0 table, 4.2-4.5'),
    (r'r0\d_.*rocprof.*kernel_stats.*|r0\d_.*configs_kernel_stats.*|r0\d_.*rocprof_configs_stats.*', 'rocprofv3 --kernel-trace --stats summary (tools/prof_summary.py)', 'kernel average durations behind the bench lines (DESIGN 5)'),
    (r'r0\d_.*pmc_.*', 'rocprofv3 --pmc counter summaries (separate passes; FETCH x2 per the guide)', 'traffic = 1.002 x algorithmic bytes; VALU / LDS instruction counts, LDS-active and conflict cycles (DESIGN 4.x)'),
    (r'r0\d_.*gpu_pytest.*|r0\d_.*pytest_gpu.*', '`pytest -m gpu` log on an MI355X (full suite unless the name says otherwise; _reverse / _serialize: reversed order, AMD_SERIALIZE_KERNEL=3)', 'parity green on the GPU at that commit'),
    (r'r0\d_.*smoke.*', '__graft_entry__.smoke() log', 'smoke, stage by stage'),
    (r'r0\d_.*mel400_pool.*', 'tools/bench_pool_ab.py: tail pools of the n_fft = 400 kernel vs static tile runs, through the product API (a -DAAMD_M400_POOLS=1 build)', 'DESIGN 4.1 round 5: built, bit-identical, - 0.8 %, compiled out'),
    (r'r0\d_.*rsm_lab.*', 'tools/rsm_lab.py: A/B of resampler variants (csrc/resample_mfma.h compiled alone per -D variant)', 'DESIGN 4.3 round 5: 8-byte operand reads, priorities, conversion placement'),
    (r'r0\d_.*mel400.*', 'tools/mel400_lab.py A/B runs of headline-kernel variants (interleaved, rotating buffers)', 'DESIGN 4.1 / HISTORY: what moved the headline kernel and what did not'),
    (r'r0\d_.*fuzz_kaldi.*', 'tools/fuzz_campaign.py --only kaldi_front: the device against oracle/kaldi_oracle.py on random Kaldi configurations', 'DESIGN 2: 1 500 seeds, 0 failed; the oracle against the reference itself on the same seeds'),
    (r'r0\d_.*launch_modes.*', 'bench.py --launch graph | eager on one box', 'DESIGN 5: the K steps as a HIP graph run 83-90 us apart, eager launches 70-71: eager stays the default'),
    (r'r0\d_.*lfw_lab.*', 'tools/lfw_ab.py: A/B of biquad-cascade variants (csrc/lfilter_wave.h compiled alone per -D variant)', 'DESIGN 4.4 round 6: slot 0, split passes; response table / 12 filter waves / mover priority left off'),
    (r'r0\d_.*valu_vgpr_banks.*', 'tools/lab/valu_bank.hip: fp32 VALU rate against the register banks of the sources', 'DESIGN 4.4: no difference'),
    (r'r0\d_.*lfilter.*', 'lfilter kernels: ISA notes, lab ablations, shape sweeps, general-order scans', 'DESIGN 4.4'),
    (r'r0\d_.*mfcc.*', 'MFCC paths: one-kernel vs two-kernel timings, epilogue ablation', 'DESIGN 4.2'),
    (r'r0\d_.*resample.*', 'Resample: f16 kernel census, rate-pair sweep', 'DESIGN 4.3'),
    (r'r0\d_.*fftconv.*|r0\d_.*fdr.*', 'fftconvolve: plan A/Bs (recompute / complex-block delay line / round 4: real-block delay line, its pipelined walk)', 'DESIGN 4.5'),
    (r'r0\d_.*ubench.*|r0\d_.*valu_issue.*|r0\d_.*streams.*', 'micro-benchmarks of the chip (VALU / LDS issue rates, streams)', 'HISTORY round 3: 2-2.8 cycles per instruction at 2-3 waves per SIMD'),
    (r'r0\d_.*microbench.*|r0\d_.*shapes.*|r0\d_.*istft.*|r0\d_.*pitchshift.*|r0\d_.*stft.*', 'tools/gpu_microbench.py / shape sweeps of the widening ops (iSTFT, autograd, RNN-T, PitchShift, STFT sizes)', 'DESIGN 4.6-4.8, 7'),
]


def main():
    d = os.path.join(ROOT, "profiles")
    out = ['# profiles/ index\n',
           'Every file here was produced on an MI355X through `gpurun` (scratch under `gpurun_out/`, copied here to be judged).  Names are',
           '`rNN_<step>_<what>`: round, step within the round (alphabetical = chronological; `zz` / `end` / `z` = the last verification',
           'of a round), content.  Intermediate re-runs that a later file of the same round supersedes were removed in round 4 (git',
           'history has them).  Regenerate with `python tools/profiles_index.py`.\n',
           '| file | what it is | claim it supports |', '|---|---|---|']
    for f in sorted(os.listdir(d)):
        if f == "INDEX.md":
            continue
        for pat, what, claim in DESC:
            if re.fullmatch(pat, f):
                out.append(f"| `{f}` | {what} | {claim} |")
                break
        else:
            out.append(f"| `{f}` | (see name) | |")
    with open(os.path.join(d, "INDEX.md"), "w") as fh:
        fh.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
