#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): audio-sec/sec of MelSpectrogram
(n_fft=400, hop=160, n_mels=80) on batch = 256 x 10 s @ 16 kHz fp32 PER GPU (weak scaling).

  python bench.py --gpus N --steps K --warmup W

  * N > 1 without a torch.distributed.run environment: bench.py re-executes ITSELF under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU,
    RCCL process group.  Under the driver's own torch.distributed.run launch the environment is used as it is; a
    WORLD_SIZE that disagrees with --gpus is an error, not a silent n_gpus = 1.
  * Batches are born sharded -- the path is embarrassingly parallel over clips, so there is no data-path collective;
    the only communication is the barrier + MAX-reduce of the timing.
  * A step = one MelSpectrogram pass over one resident 256-clip batch.  Steps rotate over RING distinct input batches
    and RING+1 distinct output buffers (>= 1 GB touched per cycle), so consecutive launches cannot be served by the
    256 MiB Infinity Cache: the roofline fraction is an HBM figure.
  * --clock-ramp N (default 600, reported in the JSON as `clock_ramp_launches`) = launches issued before the W warm-up
    steps to take a fresh process from idle clocks to steady state (the first ~300 launches run ~15 % slow while DVFS
    ramps).  It is set-up, outside both W and the timed K; pass --clock-ramp 0 to see the cold number.

Prints ONE JSON line on rank 0 with the driver's fields plus `roofline` (dominant kernel vs the HBM roofline, timed
live with HIP events on the launch stream; `traffic` measured in-run with rocprofv3 PMC passes when rocprofv3 is on
PATH) and `cpu_baseline` (the reference's CPU composition timed on this box's host cores).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH, SECONDS, SR, N_FFT, HOP, N_MELS = 256, 10.0, 16000, 400, 160, 80
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
RING = 4                  # distinct input batches in flight: 4 x (164 MB in + 82 MB out) ~ 1 GB > 256 MiB L3
KERNEL_KEY = "melspec400"


def committed_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_melspec400.json")) as f:
            return float(json.load(f)["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def measure_traffic(timeout_s=150):
    """HBM bytes per launch of the headline kernel from two rocprofv3 PMC passes run NOW on this GPU (FETCH_SIZE and
    WRITE_SIZE need separate passes: TCC counter slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"), each a child
    process of this file in --pmc-child mode (same shapes, same ring).  gfx950 correction per the guide's HBM section:
    FETCH_SIZE counts the 128-byte requests of 16-B/lane streaming reads at 64 B -> x2; WRITE_SIZE as reported.
    Returns (bytes or None, detail dict)."""
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None, {"traffic_source": "committed", "why": "rocprofv3 not on PATH"}
    vals = {}
    work = tempfile.mkdtemp(prefix="aamd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, ctr)
            cmd = [prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            per = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if KERNEL_KEY in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None, {"traffic_source": "committed", "why": f"{ctr} pass gave no rows (rc {r.returncode}): "
                              + (r.stderr or "")[-300:]}
            v = [per[k] for k in sorted(per, key=int)]
            v = v[1:] if len(v) > 1 else v                       # drop the first (cold) dispatch
            vals[ctr] = sum(v) / len(v)                          # KB per dispatch
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, {"traffic_source": "committed", "why": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    traffic = vals["FETCH_SIZE"] * 1024.0 * 2.0 + vals["WRITE_SIZE"] * 1024.0
    return traffic, {"traffic_source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                       "FETCH x2 (gfx950 16-B/lane streaming-read correction)",
                     "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"]}


def pmc_child():
    """A few launches of the bench's own loop body for a rocprofv3 counter pass."""
    import torch
    import audio_amd.transforms as T
    dev = torch.device("cuda", 0)
    mel = T.MelSpectrogram(sample_rate=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS).to(dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    xs = [(0.5 * torch.randn(BATCH, int(SECONDS * SR), device=dev, generator=g)).clamp_(-1, 1) for _ in range(RING)]
    ys = [None] * RING
    with torch.no_grad():
        for i in range(2 * RING + 1):
            ys[i % RING] = mel(xs[i % RING])
    torch.cuda.synchronize()


def selftest_cpu(args, world, rank, launched):
    """The multi-process plumbing of main() with gloo and a CPU stand-in for the step (no product code involved)."""
    import torch
    dist = None
    ranks = 1
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        one = torch.ones(1)
        dist.all_reduce(one)
        ranks = int(one.item())
        assert ranks == world
    x = torch.randn(4, 16000)
    w = torch.hann_window(N_FFT)
    for _ in range(args.warmup):
        torch.stft(x, N_FFT, HOP, window=w, return_complex=True)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        torch.stft(x, N_FFT, HOP, window=w, return_complex=True)
    if dist is not None:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "selftest (CPU plumbing only, not a measurement)", "value": 0.0, "unit": "none",
                          "n_gpus": world, "rccl_ranks": ranks, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3, "data": "selftest"}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--clock-ramp", type=int, default=600,
                    help="set-up launches before the warm-up steps (idle clocks -> steady state); reported in the JSON")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 PMC passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="plumbing self-test WITHOUT GPUs (tests/test_bench_cli.py): same launch / rendezvous / barrier / "
                         "MAX-reduce / JSON path over gloo, the step replaced by a tiny CPU STFT; the line says "
                         "data=selftest and is never a measurement")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not launched:
        # self-spawn: one process per GPU under torch.distributed.run, RCCL over xGMI
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
                 "number for a different GPU count than asked for")

    import torch
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank, launched)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert torch.cuda.device_count() > local_rank, \
        f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPUs visible"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl_ranks = 1
    if launched:                                              # RCCL group (also for world == 1 under torchrun)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        # RCCL prints a version banner to STDOUT when the communicator comes up; the driver reads ONE JSON line from
        # stdout, so the file descriptor points at stderr until the communicator exists
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            try:
                dist.init_process_group("nccl", device_id=dev)
            except TypeError:                                 # older signature without device_id
                dist.init_process_group("nccl")
            dist.barrier(device_ids=[local_rank])             # forces communicator creation (and the banner) now
            one = torch.ones(1, device=dev)
            dist.all_reduce(one)                              # how many ranks RCCL really joined
            rccl_ranks = int(one.item())
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        assert rccl_ranks == world, f"RCCL joined {rccl_ranks} ranks, expected {world}"
        print(f"[bench] rank {rank}/{world} on {torch.cuda.get_device_name(local_rank)} (cuda:{local_rank}); "
              f"RCCL all-reduce saw {rccl_ranks} ranks", file=sys.stderr, flush=True)

    import audio_amd.transforms as T
    mel = T.MelSpectrogram(sample_rate=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS).to(dev)
    L = int(SECONDS * SR)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    # RING distinct batches resident in HBM (this rank's shard of the stream of batches)
    xs = [(0.5 * torch.randn(BATCH, L, device=dev, generator=g)).clamp_(-1, 1) for _ in range(RING)]
    ys = [None] * RING     # the consumer holds the last RING feature batches: outputs rotate over RING+1 blocks

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    def step(i):
        ys[i % RING] = mel(xs[i % RING])

    with torch.no_grad():
        for i in range(args.clock_ramp):                     # set-up: idle clocks -> steady state (see docstring)
            step(i)
        torch.cuda.synchronize()
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()                      # the kernels are launched on torch's current stream
        for i in range(args.steps):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps
    y = ys[(args.steps - 1) % RING] if args.steps else mel(xs[0])

    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    n_frames = y.shape[-1]
    assert tuple(y.shape) == (BATCH, N_MELS, n_frames) and n_frames == 1001

    if rank == 0:
        audio_s = world * BATCH * SECONDS * args.steps
        algo_bytes = BATCH * L * 4 + BATCH * n_frames * N_MELS * 4          # 245 821 440 B
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, tdetail = (None, {"traffic_source": "committed", "why": "--no-traffic or N > 1"})
        if world == 1 and not args.no_traffic:
            del xs, ys
            torch.cuda.empty_cache()
            traffic, tdetail = measure_traffic()
        if traffic is None:
            traffic = committed_traffic()
        out = {
            "metric": "audio-sec/sec MelSpectrogram (b=256, 16kHz, n_fft=400, n_mels=80)",
            "value": audio_s / wall,
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_ramp_launches": args.clock_ramp,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "MelSpectrogram n_fft=400 hop=160 n_mels=80, batch=256 x 10 s @16 kHz fp32 per GPU "
                                   "(BASELINE configs[1])", "per_gpu_batch": BATCH, "clip_seconds": SECONDS,
                       "sharding": "clips born sharded across ranks, no data-path collective",
                       "buffer_ring": f"{RING} input batches x {RING + 1} output buffers rotate "
                                      f"({RING * algo_bytes / 1e6:.0f} MB per cycle > 256 MiB Infinity Cache)"},
            "roofline": dict({"bound": "hbm", "kernel": "melspec400_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kernel_ms,
                              "read_only_frac": (BATCH * L * 4) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}, **tdetail),
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import torch_cpu_ref
            n_clips = BATCH
            v, cores, calls = torch_cpu_ref.time_mel_baseline(n_clips, SECONDS, SR, N_FFT, HOP, N_MELS)
            v1 = torch_cpu_ref.time_mel_baseline_single_thread(16, SECONDS, SR, N_FFT, HOP, N_MELS)
            # n in {1, all host cores}, report the best and state the thread count it was measured with
            best, best_cores = (v, cores) if v >= v1 else (v1, 1)
            out["cpu_baseline"] = {"value": best, "unit": "audio-sec/sec", "cores": best_cores, "kind": "port",
                                   "value_all_threads": v, "threads_all": cores, "value_1_thread": v1,
                                   "sample_1_thread": "16 of the 256 clips on one host thread",
                                   "sample": f"all {n_clips} clips x 10 s of one batch, best of {calls} calls; a port, not "
                                             "torchaudio itself (the GPU box has no /root/reference): the same ATen ops "
                                             "torchaudio's CPU path issues (torch.stft + abs().pow(2) + matmul)"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
