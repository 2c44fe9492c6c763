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

  * --op mfcc runs BASELINE configs[3] instead (MFCC n_mfcc=40 on batch = 512 x 10 s per GPU, 2-D input: ONE top_db
    cut-off for the whole -- sharded -- batch, i.e. the path's only collective, an fp32 all-reduce(MAX) between the mel/dB
    kernel and the clamp + DCT kernel, installed by audio_amd.distributed.ShardedTransform); the metric name says so.
  * --scatter-gather additionally times, outside the K steps, moving a ROOT-born batch of N x per-GPU-batch clips to the
    ranks (scatter_batch) and the features back (gather_batch): SURVEY 8(d) "scatter/gather timed separately".
  * N > 1: the line carries every rank's own wall time (`per_rank_ms_per_step`, min / max) next to the MAX that `value`
    is computed from.
  * --curve: the whole weak-scaling curve (1, 2, 4, ... N ranks, one launch each) in ONE invocation, one JSON line with
    `curve: [{n, value, per_rank_ms_min_max, rccl_ranks, efficiency_vs_1}]`.
  * K < 500 (the driver's --steps 20): `roofline.frac` / `achieved` are priced on 1000 further steps of the same loop, run
    right behind the timed K (`priced_on`); the K-step sample is `frac_timed_region`.  `value` is always the K timed steps.

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
N_MFCC = 40
OPS = {
    # op: (per-GPU batch, metric name, workload text, algorithmic bytes per step as f(batch, L, n_frames))
    "mel": (256, "audio-sec/sec MelSpectrogram (b=256, 16kHz, n_fft=400, n_mels=80)",
            "MelSpectrogram n_fft=400 hop=160 n_mels=80, batch=256 x 10 s @16 kHz fp32 per GPU (BASELINE configs[1])",
            lambda b, L, t: b * L * 4 + b * t * N_MELS * 4),
    "mfcc": (512, "audio-sec/sec MFCC (b=512, 16kHz, n_fft=400, n_mels=80, n_mfcc=40, top_db=80 over the batch)",
             "MFCC (MelSpectrogram + amplitude_to_DB(top_db=80, batch-global cut-off) + DCT-II) n_mfcc=40, batch=512 x 10 s "
             "@16 kHz fp32 per GPU (BASELINE configs[3])",
             lambda b, L, t: b * L * 4 + b * t * N_MFCC * 4),
}


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


def measure_issue(kernel_ms, timeout_s=150):
    """Second roofline of the headline kernel: how busy the vector ALU issue port is.  One more
    rocprofv3 PMC pass (SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE, GRBM_GUI_ACTIVE).  A wave64 VALU instruction
    occupies a SIMD-32 for 2 cycles at best (MI355X_MICROARCH.md: `v_fma_f32` 2 cyc; measured 2.4-2.8 in this kernel's
    occupancy), so  valu_issue_frac = SQ_INSTS_VALU x 2 / (SIMDs x shader cycles of the launch);  lds_pipe_frac =
    SQ_LDS_IDX_ACTIVE / (CUs x cycles).  The kernel is priced against the HBM roofline because that is what north_star
    names, but it is these two ports (and the clock the chip sustains under HBM traffic) that bound it."""
    prof = shutil.which("rocprofv3")
    if prof is None:
        return {}
    work = tempfile.mkdtemp(prefix="aamd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for group in (("SQ_INSTS_VALU", "SQ_INSTS_LDS"), ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")):
            out = os.path.join(work, group[0])
            cmd = [prof, "--kernel-trace", "--pmc"] + list(group) + ["--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child"]
            subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            per = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if KERNEL_KEY in row["Kernel_Name"]:
                        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for ctr, d in per.items():
                v = [d[k] for k in sorted(d, key=int)]
                v = v[1:] if len(v) > 1 else v
                vals[ctr] = sum(v) / len(v)
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return {}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    if "SQ_INSTS_VALU" not in vals:
        return {}
    try:
        import torch
        cus = torch.cuda.get_device_properties(0).multi_processor_count
    except Exception:
        cus = 256
    # shader cycles of one launch: the timed launch duration x the clock the chip sustains under this kernel (1.9-2.0 GHz
    # measured with the cycle counter against the 100 MHz wall clock, profiles/r03_b_mel400_lab_io_clock_ab.txt; the
    # GRBM counters run at another rate while counters are collected and are not used)
    clock = 2.0e9
    cycles = kernel_ms * 1e-3 * clock
    res = {"valu_issue_frac": vals["SQ_INSTS_VALU"] * 2.0 / (cus * 4 * cycles),
           "valu_issue_frac_at_measured_2.78_cycles_per_instr": vals["SQ_INSTS_VALU"] * 2.78 / (cus * 4 * cycles),
           "shader_clock_GHz_assumed": clock / 1e9,
           # round 6, by the WALL clock (tools/lab/valu_wall.hip, profiles/r06_zy_valu_issue_wall_clock.txt): whatever the occupancy
           # and the parallelism inside a wave, a SIMD of this chip sustains one wave64 fp32 instruction per 1.05-1.25 ns (the
           # cycle counter slows down as the vector ALUs fill up) -- the share of that rate this launch uses, priced at 1.15 ns
           "valu_issue_frac_of_sustained_wall_clock_rate": vals["SQ_INSTS_VALU"] / (cus * 4) * 1.15e-9 / (kernel_ms * 1e-3),
           "SQ_INSTS_VALU": vals["SQ_INSTS_VALU"], "SQ_INSTS_LDS": vals.get("SQ_INSTS_LDS"),
           "limited_by": "VALU issue + LDS pipe + in-order issue behind vector-memory instructions, at the 1.9-2.0 GHz the "
                         "chip sustains under this kernel's HBM traffic (profiles/r03_b_mel400_lab_io_clock_ab.txt); `bound` "
                         "names the roofline `frac` is priced against"}
    if "SQ_LDS_IDX_ACTIVE" in vals:
        res["lds_pipe_frac"] = vals["SQ_LDS_IDX_ACTIVE"] / (cus * cycles)
    return res


def box_probe(dev):
    """What distinguishes a fast box from a slow one (the same binary ran 65.9 ... 72.5 us across the pool and
    the copy rate did not predict it): the shader clock the chip sustains under a pure vector-ALU load and the skew between
    its XCDs, from aamd_box_probe (every workgroup times the same FMA chain with the cycle counter and the 100 MHz clock)."""
    import ctypes
    import torch
    from audio_amd import _lib
    try:
        L = _lib.lib()
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        n = 4 * cus                                            # 4 x 256 threads per CU: one wave per SIMD and then some
        rec = torch.zeros((n, 4), dtype=torch.int64, device=dev)
        for iters in (2000, 20000, 20000):                     # warm-up, ramp, measured (~1.3 ms of dependent FMAs per wave)
            _lib.check(L.aamd_box_probe(rec.data_ptr(), n, iters, _lib.current_stream(dev)))
        torch.cuda.synchronize()
        r = rec.cpu().double()
        ghz = (r[:, 0] / (r[:, 1] * 10.0))                     # cycles per ns: 100 MHz ticks are 10 ns
        per_xcd = {}
        for x in range(8):
            m = r[:, 2] == x
            if m.any():
                per_xcd[x] = {"blocks": int(m.sum()), "clock_GHz": round(float(ghz[m].mean()), 4),
                              "mean_us": round(float((r[m, 1] * 0.01).mean()), 2)}
        us = [v["mean_us"] for v in per_xcd.values()]
        return {"alu_clock_GHz": round(float(ghz.mean()), 4), "alu_clock_GHz_min_max": [round(float(ghz.min()), 4), round(float(ghz.max()), 4)],
                "xcd_skew": round((max(us) - min(us)) / (sum(us) / len(us)), 4) if us else None, "per_xcd": per_xcd,
                "probe": "aamd_box_probe: 20000 x 64 dependent fp32 FMAs per lane, 4 workgroups of 256 per CU; cycles (s_memtime) "
                         "/ 100 MHz ticks (s_memrealtime) per workgroup; xcd_skew = (max - min) / mean of the per-XCD mean durations"}
    except Exception as e:                                     # a measurement aid must never break the bench line
        return {"alu_clock_GHz": None, "probe_error": f"{type(e).__name__}: {e}"[:200]}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


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
    from audio_amd import distributed as D
    x = torch.randn(4, 16000)
    w = torch.hann_window(N_FFT)

    def cpu_step():
        z = torch.stft(x, N_FFT, HOP, window=w, return_complex=True).abs().pow(2)
        if args.op == "mfcc":                                 # the collective of configs[3]: batch-global maximum
            gmax = z.amax().reshape(1)
            D.allreduce_group_max(gmax)
            z = torch.maximum(z, gmax - 80.0)
        return z

    for _ in range(args.warmup):
        cpu_step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cpu_step()
    mine = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = None
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        got = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([mine], dtype=torch.float64))
        per_rank = [float(v.item()) / max(args.steps, 1) * 1e3 for v in got]
    sg = None
    if args.scatter_gather:
        rows = 4 * world + 1                                  # ragged on purpose
        root = torch.arange(rows * 800, dtype=torch.float32).reshape(rows, 800) if rank == 0 else None
        t1 = time.perf_counter()
        local = D.scatter_batch(root, (rows, 800), torch.device("cpu"), root=0)
        t2 = time.perf_counter()
        back = D.gather_batch(local * 2.0, rows, root=0)
        t3 = time.perf_counter()
        if rank == 0:
            assert torch.equal(back, root * 2.0)
        sg = {"scatter_ms": (t2 - t1) * 1e3, "gather_ms": (t3 - t2) * 1e3, "root_batch_rows": rows}
    if rank == 0:
        line = {"metric": "selftest (CPU plumbing only, not a measurement)",
                "value": world * args.steps / max(float(t.item()), 1e-9), "unit": "selftest steps/s (all ranks)",
                "n_gpus": world, "rccl_ranks": ranks, "steps": args.steps, "warmup": args.warmup, "op": args.op,
                "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3, "data": "selftest"}
        if per_rank is not None:
            line["per_rank_ms_per_step"] = per_rank
            line["per_rank_ms_min_max"] = [min(per_rank), max(per_rank)]
        if sg is not None:
            line["scatter_gather"] = sg
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def curve(args):
    """`--curve`: one child launch per point (1, 2, 4, ..., N ranks; N itself if it is not a power of two), each the exact
    command the driver would run for that N, so a point of the curve and a stand-alone run are the same measurement.  Side
    measurements (configs, PMC passes, CPU baseline) run once, at N = 1.  The efficiency printed here is value(n) / (n x
    value(1)) -- a convenience; the driver computes its own from its own runs."""
    ns, n = [], 1
    while n < args.gpus:
        ns.append(n)
        n *= 2
    ns.append(args.gpus)
    passthrough = []
    skip = False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a == "--curve":
            continue
        if a == "--gpus":
            skip = True
            continue
        if a.startswith("--gpus="):
            continue
        passthrough.append(a)
    points, lines = [], {}
    for n in ns:
        extra = [] if n == 1 else ["--no-configs", "--no-traffic", "--no-cpu-baseline"]
        extra = [e for e in extra if e not in passthrough]
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n)] + passthrough + extra
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        sys.stderr.write(r.stderr[-4000:])
        line = None
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                line = json.loads(ln)
                break
        if r.returncode != 0 or line is None:
            points.append({"n": n, "error": f"rc {r.returncode}: " + (r.stderr or r.stdout)[-300:]})
            continue
        lines[n] = line
        points.append({"n": n, "value": line["value"], "ms_per_step": line["ms_per_step"],
                       "per_rank_ms_min_max": line.get("per_rank_ms_min_max"), "rccl_ranks": line.get("rccl_ranks")})
    base = next((p["value"] for p in points if p["n"] == 1 and "value" in p), None)
    for p in points:
        p["efficiency_vs_1"] = (p["value"] / (p["n"] * base)) if base and "value" in p else None
    top = dict(lines[max(lines)]) if lines else {"metric": OPS[args.op][1], "value": None, "n_gpus": args.gpus}
    if 1 in lines and max(lines) != 1:
        for k in ("roofline", "cpu_baseline", "configs", "box_calibration", "steady_state_1000_steps"):
            if k in lines[1]:
                top[k + "_at_n1"] = lines[1][k]
    top["curve"] = points
    top["curve_note"] = ("one torch.distributed.run launch per point, same K / W / shapes (weak scaling: per-GPU batch fixed); "
                         "top-level fields = the largest N that ran; *_at_n1 = the side measurements of the 1-GPU point")
    print(json.dumps(top), flush=True)
    return 0 if len(lines) == len(ns) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--clock-ramp", type=int, default=600,
                    help="set-up launches before the warm-up steps (idle clocks -> steady state); reported in the JSON")
    ap.add_argument("--op", choices=sorted(OPS), default="mel",
                    help="mel = BASELINE configs[1] (the headline metric, default); mfcc = configs[3], the one op with a collective")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="also time scatter_batch / gather_batch of a root-born batch (outside the K steps)")
    ap.add_argument("--launch", choices=["graph", "eager"], default="eager",
                    help="how the K timed steps are launched: 'eager' (default) = one Python call per step; 'graph' = the K steps "
                         "captured ONCE, before the timed region, in a HIP graph (K kernel nodes over the rotating batches) and "
                         "replayed inside it.  Measured in round 6 (profiles/r06_zv_launch_modes.txt): the graph's kernel nodes "
                         "run 83-90 us apart where eager launches run 70-71 us apart -- not the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the other BASELINE configs (cfg3 / cfg4 / cfg5a / cfg5b shards, measured after the timed K steps "
                         "at N = 1 and reported under `configs`)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 PMC passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="plumbing self-test WITHOUT GPUs (tests/test_bench_cli.py): same launch / rendezvous / barrier / "
                         "MAX-reduce / JSON path over gloo, the step replaced by a tiny CPU STFT; the line says "
                         "data=selftest and is never a measurement")
    ap.add_argument("--curve", action="store_true",
                    help="the weak-scaling curve in ONE invocation: runs 1, 2, 4, ... up to --gpus ranks back to back (each as "
                         "its own torch.distributed.run launch with the same K / W) and prints ONE line whose `curve` holds "
                         "{n, value, ms_per_step, per_rank_ms_min_max, rccl_ranks, efficiency_vs_1} per point; the top-level "
                         "fields are those of the largest N.  Only from a plain `python bench.py` (not under torchrun).")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.curve:
        if launched:
            sys.exit("bench.py: --curve starts its own torch.distributed.run launches; run it as `python bench.py --gpus N --curve`")
        return curve(args)
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
    from audio_amd import distributed as D
    batch, metric, workload, algo_fn = OPS[args.op]
    if args.op == "mel":
        mod = T.MelSpectrogram(sample_rate=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS).to(dev)
        run = mod
        n_out, kernels = N_MELS, ["melspec400_kernel"]
    else:
        mod = T.MFCC(sample_rate=SR, n_mfcc=N_MFCC,
                     melkwargs={"n_fft": N_FFT, "hop_length": HOP, "n_mels": N_MELS}).to(dev)
        # 2-D input: the reference takes ONE top_db cut-off over the whole batch (functional.py:393-402); sharded, that is
        # an all-reduce(MAX) of one float between the two kernels -- ShardedTransform installs it (a no-op group at N = 1)
        run = D.ShardedTransform(mod)
        n_out, kernels = N_MFCC, ["melspec400_kernel (mel + dB + group max)", "mfcc_dct_mfma_kernel (clamp + DCT)"]
    L = int(SECONDS * SR)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    ring = RING if args.op == "mel" else 3                   # mfcc: 3 x (328 MB in + 82 MB out + 164 MB dB mel) > 256 MiB
    # `ring` distinct batches resident in HBM (this rank's shard of the stream of batches)
    xs = [(0.5 * torch.randn(batch, L, device=dev, generator=g)).clamp_(-1, 1) for _ in range(ring)]
    ys = [None] * ring     # the consumer holds the last `ring` feature batches: outputs rotate over ring + 1 blocks

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    def step(i):
        ys[i % ring] = run(xs[i % ring])

    with torch.no_grad():
        for i in range(args.clock_ramp):                     # set-up: idle clocks -> steady state (see docstring)
            step(i)
        torch.cuda.synchronize()
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        # --launch graph: the K timed steps as ONE HIP graph (captured here, outside the timed region, and replayed once untimed so
        # that it is instantiated and uploaded): K kernel nodes over the same rotating batches, nothing skipped or cached.  With one
        # Python call per step the host needs 30-50 us before the FIRST kernel of the timed region is enqueued (3 % of a 20-step
        # region on the host clock) -- but the graph's nodes are dispatched 83-90 us apart against 70-71 us for eager launches of
        # this 66 us kernel (profiles/r06_zv_launch_modes.txt), so eager stays the default.
        use_graph = args.launch == "graph" and 0 < args.steps <= 256 and args.op == "mel"
        graph = None
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for i in range(args.steps):
                    step(i)
            graph.replay()
            torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()                      # the kernels are launched on torch's current stream
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i)
        e1.record()
        torch.cuda.synchronize()
        my_wall = time.perf_counter() - t0                   # this rank's own K steps
        barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps
    eager_ms = None
    if graph is not None:
        # the same K steps with one Python call per step (rounds 1-5's timed region), host clock, outside the timed region
        with torch.no_grad():
            torch.cuda.synchronize()
            te = time.perf_counter()
            for i in range(args.steps):
                step(i)
            torch.cuda.synchronize()
            eager_ms = (time.perf_counter() - te) / args.steps * 1e3
    y = ys[(args.steps - 1) % ring] if args.steps else run(xs[0])
    steady = None
    if args.steps < 500:
        # a short timed region (the driver's --steps 20 is 1.4 ms of GPU time) is a small sample: the same loop over 1000 more
        # steps, OUTSIDE the timed K and never used for `value`, is reported beside it
        with torch.no_grad():
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for i in range(1000):
                step(i)
            s1.record()
            torch.cuda.synchronize()
        steady = {"steps": 1000, "ms_per_step": s0.elapsed_time(s1) / 1000.0,
                  "note": "outside the timed region; not used for `value` / `ms_per_step`"}

    # box calibration, outside the timed region: a plain device-to-device copy of one input batch (read + write).  Boxes of
    # this pool differ (the same kernel ran 66 us on most and 94 us on one, profiles/r03_zz_lfilter_issue.txt); the copy rate
    # measured in the same process says which kind this run was on
    with torch.no_grad():
        dst = torch.empty_like(xs[0])
        for i in range(10):
            dst.copy_(xs[i % ring])
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for i in range(50):
            dst.copy_(xs[i % ring])
        c1.record()
        torch.cuda.synchronize()
        copy_ms = c0.elapsed_time(c1) / 50
        calibration = {"hbm_copy_GBps": 2 * xs[0].numel() * 4 / (copy_ms * 1e-3) / 1e9, "copy_ms": copy_ms,
                       "note": "torch copy_ of one input batch (read + write), 50 launches, outside the timed region; "
                               "5.3 TB/s on the boxes the profiles/ numbers come from"}
        del dst
        calibration.update(box_probe(dev))

    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    per_rank = None
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mine = torch.tensor([my_wall], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [float(v.item()) / max(args.steps, 1) * 1e3 for v in gathered]
    wall = float(t.item())
    n_frames = y.shape[-1]
    assert tuple(y.shape) == (batch, n_out, n_frames) and n_frames == 1001

    sg = None
    if args.scatter_gather:
        # a batch BORN ON RANK 0 (world x per-GPU batch clips): scatter to the ranks, run, gather the features back; each
        # leg timed on its own between barriers, MAX over ranks ("scatter/gather timed separately")
        reps = 5
        full_rows = world * batch
        root_batch = (0.5 * torch.randn(full_rows, L, device=dev, generator=g)).clamp_(-1, 1) if rank == 0 else None
        legs = {"scatter_ms": [], "gather_ms": []}
        with torch.no_grad():
            for rep in range(reps + 1):
                torch.cuda.synchronize(); barrier()
                t0 = time.perf_counter()
                local = D.scatter_batch(root_batch, (full_rows, L), dev, root=0)
                torch.cuda.synchronize(); barrier()
                t1 = time.perf_counter()
                feat = run(local)
                torch.cuda.synchronize(); barrier()
                t2 = time.perf_counter()
                full = D.gather_batch(feat, full_rows, root=0)
                torch.cuda.synchronize(); barrier()
                t3 = time.perf_counter()
                if rep:                                       # first repetition = warm-up (communicator buffers, allocator)
                    legs["scatter_ms"].append((t1 - t0) * 1e3)
                    legs["gather_ms"].append((t3 - t2) * 1e3)
            assert rank != 0 or tuple(full.shape) == (full_rows, n_out, n_frames)
        v = torch.tensor([min(legs["scatter_ms"]), min(legs["gather_ms"])], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
        in_b, out_b = full_rows * L * 4 * (world - 1) / max(world, 1), full_rows * n_frames * n_out * 4 * (world - 1) / max(world, 1)
        sg = {"scatter_ms": float(v[0].item()), "gather_ms": float(v[1].item()), "root_batch_rows": full_rows,
              "bytes_leaving_root": in_b, "bytes_entering_root": out_b,
              "scatter_GBps": (in_b / (float(v[0].item()) * 1e-3) / 1e9) if world > 1 else None,
              "gather_GBps": (out_b / (float(v[1].item()) * 1e-3) / 1e9) if world > 1 else None,
              "note": "best of 5, barrier to barrier, MAX over ranks; N = 1 moves nothing (same-device views)"}
        del root_batch

    if rank == 0:
        audio_s = world * batch * SECONDS * args.steps
        algo_bytes = algo_fn(batch, L, n_frames)             # mel: 245 821 440 B, mfcc: 409 661 440 B
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # with a short timed region (the driver's --steps 20 = 1.4 ms of GPU time) the event-timed sample ran
        # 4.5 % ahead of the same process's own 1000-step steady state.  `roofline.achieved` / `frac` are therefore priced on the
        # steady-state figure whenever K < 500; the K-step sample stays in the line as `*_timed_region` (and `value` /
        # `ms_per_step` stay what the contract says: the K timed steps on the wall clock).
        priced_ms, priced_on = kernel_ms, f"HIP events around the K = {args.steps} timed steps"
        if steady is not None:
            priced_ms = steady["ms_per_step"]
            priced_on = ("HIP events around 1000 further steps of the same loop, run right after the timed K steps (K < 500 is too "
                         "small a sample); the K-step sample is `frac_timed_region`")
        achieved_timed = achieved
        achieved = algo_bytes / (priced_ms * 1e-3) / 1e9
        traffic, tdetail = (None, {"traffic_source": "committed", "why": "--no-traffic, --op mfcc or N > 1"})
        pmc = {}
        mfcc_report = None
        if args.op == "mfcc":
            # the label says what was measured -- the module reports which path its calls took
            mfcc_report = mod.fused_report()
            if mfcc_report["calls_two_kernel"] == 0:
                kernels = ["melspec400_kernel<EPI400_MFCC> pass 0 (mel + dB + DCT on the f16 matrix pipe + group max)",
                           "melspec400_kernel<EPI400_MFCC> fix-up launch (each workgroup checks its share of the tile minima)"]
        xs = ys = None
        torch.cuda.empty_cache()
        configs = None
        if world == 1 and not args.no_configs:
            # the other BASELINE configs (per-GPU shards), after and outside the timed K steps of the headline metric
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                import bench_configs
                which = ["spec", "cfg4", "cfg4_per_item", "cfg3", "cfg5a", "cfg5b"] if args.op == "mel" else \
                        ["cfg2", "cfg3", "cfg5a", "cfg5b"]
                configs = bench_configs.measure_configs(dev, steps=200, warmup=100, which=which)
            except Exception as e:                            # never lose the headline line to a side measurement
                configs = [{"error": f"{type(e).__name__}: {e}"[:400]}]
        if world == 1 and not args.no_traffic and args.op == "mel":
            traffic, tdetail = measure_traffic()
            pmc = measure_issue(priced_ms)
        if traffic is None and args.op == "mel":
            traffic = committed_traffic()
        out = {
            "metric": metric,
            "value": audio_s / wall,
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_ramp_launches": args.clock_ramp,
            "launch": ({"mode": "hip_graph", "note": "the K timed steps are K kernel nodes of one HIP graph, captured and replayed once "
                        "before the timed region; same rotating batches, nothing skipped", "eager_ms_per_step_same_K": eager_ms}
                       if graph is not None else {"mode": "eager", "note": "one Python call per step"}),
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "per_gpu_batch": batch, "clip_seconds": SECONDS,
                       "sharding": "clips born sharded across ranks, no data-path collective" if args.op == "mel" else
                                   "clips born sharded across ranks; one fp32 all-reduce(MAX) per step (the batch-global "
                                   "top_db cut-off) between the two kernels",
                       "buffer_ring": f"{ring} input batches x {ring + 1} output buffers rotate "
                                      f"({ring * algo_bytes / 1e6:.0f} MB per cycle > 256 MiB Infinity Cache)"},
            "roofline": dict({"bound": "hbm", "kernel": " + ".join(kernels), "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "priced_on": priced_on, "priced_ms": priced_ms,
                              "achieved_timed_region": achieved_timed, "frac_timed_region": achieved_timed / HBM_PEAK_GBS,
                              "frac_on_wall_clock": algo_bytes / (wall / args.steps) / 1e9 / HBM_PEAK_GBS,
                              "timing": "`achieved` / `frac`: see `priced_on`; `*_timed_region` from HIP events around the K "
                                        "steps on the launch stream (`kernel_ms`); `frac_on_wall_clock` from `ms_per_step` "
                                        "(host clock, barrier to barrier)",
                              "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kernel_ms,
                              "read_only_frac": (batch * L * 4) / (priced_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}, **tdetail, **pmc),
        }
        out["box_calibration"] = calibration
        if mfcc_report is not None:
            out["mfcc_path"] = mfcc_report
        if configs is not None:
            out["configs"] = configs
        if per_rank is not None:
            out["per_rank_ms_per_step"] = per_rank
            out["per_rank_ms_min_max"] = [min(per_rank), max(per_rank)]
        if sg is not None:
            out["scatter_gather"] = sg
        if steady is not None:
            steady["frac_of_hbm_peak"] = algo_bytes / (steady["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["steady_state_1000_steps"] = steady
        if world == 1 and not args.no_cpu_baseline and args.op == "mel":
            from oracle import torch_cpu_ref
            sweep = torch_cpu_ref.sweep_mel_baseline(batch, SECONDS, SR, N_FFT, HOP, N_MELS)
            best = max(sweep, key=lambda r: r["audio_sec_per_sec"])
            proto = torch_cpu_ref.protocol_set_num_threads(batch, SECONDS, SR, N_FFT, HOP, N_MELS)
            out["cpu_baseline"] = {"value": best["audio_sec_per_sec"], "unit": "audio-sec/sec", "cores": best["threads"],
                                   "host_cores": os.cpu_count(), "threads_best": best["threads"], "cpu_model": cpu_model(),
                                   "kind": "port", "sweep": sweep,
                                   "protocol_set_num_threads": proto,
                                   "protocol_note": "BASELINE.md section 4 as written -- ONE call over the whole batch under "
                                                    "torch.set_num_threads(n), n in {1, all host cores}, best of >= 3 -- is "
                                                    "`protocol_set_num_threads`; `value` is the dealt-threads sweep, which is "
                                                    "MORE favourable to the CPU (the intra-op pool of one call does not scale)",
                                   "sample": f"all {batch} clips x 10 s of one batch per call, clips dealt to `threads` host "
                                             "threads that each run the single-threaded composition on their share (the "
                                             "intra-op pool of one big call does not scale past a few cores); thread counts "
                                             "1 / 8 / 16 / 32 / 64 / all, ~3 s each, the best reported; a port, not "
                                             "torchaudio itself (the GPU box has no /root/reference): the same ATen ops "
                                             "torchaudio's CPU path issues (torch.stft + abs().pow(2) + matmul)"}
            # ... and one CPU figure per BASELINE config of `configs`: ports of the reference's CPU compositions
            # on bounded samples, rows dealt to host threads, ~2.5 s per thread count (oracle/cpu_baselines.py; for lfilter the
            # core loop is the reference's own lfilter.cpp where oracle/_ref travelled with the tree)
            if configs is not None and not any("error" in c for c in configs):
                try:
                    from oracle import cpu_baselines
                    per_cfg = cpu_baselines.measure(budget_s=2.5)
                    for c in configs:
                        key = "cfg4" if c.get("id") == "cfg4_per_item" else c.get("id")
                        if key in per_cfg:
                            c["cpu_baseline"] = per_cfg[key]
                        elif key == "cfg2":
                            c["cpu_baseline"] = "the line's `cpu_baseline`"
                except Exception as e:                        # never lose the line to a side measurement
                    out["configs_cpu_baseline_error"] = f"{type(e).__name__}: {e}"[:300]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
