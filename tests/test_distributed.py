"""world_size-2 `gloo` tests of audio_amd.distributed (CPU; the same code runs over RCCL on GPUs).

The transforms themselves need an MI355X, so the sharding plumbing is exercised with CPU stand-ins
built from the oracle's ATen port that follow the SAME hook protocol as audio_amd.transforms.MFCC
(mel -> dB + per-group max -> group_max_hook -> clamp + DCT)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.timeout(300)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audio_amd import _host                      # noqa: E402
from audio_amd import distributed as D           # noqa: E402
from oracle import torch_cpu_ref as R            # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 5, 8, 256, 1001):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class CpuMFCC:
    """CPU stand-in with audio_amd.transforms.MFCC's two-phase structure and hook."""

    def __init__(self):
        self.window = torch.hann_window(400)
        self.fb = _host.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
        self.dct = _host.create_dct(40, 80, "ortho")
        self.group_max_hook = None

    def __call__(self, x):
        if x.numel() == 0:      # torch.stft rejects empty batches; the HIP path returns an empty result
            mel = torch.zeros(tuple(x.shape[:-1]) + (80, 1 + x.shape[-1] // 160))
        else:
            mel = R.mel_spectrogram(x, self.window, self.fb, 400, 160)         # (..., 80, T)
        db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
        packed = x.shape[-2] if x.dim() > 1 else 1
        rows = max(int(np.prod(x.shape[:-1])), 0)
        n_groups = max(rows // max(packed, 1), 1)
        if db.numel():
            gmax = db.reshape(n_groups, -1).amax(-1)
        else:
            gmax = torch.full((n_groups,), float("-inf"))
        if self.group_max_hook is not None:
            self.group_max_hook(gmax)
        if db.numel():
            shp = db.shape
            db = torch.max(db.reshape(n_groups, -1), (gmax - 80.0)[:, None]).reshape(shp)
        return torch.matmul(db.transpose(-1, -2), self.dct).transpose(-1, -2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)
        full = (0.5 * torch.randn(n_rows, 4000, generator=g)).clamp_(-1, 1)
        full[1] *= 1e-5                                    # a quiet clip: clamped only by the GLOBAL cut-off
        full[:, 1500:2100] = 0.0
        m = CpuMFCC()
        expect = m(full)                                   # unsharded reference (every rank can compute it)

        # 1. ragged scatter -> sharded transform with the MAX all-reduce hook -> gather on root
        st = D.ShardedTransform(m)
        got = st.run_from_root(full if rank == 0 else None, tuple(full.shape), torch.device("cpu"))
        res = {"rank": rank}
        if rank == 0:
            res["mfcc_2d_err"] = float((got - expect).abs().max())
            res["shape_ok"] = tuple(got.shape) == tuple(expect.shape)
        assert m.group_max_hook is None                    # hook restored

        # 2. without the all-reduce the quiet clip's shard would pick a different cut-off
        lo, hi = D.shard_range(n_rows, world, rank)
        local_only = m(full[lo:hi])
        res["local_differs"] = float((local_only - expect[lo:hi]).abs().max())

        # 3. (B, C, L) input: per-item cut-offs, no exchange needed -> sharded == unsharded
        x3 = full[:, None, :]
        e3 = m(x3)
        g3 = D.ShardedTransform(m)(x3[lo:hi])
        res["mfcc_3d_err"] = float((g3 - e3[lo:hi]).abs().max())

        # 4. all_gather_batch (ragged) gives every rank the full result
        ag = D.all_gather_batch(g3, n_rows)
        res["allgather_err"] = float((ag - e3).abs().max())

        # 5. an empty shard still joins the collective (n_rows < world)
        tiny = full[:1]
        t_exp = m(tiny)
        t_got = D.ShardedTransform(m).run_from_root(tiny if rank == 0 else None, tuple(tiny.shape), torch.device("cpu"))
        if rank == 0:
            res["tiny_err"] = float((t_got - t_exp).abs().max())
        q.put(res)
    except Exception as e:          # surface the failure instead of leaving the peer in a collective
        q.put({"rank": rank, "error": repr(e)})
        os._exit(1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [5, 6])
def test_sharded_mfcc_matches_unsharded_gloo_world2(n_rows):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    assert not any("error" in r for r in results), results
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by_rank = {r["rank"]: r for r in results}
    r0 = by_rank[0]
    assert r0["shape_ok"] and r0["mfcc_2d_err"] <= 1e-4 and r0["tiny_err"] <= 1e-4
    # the rank that does NOT hold the loud clips must have been changed by the exchange
    assert max(r["local_differs"] for r in results) > 1.0
    for r in results:
        assert r["mfcc_3d_err"] <= 1e-4 and r["allgather_err"] <= 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# The PRODUCT's hook on one GPU: the gloo tests above shard a CPU stand-in; this one runs
# audio_amd.transforms.MFCC itself through audio_amd.distributed.ShardedTransform on two half-batches, with the
# all-reduce replaced by what it would deliver (the maximum over both shards), and requires the concatenation to equal
# the unsharded call BIT FOR BIT -- same clamp decisions, same DCT.
# ---------------------------------------------------------------------------------------------------------------------
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True, "auto"])
@pytest.mark.parametrize("split", [(5, 3), (7, 1), (8, 0)])
def test_product_mfcc_hook_two_shards_on_one_gpu(fused, split, monkeypatch):
    import audio_amd.transforms as T
    from audio_amd import distributed as D
    g = torch.Generator().manual_seed(5)
    n = sum(split)
    x = (0.3 * torch.randn(n, 16000, generator=g)).clamp_(-1, 1)
    x[2, :] *= 1e-3                      # quiet clips: most of their frames sit under the batch-global cut-off,
    x[7, :] *= 1e-3                      # but not under the cut-off of a shard that holds nothing louder
    x[6, 8000:] = 0.0                    # digital silence: the clamp decides these values entirely
    xd = x.cuda()
    mod = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs={"n_fft": 400, "hop_length": 160, "n_mels": 80}).cuda()
    mod.fused = fused
    seen = {}
    mod.group_max_hook = lambda gm: seen.__setitem__("full", gm.clone())
    full = mod(xd)
    mod.group_max_hook = None
    shards = [xd[:split[0]], xd[split[0]:]]
    sharded = D.ShardedTransform(mod)
    # pass 1: what each rank's kernel would hand to the all-reduce.  EVERY rank calls it exactly once -- the rank whose
    # shard is empty included (the one-kernel path returned before the hook and the job hung)
    local_max = []
    monkeypatch.setattr(D, "allreduce_group_max", lambda gm, group=None: local_max.append(gm.clone()))
    for s in shards:
        before = len(local_max)
        out = sharded(s)
        assert len(local_max) == before + 1, "a rank skipped the all-reduce"
        assert out.shape == (s.shape[0], 40, 101)
    if 0 in split:
        assert float(local_max[-1].max()) == float("-inf")          # the empty shard contributes the identity of MAX
    combined = torch.stack(local_max).amax(0)
    assert torch.equal(combined, seen["full"])                      # MAX over the shards IS the unsharded maximum
    assert not torch.equal(local_max[0], local_max[1])
    # pass 2: every rank receives the combined maximum, as dist.all_reduce(MAX) delivers it
    monkeypatch.setattr(D, "allreduce_group_max", lambda gm, group=None: gm.copy_(combined))
    parts = [sharded(s) for s in shards]
    got = torch.cat(parts, 0)
    assert got.shape == full.shape
    assert torch.equal(got, full)                                   # bit for bit: clamp decisions included
    if fused == "auto":
        # under a hook "auto" never decides (every rank must run the same arithmetic): the one-kernel path ran every time
        rep = mod.fused_report()
        assert rep["decided"] is None and rep["calls_two_kernel"] == 0 and rep["path"] == "fused"
    # and WITHOUT the exchange the shard that does not hold the loudest clip clamps differently (the hook matters)
    lone = [mod(s) for s in shards if s.shape[0]]
    if len(lone) == 2:
        assert not torch.equal(torch.cat(lone, 0), full)
    assert mod.group_max_hook is None                               # ShardedTransform restored the caller's hook


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["float64", "grad"])
def test_product_mfcc_hook_precision_and_training_branch(mode, monkeypatch):
    """the float64 / differentiable MFCC branch honours the hook too -- a sharded float64 or training-mode
    MFCC uses the batch-global cut-off, an empty shard joins the exchange, and the gradient of the sharded call equals the
    unsharded one on the rows of each shard (the remote maximum enters as a constant)."""
    import audio_amd.transforms as T
    from audio_amd import distributed as D
    g = torch.Generator().manual_seed(9)
    dt = torch.float64 if mode == "float64" else torch.float32
    x = (0.3 * torch.randn(6, 8000, generator=g)).clamp_(-1, 1).to(dt)
    x[4] *= 1e-3
    x[5, 3000:] = 0.0
    xd = x.cuda()
    mod = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs={"n_fft": 400, "hop_length": 160, "n_mels": 80}).cuda().to(dt)

    def run(inp):
        inp = inp.clone().requires_grad_(mode == "grad")
        out = mod(inp)
        grad = None
        if mode == "grad" and inp.numel():
            grad = torch.autograd.grad(out.square().sum(), inp)[0]
        return out.detach(), grad

    full, gfull = run(xd)
    sharded = D.ShardedTransform(mod)
    local_max = []
    monkeypatch.setattr(D, "allreduce_group_max", lambda gm, group=None: local_max.append(gm.clone()))
    shards = [xd[:4], xd[4:], xd[:0]]
    for s in shards:
        run_out = sharded(s.clone().requires_grad_(mode == "grad"))
        assert run_out.shape[0] == s.shape[0]
    assert len(local_max) == 3 and float(local_max[2].max()) == float("-inf")
    combined = torch.stack([m.to(torch.float64) for m in local_max]).amax(0)
    monkeypatch.setattr(D, "allreduce_group_max", lambda gm, group=None: gm.copy_(combined.to(gm.dtype)))
    outs, grads = [], []
    for s in shards[:2]:
        inp = s.clone().requires_grad_(mode == "grad")
        o = sharded(inp)
        outs.append(o.detach())
        if mode == "grad":
            grads.append(torch.autograd.grad(o.square().sum(), inp)[0])
    got = torch.cat(outs, 0)
    tol = 1e-9 if mode == "float64" else 2e-5
    assert float((got - full).abs().max()) <= tol * float(full.abs().max())
    lone = torch.cat([mod(s) for s in shards[:2]], 0)
    assert float((lone - full).abs().max()) > 1e-3 * float(full.abs().max())     # per-shard cut-offs would differ visibly
    if mode == "grad":
        gg = torch.cat(grads, 0)
        # rows of the shard that holds the maximum carry the amax sub-gradient as in the unsharded call; for the other shard
        # the cut-off is a constant, so only the single element that holds the batch maximum may differ
        assert float((gg[4:] - gfull[4:]).abs().max()) <= 2e-5 * float(gfull[4:].abs().max()) + 1e-9
        assert torch.isfinite(gg).all()
        # ... and on the rank that holds the maximum only the samples under the arg-max frame can differ (the clamped
        # elements of the OTHER shard send their sub-gradient there in the unsharded call; no backward collective here)
        diff = (gg[:4] - gfull[:4]).abs().amax(0)
        assert int((diff > 2e-5 * float(gfull.abs().max())).sum()) <= 400


@pytest.mark.gpu
def test_rccl_communicator_and_allreduce_max_on_this_box():
    """One-rank RCCL leg inside `pytest -m gpu`: the driver's GPU run proves, on its own box, that
    backend "nccl" (= RCCL) creates a communicator, that all_reduce(MAX) runs on the stream the product uses, and that the
    product's sharded MFCC goes through it (torch.distributed.run, one process, 127.0.0.1)."""
    import subprocess
    script = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from audio_amd import distributed as D
import audio_amd.transforms as T
rank, world, dev = D.init_from_env()
assert world == 1 and dev.type == "cuda"
if not dist.is_initialized():
    dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl"
t = torch.tensor([1.5, -3.0], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert t.tolist() == [1.5, -3.0]
g = torch.Generator().manual_seed(0)
x = (0.3 * torch.randn(4, 16000, generator=g)).clamp_(-1, 1).to(dev)
x[3, 4000:] = 0
m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs={"n_fft": 400, "hop_length": 160, "n_mels": 80}).to(dev)
calls = []
def forced(gm, group=None):        # the collective itself, although one rank is the whole world
    calls.append(gm.data_ptr())
    dist.all_reduce(gm, op=dist.ReduceOp.MAX, group=group)
D.allreduce_group_max = forced
for fused in (True, False):
    m.fused = fused
    full = m(x)
    y = D.ShardedTransform(m)(x)
    assert torch.equal(y, full), fused
    e = D.ShardedTransform(m)(x[:0])
    assert e.shape[0] == 0
torch.cuda.synchronize()
assert len(calls) == 4, calls
loc = D.scatter_batch(x, tuple(x.shape), dev)
back = D.gather_batch(m(loc), x.shape[0])
assert torch.equal(back, full)
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.get_device_name(0))
""" % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-c", script]
    # torch.distributed.run takes a script path, not -c: write it next to the test's temp dir
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rccl_leg.py")
        with open(path, "w") as fh:
            fh.write(script)
        cmd = cmd[:-2] + [path]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0 and "RCCL_OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])


_TWO_RANK_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from audio_amd import distributed as D
import audio_amd.transforms as T
rank, world, dev = D.init_from_env()
assert world >= 2 and dev.type == "cuda" and dev.index == int(os.environ["LOCAL_RANK"])
if not dist.is_initialized():
    dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl"
one = torch.ones(1, device=dev)
dist.all_reduce(one)
assert int(one.item()) == world                     # RCCL really joined `world` ranks (not `world` one-rank groups)
# one root-born batch; rank 0 holds it.  Rows are loud except the last, whose tail is silent: the batch-global top_db cut-off
# (functional.py:393-402 on 2-D input) then clamps rows that live on OTHER ranks than the one holding the maximum
N, L = 2 * world + 1, 24000                         # ragged: shard sizes differ by one
g = torch.Generator().manual_seed(11)
full_x = (0.4 * torch.randn(N, L, generator=g)).clamp_(-1, 1)
full_x[-1, 6000:] = 0.0
full_x[0] *= 2.0
m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs={"n_fft": 400, "hop_length": 160, "n_mels": 80}).to(dev)
for fused in (False, True):
    m.fused = fused
    want = m(full_x.to(dev))                        # the unsharded call, on every rank
    a, b = D.shard_range(N, world, rank)
    y = D.ShardedTransform(m)(full_x[a:b].to(dev))  # this rank's shard through the REAL all_reduce(MAX)
    assert torch.equal(y, want[a:b]), ("sharded != unsharded", fused, rank)
    # scatter from the root, transform, gather back on the root
    loc = D.scatter_batch(full_x.to(dev) if rank == 0 else None, (N, L), dev, root=0)
    assert torch.equal(loc.cpu(), full_x[a:b])
    back = D.gather_batch(D.ShardedTransform(m)(loc), N, root=0)
    if rank == 0:
        assert torch.equal(back, want), ("gathered != unsharded", fused)
    # an EMPTY shard on the last rank: it must join the exchange and return an empty result
    n_small = world - 1
    a2, b2 = D.shard_range(n_small, world, rank)
    want2 = m(full_x[:n_small].to(dev))
    y2 = D.ShardedTransform(m)(full_x[a2:b2].to(dev))
    assert y2.shape[0] == b2 - a2 and torch.equal(y2, want2[a2:b2]), ("empty / tiny shards", fused, rank)
torch.cuda.synchronize()
dist.barrier(device_ids=[dev.index])
dist.destroy_process_group()
if rank == 0:
    print("RCCL2_OK", world, torch.cuda.get_device_name(0))
"""


@pytest.mark.gpu
def test_two_rank_rccl_sharded_mfcc_bit_equal_to_unsharded_when_two_gpus_are_visible():
    """the moment a box shows >= 2 GPUs, the product's sharded MFCC runs over REAL RCCL with two ranks
    (torch.distributed.run, one process per GPU, 127.0.0.1): the batch-global top_db cut-off through all_reduce(MAX) on the
    product's stream, ragged and empty shards, scatter_batch / gather_batch -- bit-equal to the unsharded call, on both MFCC
    paths.  Self-skips on a 1-GPU box (every box of this pool so far; the gloo tests above cover the logic on CPU)."""
    import subprocess
    import tempfile
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs >= 2 visible GPUs for a 2-rank RCCL group (this box shows {torch.cuda.device_count()})")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rccl_two_ranks.py")
        with open(path, "w") as fh:
            fh.write(_TWO_RANK_SCRIPT % ROOT)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), path]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0 and "RCCL2_OK 2" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])


def test_two_rank_script_is_valid_python_and_names_the_product_entry_points():
    """(CPU) the script of the 2-GPU test above cannot rot unnoticed on 1-GPU boxes: it compiles, and every audio_amd.distributed
    name it uses exists."""
    src = _TWO_RANK_SCRIPT % ROOT
    compile(src, "rccl_two_ranks.py", "exec")
    for name in ("init_from_env", "shard_range", "ShardedTransform", "scatter_batch", "gather_batch"):
        assert name in src and hasattr(D, name), name
