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
@pytest.mark.parametrize("fused", [False, True])
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
    # pass 1: what each rank's kernel would hand to the all-reduce
    local_max = []
    monkeypatch.setattr(D, "allreduce_group_max", lambda gm, group=None: local_max.append(gm.clone()))
    for s in shards:
        if s.shape[0]:
            sharded(s)
    combined = torch.stack(local_max).amax(0)
    assert torch.equal(combined, seen["full"])                      # MAX over the shards IS the unsharded maximum
    assert len(local_max) < 2 or not torch.equal(local_max[0], local_max[1])
    # pass 2: every rank receives the combined maximum, as dist.all_reduce(MAX) delivers it
    monkeypatch.setattr(D, "allreduce_group_max", lambda gm, group=None: gm.copy_(combined))
    parts = [sharded(s) for s in shards if s.shape[0]]
    got = torch.cat(parts, 0)
    assert got.shape == full.shape
    assert torch.equal(got, full)                                   # bit for bit: clamp decisions included
    # and WITHOUT the exchange the shard that does not hold the loudest clip clamps differently (the hook matters)
    lone = [mod(s) for s in shards if s.shape[0]]
    if len(lone) == 2:
        assert not torch.equal(torch.cat(lone, 0), full)
    assert mod.group_max_hook is None                               # ShardedTransform restored the caller's hook
