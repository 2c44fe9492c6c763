"""bench.py's launch contract (`--gpus` was parsed and ignored).  No GPU here: --selftest-cpu runs the
same self-spawn / rendezvous / barrier / MAX-reduce / one-JSON-line path over gloo with a CPU stand-in for the step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_self_spawns_two_ranks_and_reports_n_gpus_2():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2 and lines[0]["steps"] == 3


def test_under_the_drivers_own_torchrun_launch():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--selftest-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_world_size_that_disagrees_with_gpus_is_an_error_not_n_gpus_1():
    r = _run(["--gpus", "1", "--selftest-cpu"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert not _json_lines(r.stdout)


def test_default_invocation_is_one_rank():
    r = _run(["--steps", "2", "--warmup", "1", "--selftest-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1


def test_mfcc_op_and_scatter_gather_modes_over_two_ranks():
    """VERDICT r2 item 5(b, c): `--op mfcc` (configs[3]: the one op with a collective, an all-reduce(MAX) per step) and
    `--scatter-gather` (root-born batch scattered / features gathered, timed separately) run multi-rank through the same
    launch path; per-rank step times are reported next to the MAX."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--op", "mfcc", "--scatter-gather", "--selftest-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["op"] == "mfcc"
    assert len(line["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in line["per_rank_ms_per_step"])
    sg = line["scatter_gather"]
    assert sg["root_batch_rows"] == 9 and sg["scatter_ms"] > 0 and sg["gather_ms"] > 0


def test_bench_source_names_both_baseline_configs():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "BASELINE configs[1]" in src and "BASELINE configs[3]" in src and "ShardedTransform" in src


def test_curve_runs_every_point_in_one_invocation_and_prints_one_line():
    """`bench.py --gpus N --curve` = the 1, 2, ..., N-rank points back to back, ONE JSON line with
    `curve: [{n, value, per_rank_ms_min_max, rccl_ranks, efficiency_vs_1}]`; the top-level fields are the largest N's."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-cpu", "--curve"], timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    pts = line["curve"]
    assert [p["n"] for p in pts] == [1, 2]
    assert all(p["value"] > 0 and p["ms_per_step"] > 0 for p in pts)
    assert pts[0]["efficiency_vs_1"] == 1.0 and pts[1]["efficiency_vs_1"] > 0
    assert pts[0]["rccl_ranks"] == 1 and pts[1]["rccl_ranks"] == 2
    assert len(pts[1]["per_rank_ms_min_max"]) == 2 and pts[0]["per_rank_ms_min_max"] is None


def test_curve_with_three_ranks_adds_the_odd_endpoint_and_refuses_a_torchrun_launch():
    r = _run(["--gpus", "3", "--steps", "2", "--warmup", "1", "--selftest-cpu", "--curve"], timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [p["n"] for p in _json_lines(r.stdout)[0]["curve"]] == [1, 2, 3]
    r = _run(["--gpus", "2", "--selftest-cpu", "--curve"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "--curve" in (r.stderr + r.stdout)
