"""Data-parallel path (vima_amd/parallel.py) on CPU with gloo, world_size 2: sharding + the single all-gather of
logits reproduce the single-process result bit for bit. The per-shard compute is the oracle (checker role only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vima_amd import parallel
from vima_testing import synthetic as syn


def test_shard_bounds_cover_exactly():
    for n in (1, 2, 7, 8, 255, 256, 2048):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.vima_oracle import OraclePolicy
        torch.set_num_threads(2)
        cfg = syn.config("2M")
        sd = syn.make_state_dict(cfg, 0)
        orc = OraclePolicy(sd, **cfg.ctor_kwargs())
        prompts = syn.make_prompt(global_batch, n_segments=1, words_per_segment=2, q_per_view=1, seed=5)
        obs = syn.make_obs(1, global_batch, 1, seed=6)

        def step(lo, hi):
            types, words, imgs = prompts
            # words / images are packed per sample in order: 2 words + 1 image per sample in this layout
            p = (types[lo:hi], words[2 * lo:2 * hi], parallel.shard_batch_dim(imgs, 0, 0, 1).__class__(
                {k: {v: imgs[k][v][lo:hi] for v in imgs[k]} for k in imgs}))
            o = {"objects": syn.MapDict({k: syn.MapDict({v: obs["objects"][k][v][:, lo:hi] for v in obs["objects"][k]})
                                         for k in obs["objects"]}), "ee": obs["ee"][:, lo:hi]}
            return orc.cold_step(p, o)

        full = step(0, global_batch) if rank == 0 else None
        gathered = parallel.data_parallel_logits(step, global_batch)
        if rank == 0:
            q.put((full, gathered))
        else:
            q.put((None, gathered))
        dist.barrier()   # both ranks are done with their collectives before either tears the group down
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [4, 5])
def test_gloo_world2_allgather_equals_full_batch(global_batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = [f for f, _ in res if f is not None][0]
    for _, g in res:
        assert g.shape == (global_batch, 700)
        # per-sample independence: sharded == full batch (fp32 CPU matmuls may differ in blocking -> tight tolerance)
        assert torch.allclose(g, full, atol=5e-6, rtol=0), (g - full).abs().max().item()
    assert torch.equal(res[0][1], res[1][1]), "all ranks must hold identical gathered logits"


def test_bench_dry_ranks_8_executes_the_whole_multi_rank_host_path():
    """the first 8-GPU run of bench.py must not be the first run of its N > 1 plumbing. `bench.py --dry-ranks 8` re-launches
    itself with 8 ranks over gloo and walks the same code path as the RCCL run -- LogitsComm creation (id drawn by rank 0, broadcast, every
    rank joins: here a StubCommBackend checks that all ranks hold the same id), the agreement all-reduce, the rank-0-written state-dict
    cache, K steps with barrier / max-over-ranks timing and per-rank all-gather brackets, a content check of the gathered logits
    ([rank 0 rows, rank 1 rows, ...]), teardown -- and prints one JSON line marked dry_run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-ranks", "8", "--steps", "2", "--warmup", "1", "--batch", "16"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["value"] is None and d["steps"] == 2
    c = d["config"]
    assert c["global_batch"] == 128 and c["parallelism"] == "dp8" and "StubCommBackend" in c["collective"]
    assert c["stub_calls_rank0"][:3] == ["unique_id", "create", "all_gather"]
    assert "rank 0 built it once" in c["state_dict"]
    assert len(c["ranks"]["ms_per_step_own"]) == 8 and len(c["ranks"]["allgather_us"]) == 8
