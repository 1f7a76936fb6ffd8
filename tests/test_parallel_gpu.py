"""Data-parallel path on the GPU box (one MI355X): the per-shard compute is the HIP policy (vima_amd), the exchange step
is exercised through the C ABI with a world-1 RCCL communicator, the 2-rank code path runs as two processes sharing the
one GPU (RCCL refuses two ranks on one device, so those two exchange over gloo on the host -- the same
`parallel.all_gather_logits` entry), and `bench.py --gpus 2` must refuse to run on a 1-GPU box."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vima_amd import parallel
from vima_testing import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allgather_logits_c_abi_world1():
    """vima_comm_unique_id / vima_comm_create / vima_allgather_logits / vima_comm_destroy on a real RCCL communicator
    of one rank: the gathered buffer equals the local logits, on the caller's stream."""
    dev = torch.device("cuda", 0)
    comm = parallel.LogitsComm(dev, rank=0, world=1)
    assert comm.world == 1 and comm.rank == 0
    x = torch.randn(256, 700, device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        y = comm.all_gather(x, 256)
    s.synchronize()
    assert y.data_ptr() != x.data_ptr() and torch.equal(x, y)
    comm.close()


def _worker(rank, world, port, global_batch, q, skinny=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vima_amd.policy import VIMAPolicy
        dev = "cuda:0"
        cfg = syn.config("4M")
        sd = syn.make_state_dict(cfg, 0)
        pol = VIMAPolicy(**cfg.ctor_kwargs(), precision="bf16", device=dev)
        pol.load_state_dict(sd, strict=True)
        pol.set_option("gemm_skinny", skinny)
        prompts = syn.make_prompt(global_batch, n_segments=2, words_per_segment=3, q_per_view=2, seed=5)
        obs = syn.make_obs(1, global_batch, 2, seed=6)

        def step(lo, hi):
            idx = list(range(lo, hi))
            p = syn.to_device(syn.cut_prompt(prompts, idx), dev)
            o = syn.to_device(syn.cut_obs(obs, idx), dev)
            ptok, pmask = pol.forward_prompt_assembly(p)
            otok, omask = pol.forward_obs_token(o)
            return pol.action_logits(pol.forward(otok, omask, None, ptok, pmask)[-1])

        full = step(0, global_batch).cpu() if rank == 0 else None
        gathered = parallel.data_parallel_logits(step, global_batch).cpu()
        q.put((full, gathered))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch,skinny", [(6, 0), (7, 0), (6, 1)])
def test_two_ranks_hip_shards_equal_full_batch(global_batch, skinny):
    """World size 2, per-shard compute = the HIP policy: concat(shard logits) == full-batch logits (samples are
    independent end to end) and both ranks hold identical gathered logits. Odd global batch exercises the padded ragged tail.
    gemm_skinny = 0: every GEMM tile accumulates K in the same order at every batch size, so the match is exact up to 1e-6.
    Default (gemm_skinny = 1, round 5): GEMMs of at most 32 rows take the K-split kernel, so a 3-sample shard (e.g. 24 prompt-object rows) and the
    6-sample batch (48 rows) sum some products in a different order -- equal to bf16 rounding, a fraction of the 1e-3 gate."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q, skinny)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = [f for f, _ in res if f is not None][0]
    for _, g in res:
        assert g.shape == (global_batch, 700)
        assert torch.allclose(g, full, atol=1e-6 if not skinny else 5e-4, rtol=0), (g - full).abs().max()
    assert torch.equal(res[0][1], res[1][1])


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout, "bench.py printed a result line although fewer GPUs than requested are visible"
    assert "requested but only" in (r.stderr + r.stdout)


def test_bench_launcher_path_with_two_ranks_on_the_one_gpu():
    """`bench.py --gpus 2` re-launches itself under torch.distributed.run, the ranks agree on the collective,
    time the same steps, take the maximum over ranks and rank 0 prints ONE line with the live world size. Only one GPU is reachable
    here and RCCL refuses two ranks on one device, so the test mode VIMA_BENCH_SHARED_GPU=1 puts both ranks on GPU 0 over gloo (tiny
    model); the line is marked as a launcher test, not a measurement."""
    import json
    env = dict(os.environ, VIMA_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "2M",
                        "--batch", "4", "--prompt-len", "64", "--qv", "2", "--words", "4", "--headline-only"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert "gloo" in d["config"]["collective"] and "shared_gpu_test" in d["config"]
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["cores"] >= 1      # carried into N > 1 lines
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_bench_launcher_path_with_eight_ranks_on_the_one_gpu():
    """the first 8-GPU run must not also be the first run of the N = 8 host paths. Eight ranks share GPU 0 over gloo
    (tiny model): rank agreement on the collective, per-rank step times, all-gather timing, MAX over ranks, NUMA pinning and the single
    JSON line have all executed at world size 8; the line carries what a bad scaling curve would need to be diagnosed."""
    import json
    env = dict(os.environ, VIMA_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--model", "2M",
                        "--batch", "2", "--prompt-len", "64", "--qv", "2", "--words", "4", "--headline-only", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp8"
    rk = d["config"]["ranks"]
    assert len(rk["ms_per_step_own"]) == 8 and len(rk["allgather_us"]) == 8
    assert 0 < rk["ms_per_step_own_min"] <= rk["ms_per_step_own_max"] <= d["ms_per_step"] * 1.001 + 1e-3
    assert 0 <= rk["slowest_rank"] < 8 and all(v >= 0 for v in rk["allgather_us"])
    assert len(d["config"]["host_affinity"]) == 8 and all("pinned" in a for a in d["config"]["host_affinity"])
    assert d["value"] > 0
