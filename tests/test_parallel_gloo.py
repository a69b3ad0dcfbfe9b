"""The N>1 path on CPU: world_size-2 gloo processes exercise chain sharding, per-chain
seeding and the one exchange step (sample gather)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aehmc_amd import parallel
    lo, hi = parallel.shard_chains(total)
    seeds = parallel.chain_seeds(1000, total)
    assert seeds == [1000 + c for c in range(lo, hi)]
    # stand-in for the per-rank samples: row c holds the chain's global index and seed
    x = torch.tensor([[c, 1000 + c, rank] for c in range(lo, hi)], dtype=torch.float64)
    parallel.barrier()
    g = parallel.gather_samples(x)  # to rank 0 only (the bench / sampling path)
    assert (g is None) == (rank != 0)
    ga = parallel.gather_samples(x, dst=None)  # all-gather variant
    assert ga.shape == (total, 3)
    assert torch.equal(ga[:, 0], torch.arange(total, dtype=torch.float64))
    if rank == 0:
        assert torch.equal(g, ga)
    assert parallel.max_over_ranks(float(rank)) == world - 1
    assert parallel.sum_over_ranks(hi - lo) == total
    # per-rank checksums of the rows' bit patterns (bench.py checks the gather with them)
    sums = parallel.shard_checksums(x)
    assert sums.shape == (world,) and sums[rank] == x.view(torch.int64).sum()
    if rank == 0:
        bounds = [parallel.shard_chains(total, r, world) for r in range(world)]
        assert [int(g[a:b].contiguous().view(torch.int64).sum()) for a, b in bounds] == sums.tolist()
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), g.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])  # even and ragged shards
def test_gather_world2_gloo(tmp_path, total):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "gathered.npy")
    assert g[:, 1].tolist() == [1000 + c for c in range(total)]


def test_shards_partition():
    from aehmc_amd import parallel
    for total in (1, 7, 8, 4096, 32768):
        for w in (1, 2, 4, 8):
            spans = [parallel.shard_chains(total, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_bench_refuses_mismatched_world():
    """`bench.py --gpus N` under a launcher that set a different WORLD_SIZE must not silently run
    one rank and report n_gpus=1 (round-1 finding); it exits before touching any GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr
    env.pop("WORLD_SIZE")
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "GPU(s) visible" in out.stderr


def test_bench_a_failing_rank_ends_the_run():
    """A rank that dies before the rendezvous must not leave its siblings (and the driver) waiting: the parent
    terminates them, relays the failing rank's stderr and exits non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(AEHMC_BENCH_ONE_DEVICE="1", AEHMC_DIST_BACKEND="gloo", AEHMC_BENCH_FAIL_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--chains", "64", "--dim", "256", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "rank 1 exited with code" in out.stderr and "injected failure" in out.stderr
