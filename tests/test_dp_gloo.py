"""N>1 path on CPU: world_size-2 gloo processes run the data-parallel bookkeeping of bench.py (request placement,
barrier, max-over-ranks of the timed region, latency gather).  No collective touches the data path, so this is all of
the multi-rank logic there is."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sduss_amd import dp


def test_greedy_assign_matches_reference_policy():
    # fixed resolution -> round-robin; mixed resolutions -> least outstanding pixels (greedy.py:26-34)
    assert dp.greedy_assign([1024] * 6, 3) == [0, 1, 2, 0, 1, 2]
    assert dp.greedy_assign([1024, 512, 512, 512, 512, 768], 2) == [0, 1, 1, 1, 1, 0]
    assert dp.greedy_assign([512, 512], 2, outstanding={0: 10 ** 9}) == [1, 1]
    shares = [dp.my_share(10, r, 4) for r in range(4)]
    assert sorted(i for s in shares for i in s) == list(range(10))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = dp.my_share(9, rank, world)
        dist.barrier()
        elapsed = dp.max_over_ranks(1.0 + rank, dist)
        lat, window = dp.gather_stream_stats([0.1 * (i + 1) for i in mine], (float(rank), 10.0 + rank), dist)
        q.put((rank, mine, elapsed, sorted(lat), window))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_dp_bookkeeping_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, mine0, e0, lat0, w0), (r1, mine1, e1, lat1, w1) = res
    assert mine0 == [0, 2, 4, 6, 8] and mine1 == [1, 3, 5, 7]
    assert e0 == e1 == 2.0                                  # max over ranks
    assert lat0 == lat1 and len(lat0) == 9                  # every request exactly once, same view on both ranks
    assert w0 == w1 == (0.0, 11.0)


def _run_bench(argv, env_extra, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_bench_gpus2_without_a_launcher_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver calls the N = 1 case) must start the 2 ranks itself -- a child
    `python -m torch.distributed.run`, the reference's one-process-per-GPU layout (executor/mp_executor.py:54-158) -- and print ONE line with
    n_gpus 2.  Dry rehearsal: gloo, the step is a sleep (no GPU in this container); the GPU-box form is tests/test_pp_gpu.py."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"MX_BENCH_REHEARSE": "dry"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "gloo" and "rehearsal" in out
    assert len(out["rank_devices"]) == 2


@pytest.mark.timeout(120)
def test_bench_refuses_a_world_smaller_than_gpus():
    """one rank answering for --gpus 2 (a launcher that started too few) is an error, not a line with n_gpus 1"""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"MX_BENCH_REHEARSE": "dry", "WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "rank(s) joined" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
