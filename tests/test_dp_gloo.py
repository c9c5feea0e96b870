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
