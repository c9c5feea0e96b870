"""Patch parallelism on the GPU: two ranks (both on cuda:0 of the one-GPU test box, exchanging through gloo) run
mx_unet_forward_pp on the row halves of a latent; their gathered output must equal the single-rank forward on the whole latent.
The arithmetic is the same real-number computation (same conv taps, same 64-key attention tiles in the same order), but not the same
rounding sequence: half the rows means other GEMM tile shapes, and the GroupNorm statistics are summed per rank first.  A flipped bf16
rounding early in the network spreads, so the two outputs differ at the network's bf16 noise floor -- the same size as either path's
distance to the fp32 oracle (measured: 1.1 % of the output range for both).  Bound: 3 % of range between the two, 4 % to the oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
        from oracle import sdxl_unet_ref as ref
        from sduss_amd.config import UNetConfig
        from sduss_amd.patch_parallel import CommLog, PatchParallelUNet
        from sduss_amd.unet import MxUNet
        ocfg = ref.UNetConfig.tiny()
        P = ref.init_params(ocfg)
        net = MxUNet(UNetConfig.tiny(), P, device="cuda:0")
        s, t, e, te, ti = ref.make_inputs(ocfg, 2, 64)
        x = s.cuda().to(torch.bfloat16)
        log = CommLog()
        pp = PatchParallelUNet(net, log=log)
        got = pp.forward(x, t.cuda(), e.cuda(), te.cuda(), ti.cuda())
        torch.cuda.synchronize()
        log.check(pp._ws.numel(), world)
        res = None
        if rank == 0:
            want = net.forward_one(x, t.cuda(), e.cuda(), te.cuda(), ti.cuda()).float()
            oracle = ref.unet_forward(P, ocfg, s, t, e, te, ti)
            d = (got.float() - want).abs()
            res = (float(d.max()), float(want.abs().max()), float((d > 0).float().mean()), float((got.float().cpu() - oracle).abs().max()),
                   float(oracle.abs().max()), len(log.calls))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank(cuda_device):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dmax, scale, frac, oerr, oscale, ncalls = res[0]
    print(f"patch-parallel x2 vs single rank: max diff {dmax:.5f} ({dmax / scale:.5f} of range), {100 * frac:.2f} % of elements differ; "
          f"vs oracle {oerr / oscale:.4f} of range; {ncalls} exchanges per forward")
    assert dmax <= 0.03 * scale, f"2 ranks differ from 1 rank by {dmax} (range {scale})"
    assert oerr <= 0.04 * oscale
    assert ncalls > 40
