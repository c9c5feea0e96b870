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


def _join(rank, world, port, backend):
    """gloo: every rank on cuda:0 of the one-GPU test box, exchanging through host memory.  nccl (= RCCL): one device per rank, device to device --
    the branch of patch_parallel.py the one-GPU pool never runs (tests *_rccl below: skipped unless the box has >= 2 GPUs)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.set_num_threads(8)                 # several ranks share the host: 128 torch threads each made the world-4 run take 190 s
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dev


def _worker(rank, world, port, q, backend="gloo"):
    dev = _join(rank, world, port, backend)
    try:
        from oracle import sdxl_unet_ref as ref
        from sduss_amd.config import UNetConfig
        from sduss_amd.patch_parallel import CommLog, PatchParallelUNet
        from sduss_amd.unet import MxUNet
        ocfg = ref.UNetConfig.tiny()
        P = ref.init_params(ocfg)
        net = MxUNet(UNetConfig.tiny(), P, device=dev)
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


def _need_gpus(n):
    if torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs (one rank per device over RCCL); this box has {torch.cuda.device_count()}")


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank_rccl(cuda_device):
    """the same over RCCL with one device per rank: inert on the one-GPU pool, live on a multi-GPU node without anyone editing code"""
    _need_gpus(2)
    test_two_ranks_equal_one_rank(cuda_device, backend="nccl")


@pytest.mark.timeout(300)
def test_stale_async_steps_rccl(cuda_device):
    _need_gpus(2)
    test_stale_async_steps(cuda_device, backend="nccl")


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank(cuda_device, backend="gloo"):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
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


@pytest.mark.timeout(300)
def test_four_ranks_equal_one_rank(cuda_device):
    """the same with the latent rows over FOUR ranks sharing the test box's GPU (16 latent rows each: 4 x 16 = 64 tokens per image at the
    deepest attention level, the smallest legal split)"""
    world = 4
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
    print(f"patch-parallel x4 vs single rank: max diff {dmax:.5f} ({dmax / scale:.5f} of range); vs oracle {oerr / oscale:.4f} of range; {ncalls} exchanges")
    assert dmax <= 0.03 * scale and oerr <= 0.04 * oscale and ncalls > 40


def _stale_worker(rank, world, port, q, backend="gloo"):
    dev = _join(rank, world, port, backend)
    try:
        from oracle import sdxl_unet_ref as ref
        from sduss_amd import lib
        from sduss_amd.config import UNetConfig
        from sduss_amd.patch_parallel import PatchParallelUNet
        from sduss_amd.unet import MxUNet
        ocfg = ref.UNetConfig.tiny()
        net = MxUNet(UNetConfig.tiny(), ref.init_params(ocfg), device=dev)
        s, t, e, te, ti = ref.make_inputs(ocfg, 2, 64)
        x0 = s.cuda().to(torch.bfloat16)
        g = torch.Generator().manual_seed(5)
        x1 = (s + 0.1 * torch.randn(s.shape, generator=g)).cuda().to(torch.bfloat16)     # the next step's latents: a small move
        args = (t.cuda(), e.cuda(), te.cuda(), ti.cuda())
        sync = PatchParallelUNet(net)
        want0, want1 = sync.forward(x0, *args), sync.forward(x1, *args)
        res = {}
        for mode in ("stale_gn", "corrected_async_gn"):
            from sduss_amd.patch_parallel import CommLog
            slog = CommLog()
            pp = PatchParallelUNet(net, mode=mode, warmup_steps=0, log=slog)      # distrifuser: synchronous while counter <= warmup_steps -> one warm-up step
            a = pp.forward(x0, *args)                    # warm-up: synchronous, fills the state
            assert pp.last_step_mode == lib.PP_WARMUP
            n_sync = len(slog.calls)
            b = pp.forward(x0, *args)                    # stale step on unchanged inputs: what it reads stale equals what is fresh
            assert pp.last_step_mode == lib.PP_STALE
            stale_calls = slog.calls[n_sync:]
            # a stale step coalesces its exchanges: <= 8 in-place asynchronous all-gathers over chunks of the state (pp_exchange.h; distrifuser
            # flushes <= 60 tensors per all_gather, utils.py:184-205), no synchronous one
            assert all(so == -1 for so, _ro, _nb in stale_calls) and 1 <= len(stale_calls) <= 8 < n_sync, (len(stale_calls), n_sync)
            c = pp.forward(x1, *args)                    # stale step on moved inputs: the other rank's rows lag one step
            d = pp.forward(x1, *args)                    # the inputs stop moving: the lag is gone one step later
            pp.reset()
            torch.cuda.synchronize()
            nrm = float(want1.float().norm())
            l2 = lambda u, v: float((u.float() - v.float()).norm()) / nrm
            res[mode] = (bool(torch.equal(a, want0)), bool(torch.equal(b, want0)), l2(c, want1), l2(c, want0), l2(d, want1), l2(want1, want0))
        # warm-up length as distrifuser (modules/pp/conv2d.py:97: `counter <= warmup_steps`): warmup_steps + 1 synchronous steps; and a change of
        # the problem shape -- even to a smaller footprint -- restarts the warm-up on a fresh state layout instead of reading the old one
        pp = PatchParallelUNet(net, mode="stale_gn", warmup_steps=1)
        modes = []
        for _ in range(3):
            pp.forward(x0, *args); modes.append(pp.last_step_mode)
        pp.forward(x0[:1], args[0][:1], args[1][:1], args[2][:1], args[3][:1]); modes.append(pp.last_step_mode)
        pp.reset()
        res["modes"] = modes
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_stale_async_steps(cuda_device, backend="gloo"):
    """mx_unet_forward_pp_stale: a warm-up step is the synchronous step; a stale step with unchanged inputs reproduces it bit for bit
    (every stale slot equals the fresh one); with moved inputs it lands between: closer to the synchronous result of the new inputs than the
    old output is, and one more step on the same inputs removes most of the lag (only second-order staleness inside the network is left)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stale_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from sduss_amd import lib
    for rank in range(world):
        # warmup_steps = 1 -> two synchronous steps, then stale; a new problem shape (batch 1, a smaller footprint) restarts the warm-up
        assert res[rank].pop("modes") == [lib.PP_WARMUP, lib.PP_WARMUP, lib.PP_STALE, lib.PP_WARMUP]
        for mode, (warm_eq, same_eq, lag, old, settled, moved) in res[rank].items():
            print(f"rank {rank} {mode}: relative L2 of the stale step to the synchronous result {lag:.4f} (to the old output {old:.4f}; the inputs "
                  f"moved the synchronous output by {moved:.4f}); one more step on the same inputs {settled:.4f}")
            assert warm_eq, f"{mode}: the warm-up step must equal the synchronous step"
            assert same_eq, f"{mode}: a stale step on unchanged inputs must equal the synchronous step"
            assert lag < 0.75 * moved, f"{mode}: stale step too far from the synchronous result (rel L2 {lag}; the inputs moved it {moved})"
            assert lag < old, f"{mode}: the stale step must be nearer the new synchronous result than the old one"
            assert settled < lag, f"{mode}: the lag must shrink once the inputs stop moving"


def _sd3_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(8)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import sd3_mmdit_ref as ref
        from sduss_amd import lib
        from sduss_amd.config import MMDiTConfig
        from sduss_amd.patch_parallel import CommLog, PatchParallelSD3
        from sduss_amd.transformer_sd3 import MxSD3Transformer
        ocfg = ref.MMDiTConfig.tiny()
        P = ref.init_params(ocfg)
        net = MxSD3Transformer(MMDiTConfig.tiny(), P, device="cuda:0")
        lat, t, e, p = ref.make_inputs(ocfg, 2, 32, ctx_len=77)
        x0 = lat.cuda().to(torch.bfloat16)
        g = torch.Generator().manual_seed(5)
        x1 = (lat + 0.1 * torch.randn(lat.shape, generator=g)).cuda().to(torch.bfloat16)
        args = (t.cuda(), e.cuda(), p.cuda())
        log = CommLog()
        sync = PatchParallelSD3(net, log=log)
        got0 = sync.forward(x0, *args)
        ncalls = len(log.calls)
        log.check(sync._ws.numel(), world)
        got1 = sync.forward(x1, *args)
        want0, want1 = net.forward_one(x0, *args), net.forward_one(x1, *args)
        oracle = ref.mmdit_forward(P, ocfg, lat, t, e, p)
        pp = PatchParallelSD3(net, mode="stale_gn", warmup_steps=0)
        a = pp.forward(x0, *args)
        b = pp.forward(x0, *args)
        assert pp.last_step_mode == lib.PP_STALE
        c = pp.forward(x1, *args)
        d = pp.forward(x1, *args)
        pp.reset()
        torch.cuda.synchronize()
        nrm = float(want1.float().norm())
        l2 = lambda u, v: float((u.float() - v.float()).norm()) / nrm
        rng = float(want0.float().abs().max())
        q.put((rank, dict(sync_vs_single=float((got0.float() - want0.float()).abs().max()) / rng,
                          sync_vs_single1=float((got1.float() - want1.float()).abs().max()) / rng,
                          oracle=float((got0.float().cpu() - oracle).abs().max()) / float(oracle.abs().max()), ncalls=ncalls,
                          warm_eq=bool(torch.equal(a, got0)), same_eq=bool(torch.equal(b, got0)), lag=l2(c, got1), old=l2(c, got0),
                          settled=l2(d, got1), moved=l2(got1, got0))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sd3_two_ranks_equal_one_rank_and_stale_steps(cuda_device):
    """mx_mmdit_forward_pp: token-split SD3 transformer over two ranks.  Synchronous: the gathered output against the single-rank forward
    (same arithmetic, other GEMM tiles: the bf16 noise floor) and the fp32 oracle.  Stale: as for the UNet."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sd3_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        r = res[rank]
        print(f"rank {rank} sd3 pp: {r}")
        assert r["sync_vs_single"] <= 0.03 and r["sync_vs_single1"] <= 0.03 and r["oracle"] <= 0.04
        tiny_layers = 4
        assert r["ncalls"] == 2 * tiny_layers + 2 * 2           # q|k and V^T per joint block, twice more for the two dual blocks
        assert r["warm_eq"] and r["same_eq"]
        assert r["lag"] < 0.75 * r["moved"] and r["lag"] < r["old"] and r["settled"] < r["lag"]


@pytest.mark.timeout(600)
def test_bench_gpus2_self_spawn_with_pp_leg_on_the_shared_gpu(cuda_device):
    """`python bench.py --gpus 2` with no launcher: bench.py starts its two ranks itself (a child torch.distributed.run; the pytest process is only the
    grandparent), both on cuda:0 over gloo (MX_BENCH_REHEARSE=1), walks the timed region and the configs[3] --pp leg (one 1024 px request row-split
    over the 2 ranks, 5 synchronous + stale steps) and prints ONE line with n_gpus 2 and a patch_parallel block.  The numbers of a rehearsal mean
    nothing; the control flow, the rank accounting and the pp leg's exchange accounting are what is asserted."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MX_BENCH_REHEARSE"] = "1"
    env["OMP_NUM_THREADS"] = "8"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "1", "--steps", "2", "--warmup", "1", "--pp-steps", "2",
                        "--stream-requests", "0", "--mix", "0", "--no-cpu-baseline", "--no-roofline", "--no-sd3", "--no-stages", "--no-parity"],
                       env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "gloo" and "rehearsal" in out
    pp = out["patch_parallel"]
    assert "error" not in pp, pp
    assert pp["ranks_per_request"] == 2 and pp["collectives_per_step"]["stale"] <= 9 < pp["collectives_per_step"]["sync"]
    assert pp["sent_MB_per_rank_per_step"]["stale"] > 1.0
    assert pp["rel_l2_vs_one_gpu"]["sync"] < 0.03 and pp["rel_l2_vs_one_gpu"]["stale_on_unchanged_inputs"] < 0.03
