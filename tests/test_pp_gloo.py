"""Patch parallelism (BASELINE configs[3]) on CPU: world_size-2 gloo processes walk the patch-parallel step plan on the host
(mx_unet_pp_comm_plan: no launches) and replay EVERY exchange of one forward as a real gloo all-gather over a host buffer
standing in for the workspace.  Checks the buffer / offset bookkeeping that distrifuser keeps in
PatchParallelismCommManager (distrifuser/distrifuser/distrifuser/utils.py:119-214): regions inside the workspace, 16-byte
aligned, send and receive disjoint, the same sequence on every rank, and every receive region = [rank 0's send | rank 1's send]."""
import ctypes as C
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_handle():
    from oracle import sdxl_unet_ref as ref
    from sduss_amd import config, lib, weights
    l = lib.load()
    pcfg = config.UNetConfig.tiny()
    P = ref.init_params(ref.UNetConfig.tiny())
    pw = weights.PackedWeights(weights.pack(pcfg, P), "cpu")
    cc = lib.UNetConfigC()
    cc.in_channels, cc.out_channels, cc.n_levels, cc.layers_per_block = pcfg.in_channels, pcfg.out_channels, len(pcfg.block_out_channels), pcfg.layers_per_block
    for i, v in enumerate(pcfg.block_out_channels):
        cc.block_out_channels[i] = v; cc.down_has_attn[i] = int(pcfg.down_has_attn[i])
        cc.transformer_layers[i] = pcfg.transformer_layers_per_block[i]; cc.num_heads[i] = pcfg.num_heads[i]
    cc.cross_attention_dim, cc.addition_time_embed_dim = pcfg.cross_attention_dim, pcfg.addition_time_embed_dim
    cc.projection_class_embeddings_input_dim, cc.norm_num_groups = pcfg.projection_class_embeddings_input_dim, pcfg.norm_num_groups
    h = l.mx_unet_create(C.byref(cc))
    assert h and l.mx_unet_set_weights(h, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
    return l, h, pw


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sduss_amd import lib
        from sduss_amd.patch_parallel import CommLog
        l, h, _pw = _tiny_handle()
        batch, hl, w, ctx = 2, 64 // world, 64, 77
        need = l.mx_unet_workspace_bytes_pp(h, batch, hl, w, ctx, world)
        assert need > 0, l.mx_last_error()
        ws = torch.zeros(need, dtype=torch.uint8)                  # host stand-in for the device workspace
        log = CommLog()
        bad = []

        def all_gather(_ctx, _stream, send, recv, nbytes):
            so, ro = send - 0x1000, recv - 0x1000
            log.calls.append((so, ro, nbytes))
            n = len(log.calls)
            ws[so:so + nbytes] = (rank * 97 + n) % 251              # this rank's payload for exchange n
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, ws[so:so + nbytes].clone())
            ws[ro:ro + nbytes * world] = torch.cat(parts)
            for r in range(world):
                if not bool((ws[ro + r * nbytes:ro + (r + 1) * nbytes] == (r * 97 + n) % 251).all()):
                    bad.append((n, r))
            return 0
        cb = lib.ALLGATHER_FN(all_gather)
        comm = lib.PPComm(rank, world, cb, None)
        rc = l.mx_unet_pp_comm_plan(h, batch, hl, w, ctx, C.byref(comm))
        assert rc == 0, l.mx_last_error()
        log.check(need, world)
        logs = [None] * world
        dist.all_gather_object(logs, log.calls)
        # stale-asynchronous mode: the exchanges are dealt into <= 8 chunks laid out [world][sum of the chunk's 256-byte-aligned slots]
        # (pp_exchange.h), so the state is world x the aligned bytes of all exchanges
        state_need = l.mx_unet_pp_state_bytes(h, batch, hl, w, ctx, world)
        state_want = world * sum((nb + 255) // 256 * 256 for _s, _r, nb in log.calls) + 256
        q.put((rank, len(log.calls), bad, logs[0] == logs[1], sum(nb for _s, _r, nb in log.calls), need, state_need == state_want))
        l.mx_unet_destroy(h)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_pp_exchange_bookkeeping_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ncalls, bad, same, total, need, state_ok in res:
        assert state_ok, "mx_unet_pp_state_bytes must be world x the exchanges' (aligned) send sizes"
        # tiny config: 8 resnets x 2 (GroupNorm sums + conv halo) x 2 + transformer norms + K / V^T per layer + conv_in / down / up / out
        assert ncalls > 40, ncalls
        assert bad == [], f"rank {rank}: wrong bytes after exchanges {bad[:4]}"
        assert same, "the ranks must issue the same sequence of exchanges"
        assert 0 < total < 64 * need
    assert res[0][1] == res[1][1]


def test_pp_plan_rejects_bad_geometry():
    from sduss_amd import lib
    l, h, _pw = _tiny_handle()
    # 8 local rows -> 2 at the deepest level: fine for the convs, but 2 x 16 tokens per image is not a multiple of 64 for attention
    assert l.mx_unet_workspace_bytes_pp(h, 2, 8, 64, 77, 8) == 0 and b"multiple of 64" in l.mx_last_error()
    assert l.mx_unet_workspace_bytes_pp(h, 2, 6, 64, 77, 2) == 0            # local rows not divisible by 2^(levels-1)
    l.mx_unet_destroy(h)


def test_mmdit_pp_plan_and_state_size_on_host():
    """the SD3 plan's exchanges, walked on the host: two per joint block (q|k rows, V^T), two more per dual block; state = their receive sizes"""
    from sduss_amd import config, lib
    from sduss_amd.transformer_sd3 import mmdit_config_c
    l = lib.load()
    pcfg = config.MMDiTConfig.tiny()
    h = l.mx_mmdit_create(C.byref(mmdit_config_c(pcfg)))
    assert h
    world, B, Hl, W, Lt = 2, 2, 16, 32, 77
    calls = []

    def ag(_ctx, _stream, send, recv, nbytes):
        calls.append((send - 0x1000, recv - 0x1000, nbytes)); return 0
    cb = lib.ALLGATHER_FN(ag)
    comm = lib.PPComm(1, world, cb, None)
    assert l.mx_mmdit_pp_comm_plan(h, B, Hl, W, Lt, C.byref(comm)) == 0, l.mx_last_error()
    d = pcfg.num_attention_heads * 64
    L = (Hl // pcfg.patch_size) * (W // pcfg.patch_size)
    n_dual = len(pcfg.dual_attention_layers)
    assert len(calls) == 2 * pcfg.num_layers + 2 * n_dual
    assert calls[0][2] == B * L * d * 2 and calls[1][2] == B * d * L * 2     # K rows, V^T columns of the local image tokens: nothing else travels
    need = l.mx_mmdit_workspace_bytes_pp(h, B, Hl, W, Lt, world)
    from sduss_amd.patch_parallel import CommLog
    log = CommLog(); log.calls = calls
    log.check(need, world)
    want = world * sum((nb + 255) // 256 * 256 for _s, _r, nb in calls) + 256
    assert l.mx_mmdit_pp_state_bytes(h, B, Hl, W, Lt, world) == want
    assert l.mx_mmdit_workspace_bytes_pp(h, B, 2, 4, Lt, world) == 0 and b"multiple of 16" in l.mx_last_error()
    l.mx_mmdit_destroy(h)


def _base_handle():
    """SDXL-base geometry without weights: the comm-plan walk only needs the config"""
    from sduss_amd import config, lib
    l = lib.load()
    pcfg = config.UNetConfig.sdxl_base()
    cc = lib.UNetConfigC()
    cc.in_channels, cc.out_channels, cc.n_levels, cc.layers_per_block = pcfg.in_channels, pcfg.out_channels, len(pcfg.block_out_channels), pcfg.layers_per_block
    for i, v in enumerate(pcfg.block_out_channels):
        cc.block_out_channels[i] = v; cc.down_has_attn[i] = int(pcfg.down_has_attn[i])
        cc.transformer_layers[i] = pcfg.transformer_layers_per_block[i]; cc.num_heads[i] = pcfg.num_heads[i]
    cc.cross_attention_dim, cc.addition_time_embed_dim = pcfg.cross_attention_dim, pcfg.addition_time_embed_dim
    cc.projection_class_embeddings_input_dim, cc.norm_num_groups = pcfg.projection_class_embeddings_input_dim, pcfg.norm_num_groups
    return l, l.mx_unet_create(C.byref(cc))


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sduss_amd import lib
        l, h = _base_handle()
        batch, hl, w, ctx = 2, 128 // world, 128, 77               # BASELINE configs[3]: one 1024 px request (CFG batch 2), rows over 8 ranks
        need = l.mx_unet_workspace_bytes_pp(h, batch, hl, w, ctx, world)
        assert need > 0, l.mx_last_error()
        calls, bad = [], []
        head = 4096                                                # bytes of every slot that really travel (the whole slot when it is smaller)

        def all_gather(_ctx, _stream, send, recv, nbytes):
            so, ro = send - 0x1000, recv - 0x1000
            calls.append((so, ro, nbytes))
            n = len(calls)
            k = min(nbytes, head)
            mine = torch.full((k,), (rank * 31 + n) % 251, dtype=torch.uint8)
            parts = [torch.empty(k, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, mine)
            for r in range(world):
                if not bool((parts[r] == (r * 31 + n) % 251).all()):
                    bad.append((n, r))
            return 0
        cb = lib.ALLGATHER_FN(all_gather)
        comm = lib.PPComm(rank, world, cb, None)
        assert l.mx_unet_pp_comm_plan(h, batch, hl, w, ctx, C.byref(comm)) == 0, l.mx_last_error()
        from sduss_amd.patch_parallel import CommLog
        log = CommLog(); log.calls = calls
        log.check(need, world)
        logs = [None] * world
        dist.all_gather_object(logs, [(nb) for _s, _r, nb in calls])
        state = l.mx_unet_pp_state_bytes(h, batch, hl, w, ctx, world)
        q.put((rank, len(calls), sum(nb for _s, _r, nb in calls), need, state, bad, all(x == logs[0] for x in logs)))
        l.mx_unet_destroy(h)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_pp_plan_world8_sdxl_base_1024_gloo():
    """BASELINE configs[3] at its real size, walked on the host by EIGHT gloo ranks: SDXL-base widths, one 1024 px request under CFG, 16 latent
    rows per rank.  Every exchange of a forward is replayed as a real gloo all-gather of (the head of) every rank's slot; all ranks issue the same
    sequence, the regions lie inside the workspace, and a stale step ships the per-rank bytes below in <= 8 coalesced collectives (pp_exchange.h)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=280) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    ncalls, sent, need, state = res[0][1:5]
    print(f"SDXL-base 1024 px, world 8, batch 2: {ncalls} exchanges per forward, {sent / 1e6:.1f} MB sent per rank per step "
          f"({sent * (world - 1) / 1e6:.0f} MB received), workspace {need / 1e6:.0f} MB, stale state {state / 1e6:.0f} MB")
    for rank, n, s_, nd, st, bad, same in res:
        assert (n, s_, nd, st) == (ncalls, sent, need, state) and bad == [] and same
    assert 200 <= ncalls < 1000 and 20e6 < sent < 200e6
    assert state >= world * sent and state <= world * (sent + 256 * ncalls) + 256
