"""Block-skip cache (mx_unet_forward_cached; the reference's CacheManager, modules/cache_manager.py:101-161) on the GPU.

The cache is approximate by design, so parity is stated through what must hold exactly:
  * a predictor that always answers "run" gives mx_unet_forward's output bit for bit (the block bodies are the same launches);
  * a block that is reused returns exactly the tensors it produced at its last run -- with every block reused, the step's output
    equals the previous step's output bit for bit even though the latents differ;
  * the forced run after four reuses, the per-block decisions and the invalidation by batch key follow cache_manager.py:128-136;
  * the input differences the predictor sees equal torch's MSELoss on the same tensors (first block: the conv_in output is not
    observable, so the check uses the property mse(x, x) == 0 and the uncached marker)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sdxl_unet_ref as ref  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def tiny(cuda_device):
    from sduss_amd.config import UNetConfig
    from sduss_amd.unet import MxUNet
    ocfg = ref.UNetConfig.tiny()
    P = ref.init_params(ocfg)
    return ocfg, MxUNet(UNetConfig.tiny(), P, device="cuda:0")


class Always:
    def __init__(self, v):
        self.v, self.rows = v, []

    def predict(self, f):
        self.rows.append(np.array(f))
        return np.full(len(f), self.v)


def _inputs(ocfg, batch, hw, seed=0):
    s, t, e, te, ti = ref.make_inputs(ocfg, batch, hw)
    g = torch.Generator().manual_seed(100 + seed)
    s = s + 0.05 * seed * torch.randn(s.shape, generator=g)
    return s.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), te.cuda(), ti.cuda()


@pytest.mark.parametrize("batch,hw,gn_patch", [(2, 32, 0), (3, 24, 0), (2, 32, 16)])
def test_always_run_is_the_exact_forward(tiny, batch, hw, gn_patch):
    from sduss_amd.block_cache import BlockSkipCache, MSE_UNCACHED
    ocfg, net = tiny
    pred = Always(1)
    bc = BlockSkipCache(pred)
    for step in range(3):
        s, t, e, te, ti = _inputs(ocfg, batch, hw, step)
        want = net.forward_one(s, t, e, te, ti, gn_patch=gn_patch)
        got = net.forward_one_cached(bc, s, t, e, te, ti, batch_key=7, gn_patch=gn_patch)
        assert torch.equal(got, want), f"step {step}: cached path with every block run differs from mx_unet_forward"
        assert bc.history[-1] == 0x7f
    # seven feature matrices per step; first step uncached, later steps carry finite differences; up blocks have 1 + 3 differences
    assert len(pred.rows) == 21
    assert all((r[:, 2:] >= MSE_UNCACHED * 0.5).all() for r in pred.rows[:7])
    assert all(np.isfinite(r).all() and (r[:, 2:] < 1e6).all() for r in pred.rows[7:])
    assert [r.shape[1] for r in pred.rows[:7]] == [3, 3, 3, 3, 6, 6, 6]
    assert [int(r[0, 0]) for r in pred.rows[:7]] == list(range(7))
    assert np.allclose(pred.rows[0][:, 1], t.float().cpu().numpy())


def test_first_block_difference_is_the_mse_of_its_input(tiny):
    """down block 0 sees conv_in(latents): identical latents give exactly 0, and scaling the change scales the mse by its square."""
    from sduss_amd.block_cache import BlockSkipCache
    ocfg, net = tiny
    pred = Always(1)
    bc = BlockSkipCache(pred)
    s, t, e, te, ti = _inputs(ocfg, 2, 32)
    d = torch.randn_like(s.float())
    d[1] = 0                                              # the second sample does not change
    net.forward_one_cached(bc, s.float(), t, e, te, ti, batch_key=1)
    net.forward_one_cached(bc, s.float() + 0.25 * d, t, e, te, ti, batch_key=1)
    m1 = pred.rows[7][:, 2].copy()
    net.forward_one_cached(bc, s.float(), t, e, te, ti, batch_key=1)      # back: the same distance again
    m2 = pred.rows[14][:, 2].copy()
    net.forward_one_cached(bc, s.float(), t, e, te, ti, batch_key=1)      # unchanged input
    m3 = pred.rows[21][:, 2].copy()
    assert m1[0] > 0 and m1[1] == 0.0 and m3.tolist() == [0.0, 0.0]
    assert abs(m2[0] - m1[0]) <= 1e-3 * m1[0]
    # conv_in is linear and the bias cancels in the difference: mse == mean(conv(0.25 d)^2), bf16 rounding of the two stored inputs aside
    import torch.nn.functional as F
    P = ref.init_params(ocfg)
    y = F.conv2d(0.25 * d[:1].cpu(), P["conv_in.weight"].float(), None, padding=1)
    want = float((y ** 2).mean())
    assert abs(m1[0] - want) <= 0.05 * want, (m1[0], want)


def test_all_blocks_reused_returns_the_previous_output(tiny):
    from sduss_amd.block_cache import BlockSkipCache
    ocfg, net = tiny
    pred = Always(1)
    bc = BlockSkipCache(pred)
    s0, t, e, te, ti = _inputs(ocfg, 2, 32, 0)
    out0 = net.forward_one_cached(bc, s0, t, e, te, ti, batch_key=3)
    pred.v = 0
    outs = []
    for step in range(1, 7):
        s, *_ = _inputs(ocfg, 2, 32, step)
        outs.append(net.forward_one_cached(bc, s, t, e, te, ti, batch_key=3))
    # four reuses, the forced run at the fifth call (cache_manager.py:134), then reuse again
    assert [h for h in bc.history] == [0x7f, 0, 0, 0, 0, 0x7f, 0]
    for k in range(4):
        assert torch.equal(outs[k], out0)
    s5, *_ = _inputs(ocfg, 2, 32, 5)
    assert torch.equal(outs[4], net.forward_one(s5, t, e, te, ti))
    assert torch.equal(outs[5], outs[4])


def test_partial_reuse_recomputes_only_what_was_asked(tiny):
    """up blocks reused, down and mid run: the output is the up path's cached output; a block runs when ANY sample of the batch asks, and
    inside a running block the samples that did not ask keep their cached outputs (update_and_return, cache_manager.py:84-99): with only
    the last sample asking for the up blocks, that sample gets the exact forward and the other one its cached up path."""
    from sduss_amd.block_cache import BlockSkipCache
    ocfg, net = tiny

    class Split:
        def __init__(self):
            self.run_up = 1
            self.one_sample = False

        def predict(self, f):
            f = np.asarray(f)
            if f.shape[1] == 3:
                return np.ones(len(f))
            if self.one_sample:
                m = np.zeros(len(f)); m[-1] = 1
                return m
            return np.full(len(f), self.run_up)
    pred = Split()
    bc = BlockSkipCache(pred)
    s0, t, e, te, ti = _inputs(ocfg, 2, 32, 0)
    out0 = net.forward_one_cached(bc, s0, t, e, te, ti, batch_key=9)
    pred.run_up = 0
    s1, *_ = _inputs(ocfg, 2, 32, 1)
    out1 = net.forward_one_cached(bc, s1, t, e, te, ti, batch_key=9)
    assert bc.history[-1] == 0x0f and torch.equal(out1, out0)
    pred.one_sample = True
    out2 = net.forward_one_cached(bc, s1, t, e, te, ti, batch_key=9)
    exact = net.forward_one(s1, t, e, te, ti)
    assert bc.history[-1] == 0x7f and torch.equal(out2[1], exact[1]) and torch.equal(out2[0], out0[0]) and not torch.equal(out2[0], exact[0])
    # only the last up block reused: its cached output is the hidden state conv_out sees -> the output of the step before
    class LastUp(Split):
        def predict(self, f):
            f = np.asarray(f)
            return np.zeros(len(f)) if int(f[0, 0]) == 6 else np.ones(len(f))
    bc2 = BlockSkipCache(LastUp())
    a = net.forward_one_cached(bc2, s0, t, e, te, ti, batch_key=9)
    b = net.forward_one_cached(bc2, s1, t, e, te, ti, batch_key=9)
    assert bc2.history == [0x7f, 0x3f] and torch.equal(a, b)
    # only the mid block reused: everything downstream is recomputed from the cached mid output and the fresh skips -> neither step's output
    class Mid(Split):
        def predict(self, f):
            f = np.asarray(f)
            return np.zeros(len(f)) if int(f[0, 0]) == 3 else np.ones(len(f))
    bc3 = BlockSkipCache(Mid())
    a = net.forward_one_cached(bc3, s0, t, e, te, ti, batch_key=9)
    b = net.forward_one_cached(bc3, s1, t, e, te, ti, batch_key=9)
    exact = net.forward_one(s1, t, e, te, ti)
    assert bc3.history == [0x7f, 0x77] and not torch.equal(b, a) and not torch.equal(b, exact)
    rel = ((b.float() - exact.float()).norm() / exact.float().norm()).item()
    assert rel < 0.5, rel                        # an approximation of the exact step, not garbage


def test_batch_key_and_shape_invalidate(tiny):
    from sduss_amd.block_cache import BlockSkipCache
    ocfg, net = tiny
    pred = Always(0)
    bc = BlockSkipCache(pred)
    s0, t, e, te, ti = _inputs(ocfg, 2, 32, 0)
    net.forward_one_cached(bc, s0, t, e, te, ti, batch_key=1)
    s1, *_ = _inputs(ocfg, 2, 32, 1)
    net.forward_one_cached(bc, s1, t, e, te, ti, batch_key=1)
    got = net.forward_one_cached(bc, s1, t, e, te, ti, batch_key=2)           # another batch composition: nothing may be reused
    assert bc.history == [0x7f, 0, 0x7f] and torch.equal(got, net.forward_one(s1, t, e, te, ti))
    s3, t3, e3, te3, ti3 = _inputs(ocfg, 3, 24, 0)
    got = net.forward_one_cached(bc, s3, t3, e3, te3, ti3, batch_key=2)       # another shape
    assert bc.history[-1] == 0x7f and torch.equal(got, net.forward_one(s3, t3, e3, te3, ti3))


def test_predictor_failure_and_bad_arguments_are_reported(tiny):
    from sduss_amd import lib
    from sduss_amd.block_cache import BlockSkipCache
    ocfg, net = tiny

    class Broken:
        def predict(self, f):
            raise ValueError("predictor file missing")
    s0, t, e, te, ti = _inputs(ocfg, 2, 32, 0)
    with pytest.raises(ValueError, match="predictor file missing"):
        net.forward_one_cached(BlockSkipCache(Broken()), s0, t, e, te, ti, batch_key=1)
    bc = BlockSkipCache(Always(1))
    net.forward_one_cached(bc, s0, t, e, te, ti, batch_key=1)
    bc.desc.state_bytes = 4096
    l = lib.load()
    ws = net._workspace(2, 32, 32, e.shape[1], int(lib.current_stream() or 0))
    out = torch.empty_like(s0)
    ts = t.float().reshape(-1).expand(2).contiguous() if t.numel() == 1 else t.float().contiguous()
    import ctypes as C
    rc = l.mx_unet_forward_cached(net._handle, lib.current_stream(), s0.data_ptr(), lib.torch_dtype_code(s0.dtype), ts.data_ptr(),
                                  e.to(torch.bfloat16).contiguous().data_ptr(), te.to(torch.bfloat16).contiguous().data_ptr(),
                                  ti.float().contiguous().data_ptr(), out.data_ptr(), 2, 32, 32, e.shape[1], 0, ws.data_ptr(), ws.numel(),
                                  C.byref(bc.desc))
    assert rc != 0 and b"state buffer too small" in l.mx_last_error() and bc.desc.cached_valid == 0
    torch.cuda.synchronize()


# ---- SD3 / SD3.5 transformer: one cache point per joint block (SD3Transformer.py:151-228, cache_manager.py:163-191) ----
from oracle import sd3_mmdit_ref as mmdit_ref  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def tiny_sd3(cuda_device):
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    ocfg = mmdit_ref.MMDiTConfig.tiny()
    return ocfg, MxSD3Transformer(MMDiTConfig.tiny(), mmdit_ref.init_params(ocfg), device="cuda:0")


def _sd3_inputs(ocfg, batch, hw, lt, seed=0):
    lat, t, e, p = mmdit_ref.make_inputs(ocfg, batch, hw, ctx_len=lt)
    g = torch.Generator().manual_seed(200 + seed)
    lat = lat + 0.05 * seed * torch.randn(lat.shape, generator=g)
    return lat.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), p.cuda()


@pytest.mark.parametrize("batch,hw,lt", [(2, 16, 37), (3, 24, 77)])
def test_sd3_always_run_is_the_exact_forward(tiny_sd3, batch, hw, lt):
    from sduss_amd.block_cache import BlockSkipCache, FORCED_RUN_AFTER_SD3, MSE_UNCACHED
    ocfg, net = tiny_sd3
    n = ocfg.num_layers
    pred = Always(1)
    bc = BlockSkipCache(pred, forced_after=FORCED_RUN_AFTER_SD3)
    for step in range(3):
        lat, t, e, p = _sd3_inputs(ocfg, batch, hw, lt, step)
        want = net.forward_one(lat, t, e, p)
        got = net.forward_one(lat, t, e, p, cache=bc, batch_key=5)
        assert torch.equal(got, want), f"step {step}"
        assert bc.history[-1] == (1 << n) - 1
    assert len(pred.rows) == 3 * n and all(r.shape == (batch, 3) for r in pred.rows)
    assert all((r[:, 2] >= MSE_UNCACHED * 0.5).all() for r in pred.rows[:n])
    assert all((r[:, 2] < 1e6).all() and (r[:, 2] > 0).all() for r in pred.rows[n:])
    assert [int(r[0, 0]) for r in pred.rows[:n]] == list(range(n))


def test_sd3_reuse_forced_run_after_two_and_invalidation(tiny_sd3):
    from sduss_amd.block_cache import BlockSkipCache, FORCED_RUN_AFTER_SD3
    ocfg, net = tiny_sd3
    n = ocfg.num_layers
    full = (1 << n) - 1
    pred = Always(1)
    bc = BlockSkipCache(pred, forced_after=FORCED_RUN_AFTER_SD3)
    lat0, t, e, p = _sd3_inputs(ocfg, 2, 16, 37, 0)
    out0 = net.forward_one(lat0, t, e, p, cache=bc, batch_key=1)
    pred.v = 0
    outs = []
    for step in range(1, 5):
        lat, *_ = _sd3_inputs(ocfg, 2, 16, 37, step)
        outs.append(net.forward_one(lat, t, e, p, cache=bc, batch_key=1))
    assert bc.history == [full, 0, 0, full, 0]                    # cache_manager.py:184-186: the third call runs
    assert torch.equal(outs[0], out0) and torch.equal(outs[1], out0)
    lat3, *_ = _sd3_inputs(ocfg, 2, 16, 37, 3)
    assert torch.equal(outs[2], net.forward_one(lat3, t, e, p)) and torch.equal(outs[3], outs[2])
    # the last block alone reused: norm_out / proj_out see the cached image stream -> the previous output
    class Last:
        def predict(self, f):
            f = np.asarray(f)
            return np.zeros(len(f)) if int(f[0, 0]) == n - 1 else np.ones(len(f))
    bc2 = BlockSkipCache(Last(), forced_after=FORCED_RUN_AFTER_SD3)
    a = net.forward_one(lat0, t, e, p, cache=bc2, batch_key=1)
    b = net.forward_one(lat3, t, e, p, cache=bc2, batch_key=1)
    assert bc2.history == [full, full >> 1] and torch.equal(a, b)
    # a middle block reused: both streams continue from its cached outputs -> an approximation of the exact step
    class Mid:
        def predict(self, f):
            f = np.asarray(f)
            return np.zeros(len(f)) if int(f[0, 0]) == 1 else np.ones(len(f))
    bc3 = BlockSkipCache(Mid(), forced_after=FORCED_RUN_AFTER_SD3)
    net.forward_one(lat0, t, e, p, cache=bc3, batch_key=1)
    b = net.forward_one(lat3, t, e, p, cache=bc3, batch_key=1)
    exact = net.forward_one(lat3, t, e, p)
    assert bc3.history[-1] == full & ~2 and not torch.equal(b, exact)
    assert ((b.float() - exact.float()).norm() / exact.float().norm()).item() < 0.5
    got = net.forward_one(lat3, t, e, p, cache=bc3, batch_key=2)   # another batch composition
    assert bc3.history[-1] == full and torch.equal(got, exact)


def test_model_slot_forward_with_the_cache_enabled(tiny):
    """enable_block_cache: the same forward(sample_dict, ..., input_indices) call the pipeline makes, one cache state per resolution; the
    whole denoising step (CFG rows, Euler update) runs through it."""
    from sduss_amd.pipeline import SDXLDenoiser, synthetic_request
    from sduss_amd.config import UNetConfig
    ocfg, net = tiny
    cfg = UNetConfig.tiny()
    den = SDXLDenoiser(net)
    shared = {}

    def fresh():
        return {"256": [synthetic_request(i, 256, 10, cfg, den, torch.device("cuda:0"), shared=shared) for i in range(2)],
                "128": [synthetic_request(10 + i, 128, 10, cfg, den, torch.device("cuda:0"), shared=shared) for i in range(1)]}
    exact = fresh()
    net.mixed_one_sequence = False          # the cache entry runs one launch sequence per resolution (one cache state each): compare like with like
    for _ in range(3):
        den.denoising_step(exact)
    net.mixed_one_sequence = True
    pred = Always(1)
    net.enable_block_cache(pred)
    try:
        got = fresh()
        for _ in range(3):
            den.denoising_step(got)
        for res in exact:
            for a, b in zip(exact[res], got[res]):
                assert torch.equal(a.latents, b.latents)
        assert sorted(net._block_caches) == ["128", "256"] and all(c.history == [0x7f] * 3 for c in net._block_caches.values())
        # reuse everything from here on: the noise prediction repeats, the latents still move by the Euler step
        pred.v = 0
        before = [r.latents.clone() for r in got["256"]]
        den.denoising_step(got)
        assert all(c.history[-1] == 0 for c in net._block_caches.values())
        assert all(not torch.equal(a, r.latents) for a, r in zip(before, got["256"]))
        # a request leaves the 256 px batch: the one that stays keeps its cached tensors (the reference's dictionaries are keyed by request id)
        got["256"] = got["256"][:1]
        den.denoising_step(got)
        assert net._block_caches["256"].history[-1] == 0 and net._block_caches["128"].history[-1] == 0
        # a new request joins it: it has nothing cached, so every block runs once, for the whole batch
        got["256"].append(synthetic_request(77, 256, 10, cfg, den, torch.device("cuda:0"), shared=shared))
        den.denoising_step(got)
        assert net._block_caches["256"].history[-1] == 0x7f and net._block_caches["128"].history[-1] == 0
    finally:
        net.disable_block_cache()
    s, t, e, te, ti = _inputs(ocfg, 2, 32)
    key = "256"
    out = net.forward({key: s}, t, e, added_cond_kwargs={"text_embeds": te, "time_ids": ti}, return_dict=False, is_sliced=False,
                      patch_size=256, input_indices={key: ["0", "1"]})[0][key]
    assert torch.equal(out, net.forward_one(s, t, e, te, ti))


def test_observer_reports_output_movement_and_fitted_forest_plugs_in(tiny):
    """observe=True: per block that ran while cached, the mean squared movement of its output (the fitting label, tools/fit_skip_predictor.py);
    a scikit-learn forest fitted on the recorded rows is a valid predictor object."""
    from sklearn.ensemble import RandomForestClassifier
    from sduss_amd.block_cache import BlockSkipCache, MSE_UNCACHED
    ocfg, net = tiny
    bc = BlockSkipCache(Always(1), observe=True)
    s0, t, e, te, ti = _inputs(ocfg, 2, 32, 0)
    s1, *_ = _inputs(ocfg, 2, 32, 1)
    net.forward_one_cached(bc, s0, t, e, te, ti, batch_key=1)
    assert bc.observed == [] and len(bc.features) == 7
    net.forward_one_cached(bc, s0, t, e, te, ti, batch_key=1)              # nothing moved
    assert [b for b, _ in bc.observed] == list(range(7)) and all((m == 0).all() for _, m in bc.observed)
    net.forward_one_cached(bc, s1, t, e, te, ti, batch_key=1)
    moved = bc.observed[7:]
    assert [b for b, _ in moved] == list(range(7)) and all((m > 0).all() and m.shape == (2,) for _, m in moved)
    # rows with a cached input pair one-to-one with the observations
    cached_rows = [f for f in bc.features if not (f[:, 2] >= MSE_UNCACHED * 0.5).any()]
    assert len(cached_rows) == len(bc.observed) == 14
    X = np.concatenate([f for f in cached_rows if f.shape[1] == 3])
    y = np.concatenate([(m > 1e-6).astype(np.int64) for f, (_, m) in zip(cached_rows, bc.observed) if f.shape[1] == 3])
    down = RandomForestClassifier(n_estimators=16, bootstrap=False, random_state=0).fit(X, y)
    Xu = np.concatenate([f for f in cached_rows if f.shape[1] == 6])
    yu = np.concatenate([(m > 1e-6).astype(np.int64) for f, (_, m) in zip(cached_rows, bc.observed) if f.shape[1] == 6])
    up = RandomForestClassifier(n_estimators=16, bootstrap=False, random_state=0).fit(Xu, yu)
    bc2 = BlockSkipCache(down, up)
    a = net.forward_one_cached(bc2, s0, t, e, te, ti, batch_key=1)
    b = net.forward_one_cached(bc2, s0, t, e, te, ti, batch_key=1)         # zero movement: the forests learnt "reuse" for it
    assert bc2.history == [0x7f, 0] and torch.equal(a, b)
    c = net.forward_one_cached(bc2, s1, t, e, te, ti, batch_key=1)
    assert bc2.history[-1] == 0x7f and torch.equal(c, net.forward_one(s1, t, e, te, ti))


def test_state_follows_the_requests_not_the_batch_positions(tiny):
    """row_ids: one state row per request.  A request that stays while others leave, join or change position keeps its cached tensors and its
    counters; what comes out of a reused block is that REQUEST's cached output wherever it now sits in the batch."""
    from sduss_amd.block_cache import BlockSkipCache
    ocfg, net = tiny
    bc = BlockSkipCache(Always(0), forced_after=1 << 30)
    s, t, e, te, ti = _inputs(ocfg, 3, 32, 0)                   # samples a, b, c
    sa, sb, sc = s[0:1], s[1:2], s[2:3]
    row = lambda *idx: (torch.cat([s[i:i + 1] for i in idx]), t[list(idx)], e[list(idx)], te[list(idx)], ti[list(idx)])
    noise = lambda x, k: (x.float() + 0.05 * torch.randn(x.shape, generator=torch.Generator().manual_seed(k)).cuda()).to(torch.bfloat16)
    # step 1: [a, b] -- nothing cached, every block runs
    x, tt, ee, tte, tti = row(0, 1)
    out1 = net.forward_one_cached(bc, x, tt, ee, tte, tti, row_ids=["a", "b"])
    assert bc.history[-1] == 0x7f and torch.equal(out1, net.forward_one(x, tt, ee, tte, tti))
    # step 2: a left; b alone with moved latents -- everything reused: b's cached output, although b now sits in position 0
    x, tt, ee, tte, tti = row(1)
    out2 = net.forward_one_cached(bc, noise(x, 1), tt, ee, tte, tti, row_ids=["b"])
    # (the layers after the last block -- conv_norm_out, conv_out -- run at batch 1 now: same arithmetic, possibly another rounding sequence)
    assert bc.history[-1] == 0 and (out2.float() - out1[1:2].float()).abs().max() <= 0.01 * out1.float().abs().max()
    # step 3: c joins -- c has nothing cached, so every block runs; c gets the exact forward, while b, which did not ask, keeps its cached
    # outputs inside the running blocks (update_and_return, cache_manager.py:84-99)
    x, tt, ee, tte, tti = row(1, 2)
    x3 = noise(x, 2)
    out3 = net.forward_one_cached(bc, x3, tt, ee, tte, tti, row_ids=["b", "c"])
    exact3 = net.forward_one(x3, tt, ee, tte, tti)
    assert bc.history[-1] == 0x7f and torch.equal(out3[1:2], exact3[1:2]) and not torch.equal(out3[0:1], exact3[0:1])
    assert (out3[0:1].float() - out2.float()).abs().max() <= 0.01 * out1.float().abs().max()
    # step 4: the two swap positions, latents move -- reused: each request's own cached output, in the new order
    x, tt, ee, tte, tti = row(2, 1)
    out4 = net.forward_one_cached(bc, noise(x, 3), tt, ee, tte, tti, row_ids=["c", "b"])
    assert bc.history[-1] == 0 and torch.equal(out4, torch.cat([out3[1:2], out3[0:1]]))
    # step 5: a comes back: it was forgotten when it left (cache_manager.py:131), so the blocks run again
    x, tt, ee, tte, tti = row(0, 2, 1)
    x5 = noise(x, 4)
    out5 = net.forward_one_cached(bc, x5, tt, ee, tte, tti, row_ids=["a", "c", "b"])
    exact5 = net.forward_one(x5, tt, ee, tte, tti)
    assert bc.history[-1] == 0x7f and torch.equal(out5[0:1], exact5[0:1])           # a: nothing cached, computed; c and b: their cached outputs
    assert (out5[1:3].float() - out4.float()).abs().max() <= 0.01 * out1.float().abs().max() and not torch.equal(out5[1:3], exact5[1:3])
    # the features the predictor saw in step 5: a uncached, c and b carry finite differences
    from sduss_amd.block_cache import MSE_UNCACHED
    last = bc.down.rows[-7]
    assert last[0, 2] >= MSE_UNCACHED * 0.5 and (last[1:, 2] < 1e6).all()


def test_reuse_counters_follow_the_requests(tiny):
    """forced run after two reuses, per request (cache_manager.py:128-136 with per-id previous_mask).  The predictor here answers "reuse" even
    for a request with nothing cached, so -- as in the reference's bookkeeping -- a request's first step already counts one; b and c then reach
    their forced runs at different steps because c joined later."""
    from sduss_amd.block_cache import BlockSkipCache
    ocfg, net = tiny
    bc = BlockSkipCache(Always(0), forced_after=2)
    s, t, e, te, ti = _inputs(ocfg, 2, 32, 0)
    one = lambda i: (s[i:i + 1], t[i:i + 1], e[i:i + 1], te[i:i + 1], ti[i:i + 1])
    net.forward_one_cached(bc, *one(0), row_ids=["b"])                       # b uncached: runs; b = 1
    net.forward_one_cached(bc, *one(0), row_ids=["b"])                       # reuse; b = 2
    net.forward_one_cached(bc, s, t, e, te, ti, row_ids=["b", "c"])          # b forced (and c uncached): runs; b = 0, c = 1
    net.forward_one_cached(bc, s, t, e, te, ti, row_ids=["b", "c"])          # reuse; b = 1, c = 2
    net.forward_one_cached(bc, s, t, e, te, ti, row_ids=["b", "c"])          # c forced: runs for the batch; c = 0, b = 2
    net.forward_one_cached(bc, s, t, e, te, ti, row_ids=["b", "c"])          # b forced: runs; b = 0, c = 1
    net.forward_one_cached(bc, s[1:2], t[1:2], e[1:2], te[1:2], ti[1:2], row_ids=["c"])   # c alone keeps its counter: reuse; c = 2
    assert bc.history == [0x7f, 0, 0x7f, 0, 0x7f, 0x7f, 0]


class Scripted:
    """a predictor that ignores its features and plays back a script of answers -- the same script drives the HIP path and the oracle"""

    def __init__(self, answers):
        self.answers, self.rows = [np.array(a) for a in answers], []

    def predict(self, f):
        self.rows.append(np.array(f))
        a = self.answers.pop(0)
        assert len(a) == len(f)
        return a.copy()


def test_cached_forward_against_the_oracle_of_the_cached_forward(tiny):
    """mx_unet_forward_cached against oracle/cache_ref.CachedUNetRef -- the reference's seven block wrappers (unet_2d_blocks.py) with its
    CacheManager semantics at the unit of the unsliced path (one patch per latent: the sample), on a SCRIPTED sequence of per-sample run /
    reuse answers over six steps while the latents move: whole blocks skipped, blocks running for SOME samples only (the others keep their
    cached outputs inside the running block: update_and_return), a request leaving and a new one joining, the forced run after four reuses.
    Checked per step: the output of every sample against the oracle's, the set of blocks that ran, and the feature rows the predictor saw --
    value by value, which also pins the column order of the up blocks' skip differences (oldest skip first, cache_manager.py:110-121)."""
    from oracle import cache_ref
    from oracle import sdxl_unet_ref as oref
    from sduss_amd.block_cache import BlockSkipCache, MSE_UNCACHED
    ocfg, net = tiny
    P = oref.init_params(ocfg)
    rng = np.random.RandomState(4)
    steps, nblk = 6, 7
    ids_per_step = [["a", "b", "c"], ["a", "b", "c"], ["a", "b", "c"], ["a", "c", "d"], ["a", "c", "d"], ["a", "c", "d"]]
    script = []
    for s_, ids in enumerate(ids_per_step):
        for b in range(nblk):
            if s_ == 0:
                script.append([1, 1, 1])                       # nothing cached yet: the reference's forests answer "run" on the MAX marker
            else:
                a = (rng.rand(3) < 0.45).astype(np.int64)
                if s_ == 3:
                    a[2] = 1                                   # "d" is new at step 3
                if s_ == 1 and b == 1:
                    a[:] = 0                                   # a whole block skipped
                if b == 6:
                    a[0] = 0                                   # request "a" never asks for the last block: forced to run it at step 5 (:134,154)
                script.append(a.tolist())
    hip_pred, ora_pred = Scripted(script), Scripted(script)
    bc = BlockSkipCache(hip_pred)
    ora = cache_ref.CachedUNetRef(P, ocfg, ora_pred)
    base = {k: oref.make_inputs(ocfg, 1, 16, seed=50 + i) for i, k in enumerate("abcd")}
    g = torch.Generator().manual_seed(9)
    worst = 0.0
    for s_, ids in enumerate(ids_per_step):
        rows = []
        for k in ids:
            smp, t, e, te, ti = base[k]
            smp = smp + 0.08 * s_ * torch.randn(smp.shape, generator=g)         # the latents move from step to step
            rows.append((smp.to(torch.bfloat16).float(), torch.full((1,), 801.0 - 40.0 * s_), e, te, ti))
        cat = [torch.cat([r[j] for r in rows]) for j in range(5)]
        with torch.inference_mode():
            want = ora.forward(ids, *cat)
        got = net.forward_one_cached(bc, cat[0].cuda().to(torch.bfloat16), cat[1].cuda(), cat[2].cuda(), cat[3].cuda(), cat[4].cuda(), row_ids=ids).float().cpu()
        assert bc.history[-1] == ora.blocks_run[-1], f"step {s_}: blocks run {bc.history[-1]:#x} vs the oracle's {ora.blocks_run[-1]:#x}"
        for i, k in enumerate(ids):
            l2 = float((got[i] - want[i]).norm() / want[i].norm())
            worst = max(worst, l2)
            assert l2 <= 0.03 and float((got[i] - want[i]).abs().max()) <= 0.06 * float(want[i].abs().max()), f"step {s_}, request {k}: rel L2 {l2:.4f}"
    print(f"cached forward vs its oracle over {steps} scripted steps: worst rel L2 {worst:.4f}; blocks run per step {[hex(h) for h in bc.history]}")
    assert any(h != 0x7f for h in bc.history[1:]) and len(hip_pred.rows) == len(ora.features) == steps * nblk
    for n, (fh, fo) in enumerate(zip(hip_pred.rows, ora.features)):
        assert fh.shape == fo.shape and np.array_equal(fh[:, 0], fo[:, 0]) and np.allclose(fh[:, 1], fo[:, 1])
        unc_h, unc_o = fh[:, 2:] >= MSE_UNCACHED * 0.5, fo[:, 2:] >= cache_ref.MAX * 0.5
        assert np.array_equal(unc_h, unc_o), f"feature row {n}: uncached markers differ"
        # the differences are taken between bf16 activations here and fp32 ones in the oracle: 10 % of each value (+ a floor for ~0 differences)
        assert np.allclose(fh[:, 2:][~unc_h], fo[:, 2:][~unc_o], rtol=0.10, atol=2e-4), f"feature row {n} (block {int(fh[0, 0])}): {fh[:, 2:]} vs {fo[:, 2:]}"


# ---------------------------------------------------------------------------------------------------------------------------------------
# the cache at the reference's own unit, the PATCH, over a mixed-resolution batch in one launch sequence (mx_unet_forward_cached_mixed)
# ---------------------------------------------------------------------------------------------------------------------------------------
class SeededMasks:
    """answers that depend only on the call number and the number of rows (never on the feature values, except the uncached marker): the same
    decisions for the HIP path and for the oracle, including whole-block skips"""

    def __init__(self, seed, p=0.5):
        self.seed, self.p, self.calls, self.rows = seed, p, 0, []

    def predict(self, f):
        f = np.asarray(f)
        self.rows.append(f.copy())
        self.calls += 1
        rng = np.random.RandomState(self.seed + self.calls)
        out = (rng.rand(len(f)) < self.p).astype(np.int64)
        if self.calls % 9 == 0:
            out[:] = 0                                      # nobody asks: the block is skipped as a whole (unless a forced run interferes)
        out[f[:, 2] > 1e18] = 1                             # nothing cached: the reference's forests answer "run" on the MAX marker
        return out


def test_patch_unit_cached_forward_all_asking_equals_the_mixed_forward(tiny):
    """with a predictor that always answers "run" the patch-unit entry computes the sliced mixed forward: every cached op takes its whole-image
    launch and the state only adds copies -- equal to mx_unet_forward_mixed up to the extra bf16 rounding where the time embedding and the
    residual are added after the op's output was stored (the reference caches conv1 / conv2 / to_out BEFORE those adds)"""
    ocfg, net = tiny
    ins = [ref.make_inputs(ocfg, b, hw, seed=70 + i) for i, (b, hw) in enumerate([(1, 16), (2, 32)])]
    cat = lambda k: torch.cat([x[k] for x in ins]).cuda()
    xs = [x[0].cuda().to(torch.bfloat16) for x in ins]
    want = net.forward_mixed(xs, cat(1), cat(2), cat(3), cat(4), gn_patch=8)
    net.enable_block_cache(Always(1))
    try:
        for _ in range(2):
            got = net.forward({"128": xs[0], "256": xs[1]}, cat(1), cat(2), added_cond_kwargs={"text_embeds": cat(3), "time_ids": cat(4)},
                              is_sliced=True, patch_size=64, input_indices={"128": ["a"], "256": ["b", "c"]})[0]
        pcache = net._patch_cache
        assert pcache.history == [0x7f, 0x7f] and pcache.patches_asked == pcache.patches_total == 2 * 7 * (4 + 2 * 16)
    finally:
        net.disable_block_cache()
    for k, w in zip(("128", "256"), want):
        l2 = float((got[k].float() - w.float()).norm() / w.float().norm())
        assert l2 <= 0.025, f"{k}: rel L2 {l2}"      # two bf16 evaluations with different rounding points (measured 1.6 %)


def test_patch_unit_cached_forward_against_its_oracle(tiny):
    """mx_unet_forward_cached_mixed against oracle/cache_patch_ref.CachedSlicedUNetRef -- the reference's cache at its own unit (is_sliced=True:
    dictionaries keyed per 256-px patch; here 64-px patches of 128 / 256 px latents, i.e. 4 and 16 patches per latent) restated literally on the
    patch pipeline: per-patch decisions, the convolutions and the two attention sub-blocks computed for the asking patches only with every other
    patch taking the op's own cached output, GroupNorm statistics / halos / LayerNorms / feed-forward on the fresh tensors of ALL patches.  Seven
    steps while the latents move, two resolutions in ONE launch sequence, random per-patch masks (the same for both sides), whole-block skips, a
    request leaving and one joining, the forced run after four reuses.  Checked per step: every request's output, the per-patch masks, the set
    of blocks that ran, and the per-patch feature rows."""
    from oracle import cache_patch_ref
    from sduss_amd.block_cache import MSE_UNCACHED
    ocfg, net = tiny
    P = ref.init_params(ocfg)
    hip_pred, ora_pred = SeededMasks(11), SeededMasks(11)
    ora = cache_patch_ref.CachedSlicedUNetRef(P, ocfg, ora_pred)
    base = {k: ref.make_inputs(ocfg, 1, hw, seed=80 + i) for i, (k, hw) in enumerate([("a", 16), ("b", 32), ("c", 32), ("d", 32), ("e", 16)])}
    comp = [{"128": ["a"], "256": ["b", "c"]}] * 3 + [{"128": ["a", "e"], "256": ["c", "d"]}] * 4
    g = torch.Generator().manual_seed(19)
    worst = 0.0
    net.enable_block_cache(hip_pred)
    try:
        for s_, ids in enumerate(comp):
            samples, rows = {}, []
            for res in ids:
                per = []
                for k in ids[res]:
                    smp, t, e, te, ti = base[k]
                    smp = (smp + 0.06 * s_ * torch.randn(smp.shape, generator=g)).to(torch.bfloat16).float()
                    per.append(smp)
                    rows.append((torch.full((1,), 801.0 - 40.0 * s_), e, te, ti))
                samples[res] = torch.cat(per)
            cat = [torch.cat([r[j] for r in rows]) for j in range(4)]
            row_ids = {res: [f"{k}#0" for k in ids[res]] for res in ids}         # what MxUNet.forward derives from input_indices (one CFG half here)
            with torch.inference_mode():
                want = ora.forward(row_ids, samples, cat[0], cat[1], cat[2], cat[3], 64)
            got = net.forward({r: samples[r].cuda().to(torch.bfloat16) for r in samples}, cat[0].cuda(), cat[1].cuda(),
                              added_cond_kwargs={"text_embeds": cat[2].cuda(), "time_ids": cat[3].cuda()}, is_sliced=True, patch_size=64,
                              input_indices=ids)[0]
            pcache = net._patch_cache
            assert pcache.history[-1] == ora.blocks_run[-1], f"step {s_}: blocks run {pcache.history[-1]:#x} vs the oracle's {ora.blocks_run[-1]:#x}"
            assert len(pcache.decisions) == len(ora.masks[-1]) == 7
            for (blk, mh), mo in zip(pcache.decisions, ora.masks[-1]):
                assert np.array_equal(mh, mo), f"step {s_}, block {blk}: per-patch masks differ"
            for res in ids:
                for i, k in enumerate(ids[res]):
                    gi, wi = got[res][i].float().cpu(), want[res][i]
                    l2 = float((gi - wi).norm() / wi.norm())
                    worst = max(worst, l2)
                    assert l2 <= 0.03 and float((gi - wi).abs().max()) <= 0.06 * float(wi.abs().max()), f"step {s_}, request {k} ({res} px): rel L2 {l2:.4f}"
        frac = pcache.patches_asked / pcache.patches_total
        print(f"patch-unit cached forward vs its oracle over {len(comp)} steps: worst rel L2 {worst:.4f}; blocks run {[hex(h) for h in pcache.history]}; "
              f"{pcache.patches_asked} of {pcache.patches_total} patch-blocks asked ({frac:.2f})")
        assert any(h != 0x7f for h in pcache.history[1:]) and 0.3 < frac < 0.9
        feats_o = [f for step in ora.features for f in step]
        assert len(hip_pred.rows) == len(feats_o)
        for n, (fh, fo) in enumerate(zip(hip_pred.rows, feats_o)):
            assert fh.shape == fo.shape and np.array_equal(fh[:, 0], fo[:, 0]) and np.allclose(fh[:, 1], fo[:, 1])
            unc_h, unc_o = fh[:, 2:] >= MSE_UNCACHED * 0.5, fo[:, 2:] >= 1e18
            assert np.array_equal(unc_h, unc_o), f"feature row {n}: uncached markers differ"
            # differences of bf16 activations here, of fp32 ones in the oracle: 12 % of a value, and a floor of 1e-3 -- the size a patch whose
            # values hardly moved reaches through the bf16 rounding of the two tensors alone
            assert np.allclose(fh[:, 2:][~unc_h], fo[:, 2:][~unc_o], rtol=0.12, atol=1e-3), f"feature rows {n} (block {int(fh[0, 0])}) differ"
    finally:
        net.disable_block_cache()


def test_mmdit_chunk_unit_cached_forward_against_its_oracle(cuda_device):
    """mx_mmdit_forward_cached_mixed against oracle/cache_patch_ref.CachedSlicedMMDiTRef: the SD3 cache at the reference's unit (token chunks keyed
    "<request id>-<k>", one cache point per joint block, forced run after two reuses), two resolutions in ONE launch sequence.  Inside a running
    block a resolution none of whose chunks asks takes the cached attention outputs (image and text side), one with any asking chunk computes its
    attention whole, and the dual blocks' image-only attn2 renews the asking chunks only when the asking ratio is <= 1/16.  Scripted per-chunk
    masks (the same on both sides) over six steps while the latents move, whole-block skips, a request leaving and one joining."""
    from oracle import cache_patch_ref, sd3_mmdit_ref as m
    from sduss_amd.block_cache import MSE_UNCACHED
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    ocfg = m.MMDiTConfig.tiny()
    P = m.init_params(ocfg)
    net = MxSD3Transformer(MMDiTConfig.tiny(), P, device="cuda:0")
    lt = 20

    class ChunkMasks(SeededMasks):
        """per call: one resolution (by row count) often silent, sometimes a single asking chunk (the sparse attn2 rule), sometimes everyone"""
        def predict(self, f):
            f = np.asarray(f)
            self.rows.append(f.copy())
            self.calls += 1
            rng = np.random.RandomState(self.seed + self.calls)
            out = (rng.rand(len(f)) < 0.35).astype(np.int64)
            mode = self.calls % 4
            if mode == 0:
                out[:] = 0
            elif mode == 1:
                out[:4] = 0                                 # the first (128 px, 4 chunks) request stays silent
            elif mode == 2:
                out[:] = 0; out[-1] = 1                     # one chunk of the last 256 px request: ratio 1/32 <= 1/16
            out[f[:, 2] > 1e18] = 1
            return out

    hip_pred, ora_pred = ChunkMasks(5), ChunkMasks(5)
    ora = cache_patch_ref.CachedSlicedMMDiTRef(P, ocfg, ora_pred)
    base = {k: m.make_inputs(ocfg, 1, hw, seed=90 + i, ctx_len=lt) for i, (k, hw) in enumerate([("a", 16), ("b", 32), ("c", 32), ("d", 32)])}
    comp = [{"128": ["a"], "256": ["b", "c"]}] * 3 + [{"128": ["a"], "256": ["c", "d"]}] * 3
    g = torch.Generator().manual_seed(29)
    worst = 0.0
    net.enable_block_cache(hip_pred)
    try:
        for s_, ids in enumerate(comp):
            lat, rows = {}, []
            for res in ids:
                per = []
                for k in ids[res]:
                    x, t, e, p_ = base[k]
                    per.append((x + 0.05 * s_ * torch.randn(x.shape, generator=g)).to(torch.bfloat16).float())
                    rows.append((torch.full((1,), 901.0 - 60.0 * s_), e, p_))
                lat[res] = torch.cat(per)
            cat = [torch.cat([r[j] for r in rows]) for j in range(3)]
            row_ids = {res: [f"{k}#0" for k in ids[res]] for res in ids}
            with torch.inference_mode():
                want = ora.forward(row_ids, lat, cat[0], cat[1], cat[2], 64)
            got = net.forward({r: lat[r].cuda().to(torch.bfloat16) for r in lat}, encoder_hidden_states=cat[1].cuda(), pooled_projections=cat[2].cuda(),
                              timestep=cat[0].cuda(), return_dict=False, is_sliced=True, patch_size=64, input_indices=ids)[0]
            pcache = net._patch_cache
            assert pcache.history[-1] == ora.blocks_run[-1], f"step {s_}: blocks run {pcache.history[-1]:#x} vs the oracle's {ora.blocks_run[-1]:#x}"
            for (blk, mh), mo in zip(pcache.decisions, ora.masks[-1]):
                assert np.array_equal(mh, mo), f"step {s_}, block {blk}: per-chunk masks differ"
            for res in ids:
                for i, k in enumerate(ids[res]):
                    gi, wi = got[res][i].float().cpu(), want[res][i]
                    l2 = float((gi - wi).norm() / wi.norm())
                    worst = max(worst, l2)
                    assert l2 <= 0.03 and float((gi - wi).abs().max()) <= 0.06 * float(wi.abs().max()), f"step {s_}, request {k} ({res} px): rel L2 {l2:.4f}"
        print(f"MMDiT chunk-unit cached forward vs its oracle over {len(comp)} steps: worst rel L2 {worst:.4f}; blocks run {[hex(h) for h in pcache.history]}; "
              f"{pcache.patches_asked} of {pcache.patches_total} chunk-blocks asked")
        assert any(h != (1 << ocfg.num_layers) - 1 for h in pcache.history[1:])
        feats_o = [f for step in ora.features for f in step]
        assert len(hip_pred.rows) == len(feats_o)
        for n, (fh, fo) in enumerate(zip(hip_pred.rows, feats_o)):
            unc_h, unc_o = fh[:, 2:] >= MSE_UNCACHED * 0.5, fo[:, 2:] >= 1e18
            assert fh.shape == fo.shape and np.array_equal(unc_h, unc_o)
            assert np.allclose(fh[:, 2:][~unc_h], fo[:, 2:][~unc_o], rtol=0.12, atol=1e-3), f"feature rows {n} differ"
    finally:
        net.disable_block_cache()
