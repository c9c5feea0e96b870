"""End-to-end parity of the denoising LOOP (north_star: "denoised latents match ... on identical seeds within a stated L-inf tolerance"):
whole 50-step SDXL / 28-step SD3 loops (tiny configs), a 30-step loop at SDXL-base width, and continuous batching (requests at different step indices with different step counts, joining and
finishing mid-run: BASELINE configs[4]) through ``SDXLDenoiser.denoising_step`` / ``SD3Denoiser.denoising_step`` against the oracle chain
(oracle/chain_ref.py), per request.

Tolerance law.  One forward of the HIP path differs from the fp32 oracle by e1 (bf16 activation storage: measured rel L2 ~0.6-0.8 % on the
tiny configs, 1.6-2.1 % at SDXL-base width at 1024 px; tests/test_unet_gpu.py).  A step adds (sigma_next - sigma) * guided noise to the latents,
so after n steps the latents carry the sum of n such errors, each weighted by its sigma decrement and amplified by the guidance scale
(u + g (t - u)); independent per-step errors add in quadrature, hence a sqrt(n) law.  The measured growth (printed by every run; the r03 numbers
are in profiles/r03_parity_loops.txt) follows it for the first ~20 steps and then SATURATES: the sigma decrements shrink towards the end of
the schedule (most of the 14.6 -> 0 range is spent in the first third of the steps), so late steps add almost nothing -- SDXL tiny 0.8 % after
one step, 1.2 % after 10, 1.2 % after 50; SDXL-base width 0.8 % / 3.0 % / 5.8 %; SD3 tiny (flow matching, uniform decrements) 0.2 % / 0.6 % / 1.0 %
after 28.  The loop is not chaotic on these models.  The bound asserted at EVERY step of every request is

    rel_L2(n)  <=  min(A * sqrt(n), C)        and        max|err|(n) <= 4 * that bound * max|oracle latents|

with (A, C) per model below, ~1.5-2x the measured curve.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import chain_ref, scheduler_ref, sd3_mmdit_ref, sdxl_unet_ref as ref  # noqa: E402  (checker only)


def _errs(got, want):
    got = got.float().cpu()
    assert torch.isfinite(got).all()
    return ((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item()


def _law(n, a, c):
    return min(a * math.sqrt(n), c)


def _log(line):
    print(line)
    path = os.environ.get("MX_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def _mirror_sdxl(r) -> chain_ref.ChainRequest:
    """CPU fp32 copy of a device Request: the same latents, embeddings and schedule tables"""
    ts, sig, _ = scheduler_ref.sdxl_euler_tables(r.num_inference_steps)
    f = lambda t: t.float().cpu()
    return chain_ref.ChainRequest(r.request_id, r.resolution, r.num_inference_steps, f(r.latents),
                                  (f(r.prompt_embeds), f(r.pooled_prompt_embeds), f(r.add_time_ids)),
                                  (f(r.negative_prompt_embeds), f(r.negative_pooled_prompt_embeds), f(r.negative_add_time_ids)), ts, sig)


def _mirror_sd3(r) -> chain_ref.ChainRequest:
    ts, sig = chain_ref.sd3_flow_tables(r.num_inference_steps)
    f = lambda t: t.float().cpu()
    return chain_ref.ChainRequest(r.request_id, r.resolution, r.num_inference_steps, f(r.latents), (f(r.prompt_embeds), f(r.pooled_prompt_embeds)),
                                  (f(r.negative_prompt_embeds), f(r.negative_pooled_prompt_embeds)), ts, sig)


@pytest.fixture(autouse=True)
def few_host_threads(request):
    """the tiny-config oracle forwards are ~100 small torch ops each: on the GPU box's 128 default threads they spend their time in thread
    wake-ups (the 59-step continuous-batching schedule took 95 s).  The full-width chain keeps the default."""
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 32 if "base_width" in request.node.name else 8))     # (full width: 128 threads took 6.3 s per CFG forward, 8 threads 2.8 s)
    yield
    torch.set_num_threads(n)


@pytest.fixture(scope="module")
def tiny_sdxl(cuda_device):
    from sduss_amd.config import UNetConfig
    from sduss_amd.unet import MxUNet
    ocfg = ref.UNetConfig.tiny()
    P = ref.init_params(ocfg)
    return ocfg, P, MxUNet(UNetConfig.tiny(), P, device="cuda:0")


@pytest.fixture(scope="module")
def tiny_sd3(cuda_device):
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    ocfg = sd3_mmdit_ref.MMDiTConfig.tiny()
    P = sd3_mmdit_ref.init_params(ocfg)
    return ocfg, P, MxSD3Transformer(MMDiTConfig.tiny(), P, device="cuda:0")


def _run_schedule(den, dev_reqs, cpu_reqs, joins, model, kind, guidance, step_kwargs, what, law):
    """Drive both sides through the same continuous-batching schedule: request i joins at global step joins[i] and leaves when it is done.
    Returns the per-request (steps, rel L2, max err) at the end; asserts the law at every step of every request."""
    a, b = law
    g = 0
    worst = 0.0
    while not all(r.done() for r in dev_reqs):
        active = [i for i, r in enumerate(dev_reqs) if joins[i] <= g and not r.done()]
        assert active, "schedule has a hole"
        d_dev, d_cpu = {}, {}
        for i in active:
            d_dev.setdefault(str(dev_reqs[i].resolution), []).append(dev_reqs[i])
            d_cpu.setdefault(str(cpu_reqs[i].resolution), []).append(cpu_reqs[i])
        den.denoising_step(d_dev, **step_kwargs)
        chain_ref.denoising_step(d_cpu, model, kind, guidance)
        for i in active:
            assert dev_reqs[i].step_index == cpu_reqs[i].step_index
            n = dev_reqs[i].step_index
            l2, mx = _errs(dev_reqs[i].latents, cpu_reqs[i].latents)
            worst = max(worst, l2 / _law(n, a, b))
            if n in (1, 2, 5, 10, 20, 30, 40, 50) or dev_reqs[i].done():
                _log(f"{what}: request {i} ({dev_reqs[i].resolution} px, {dev_reqs[i].num_inference_steps} steps) after step {n}: rel L2 {l2:.4f} max {mx:.4f}"
                     f" (bound {_law(n, a, b):.4f})")
            assert l2 <= _law(n, a, b), f"{what}: request {i} after {n} steps: rel L2 {l2:.4f} > {_law(n, a, b):.4f}"
            assert mx <= 4 * _law(n, a, b), f"{what}: request {i} after {n} steps: max err {mx:.4f} of range"
        g += 1
    _log(f"{what}: {g} global steps, worst rel L2 / bound = {worst:.3f}")
    return g


# ---------------------------------------------------------------------------------------------------------------------------------------
# (a) continuous batching: different step indices and step counts in one batch, joining and finishing mid-run (configs[4])
# ---------------------------------------------------------------------------------------------------------------------------------------
SDXL_TINY_LAW = (0.012, 0.03)       # rel L2 after n steps <= min(1.2 % sqrt(n), 3 %)    (measured: 0.83 % after 1 step, 1.37 % at the end)
SD3_TINY_LAW = (0.005, 0.025)       #                         min(0.5 % sqrt(n), 2.5 %)  (measured: 0.20 % after 1 step, 1.03 % after 28)


def test_sdxl_continuous_batching_heterogeneous_steps(tiny_sdxl):
    """Four requests with 30 / 40 / 50 / 30 steps at two resolutions: request 0 starts alone, 1 joins at global step 4, 2 at step 9 (while
    0 is at index 9), 3 at step 20; 0 finishes at step 30 while the others are mid-run.  Every step batches requests whose sigma / timestep
    come from different rows of different tables (scheduling_euler_discrete.py:171-175, 213-217); mixed resolutions run sliced as the mixed
    policies force (policy/FCFS_Mixed.py:69-70).  Oracle: the whole-image equivalent of the sliced path per resolution
    (gn_patch + corner rule, proved equal to the literal patch pipeline in tests/test_cpu.py)."""
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import SDXLDenoiser, synthetic_request
    ocfg, P, net = tiny_sdxl
    den = SDXLDenoiser(net, guidance_scale=5.0)
    spec = [(128, 30, 0), (256, 40, 4), (128, 50, 9), (256, 30, 20)]
    dev = [synthetic_request(i, res, steps, UNetConfig.tiny(), den, "cuda:0") for i, (res, steps, _j) in enumerate(spec)]
    cpu = [_mirror_sdxl(r) for r in dev]
    patch = 128
    model = lambda x, t, e, te, ti: ref.unet_forward(P, ocfg, x, t, e, te, ti, gn_patch=patch // 8, sliced_corners=True)
    with torch.inference_mode():
        g = _run_schedule(den, dev, cpu, [j for _r, _s, j in spec], model, "sdxl", 5.0, dict(is_sliced=True, patch_size=patch),
                          "sdxl continuous batching", SDXL_TINY_LAW)
    assert g == 59 and all(r.done() for r in dev)           # request 2: joins at 9, 50 steps


def test_sd3_continuous_batching_heterogeneous_steps(tiny_sd3):
    """the same schedule shape through SD3Denoiser (flow-match Euler, guidance 7): 20 / 28 / 24 steps at two resolutions"""
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request
    ocfg, P, net = tiny_sd3
    den = SD3Denoiser(net, guidance_scale=7.0)
    spec = [(128, 20, 0), (256, 28, 3), (128, 24, 7)]
    dev = [synthetic_sd3_request(i, res, steps, MMDiTConfig.tiny(), den, "cuda:0", ctx_len=21) for i, (res, steps, _j) in enumerate(spec)]
    cpu = [_mirror_sd3(r) for r in dev]
    model = lambda x, t, e, p: sd3_mmdit_ref.mmdit_forward(P, ocfg, x, t, e, p)
    with torch.inference_mode():
        _run_schedule(den, dev, cpu, [j for _r, _s, j in spec], model, "sd3", 7.0, dict(is_sliced=True, patch_size=64), "sd3 continuous batching",
                      SD3_TINY_LAW)
    assert all(r.done() for r in dev)


# ---------------------------------------------------------------------------------------------------------------------------------------
# (b) whole loops
# ---------------------------------------------------------------------------------------------------------------------------------------
def test_sdxl_full_50_step_loop_tiny(tiny_sdxl):
    """two requests, 50 steps each, 256 px (UNet batch 4 under CFG), unsliced: the reference's default loop length"""
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import SDXLDenoiser, synthetic_request
    ocfg, P, net = tiny_sdxl
    den = SDXLDenoiser(net, guidance_scale=5.0)
    dev = [synthetic_request(i, 256, 50, UNetConfig.tiny(), den, "cuda:0") for i in range(2)]
    cpu = [_mirror_sdxl(r) for r in dev]
    model = lambda x, t, e, te, ti: ref.unet_forward(P, ocfg, x, t, e, te, ti)
    with torch.inference_mode():
        assert _run_schedule(den, dev, cpu, [0, 0], model, "sdxl", 5.0, {}, "sdxl 50-step loop (tiny)", SDXL_TINY_LAW) == 50


def test_sd3_full_28_step_loop_tiny(tiny_sd3):
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request
    ocfg, P, net = tiny_sd3
    den = SD3Denoiser(net, guidance_scale=7.0)
    dev = [synthetic_sd3_request(i, 256, 28, MMDiTConfig.tiny(), den, "cuda:0", ctx_len=21) for i in range(2)]
    cpu = [_mirror_sd3(r) for r in dev]
    model = lambda x, t, e, p: sd3_mmdit_ref.mmdit_forward(P, ocfg, x, t, e, p)
    with torch.inference_mode():
        assert _run_schedule(den, dev, cpu, [0, 0], model, "sd3", 7.0, {}, "sd3 28-step loop (tiny)", SD3_TINY_LAW) == 28


# min(3.5 % sqrt(n), 10 %) for the 30-step schedule at base width.  What the bound asserts is the SATURATION level: a random-init 2.6 B-parameter
# UNet iterated on its own output amplifies any rounding difference step over step until the two trajectories are as far apart as the map lets
# them drift (measured: 5.8 % after 50 steps of the 50-step schedule, the same level after 20 steps of the 30-step one), so the early growth
# is not a sqrt(n) random walk and differs from build to build with the roundings (30-step schedule: 1.16 % / 1.8 % / 3.6 % after 1 / 2 / 4
# steps on one build of the library, 1.27 % / 2.6 % / 5.7 % after 1 / 2 / 5 on the next, whose only arithmetic change was a GELU
# approximation at the 1e-6 level).  A = 3.5 % leaves the first steps room for that; the 10 % cap is the statement.  (50-step schedule:
# 0.76 % after 1 step, 3.0 % after 10, 5.2 % after 20, 5.8 % after 50.)
SDXL_BASE_LAW = (0.035, 0.10)


def test_sdxl_base_width_30_step_loop_256px(full_width_sdxl):
    """SDXL-base widths (2.57 B parameters) on one 256 px request, the whole loop of a 30-step request (the shortest of the reference traces'
    step counts, exp/sdxl/qps_*.csv; the fp32 oracle costs seconds per step at this width) under CFG against the oracle chain on the ORIGINAL
    weights (the LayerNorm fold's extra weight rounding is inside the bound).  The 50-step loop runs on the tiny config above; the measured
    growth at this width over 50 steps is in profiles/r03_parity_loops.txt (saturates at 5.8 % after ~30 steps)."""
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import SDXLDenoiser, synthetic_request
    ocfg, P, _held, net = full_width_sdxl
    den = SDXLDenoiser(net, guidance_scale=5.0)
    dev = [synthetic_request(0, 256, 30, UNetConfig.sdxl_base(), den, "cuda:0")]
    cpu = [_mirror_sdxl(r) for r in dev]
    P32 = {k: v.float() for k, v in P.items()}
    model = lambda x, t, e, te, ti: ref.unet_forward(P32, ocfg, x, t, e, te, ti)
    with torch.inference_mode():
        assert _run_schedule(den, dev, cpu, [0], model, "sdxl", 5.0, {}, "sdxl-base 30-step loop at 256 px", SDXL_BASE_LAW) == 30


LOOP_CALIBRATION_RATIO = 1.5     # HIP loop error <= this x the error of the SAME loop with the oracle's graph evaluated by stock torch ops in bf16


def _loop_calibrated(what, den_step, dev_req, chains, n_steps, cpu_steps, law, kind, guidance):
    """Drive four copies of one request through the same steps: the HIP path; the fp32 oracle chain on the CPU (first `cpu_steps` steps only: ~25-80 s
    each at this size); the SAME oracle code in fp32 on the GPU (stock torch ops; must agree with the CPU chain while both run); and the same graph
    by stock ops in bf16 -- the yardstick.  Asserted at every step: the absolute law, and HIP error <= LOOP_CALIBRATION_RATIO x the stock bf16 error."""
    (c_cpu, m_cpu), (c_g32, m_g32), (c_b16, m_b16) = chains
    a, b = law
    res = str(dev_req.resolution)
    for n in range(1, n_steps + 1):
        den_step({res: [dev_req]})
        chain_ref.denoising_step({res: [c_g32]}, m_g32, kind, guidance)
        chain_ref.denoising_step({res: [c_b16]}, m_b16, kind, guidance)
        assert dev_req.step_index == c_g32.step_index == c_b16.step_index == n
        if n <= cpu_steps:
            chain_ref.denoising_step({res: [c_cpu]}, m_cpu, kind, guidance)
            ol2, omx = _errs(c_g32.latents, c_cpu.latents)
            _log(f"{what}: after step {n}: fp32 oracle chain on the GPU vs on the CPU: rel L2 {ol2:.2e} max {omx:.2e}")
            assert ol2 <= 2e-3 and omx <= 1e-2, f"{what}: the two evaluations of the fp32 oracle chain part after {n} steps ({ol2})"
        l2, mx = _errs(dev_req.latents, c_g32.latents)
        sl2, smx = _errs(c_b16.latents, c_g32.latents)
        _log(f"{what}: after step {n}: HIP rel L2 {l2:.4f} max {mx:.4f} | stock bf16 rel L2 {sl2:.4f} max {smx:.4f} | ratio {l2 / sl2:.2f} (L2) {mx / smx:.2f} (L-inf)"
             f" | law {_law(n, a, b):.4f}")
        assert l2 <= _law(n, a, b) and mx <= 4 * _law(n, a, b), f"{what}: after {n} steps: rel L2 {l2:.4f}, max {mx:.4f} of range"
        assert l2 <= LOOP_CALIBRATION_RATIO * sl2, f"{what}: after {n} steps: HIP rel L2 {l2:.4f} > {LOOP_CALIBRATION_RATIO} x stock bf16 {sl2:.4f}"


def test_sdxl_base_1024_loop_first_steps(full_width_sdxl):
    """The loop AT THE HEADLINE SIZE: SDXL-base widths, ONE 1024 x 1024 request of the 50-step schedule under CFG (UNet batch 2), the first EIGHT steps (round 4:
    four) -- scale, UNet, CFG combine, Euler, latents kept in bf16 between steps on every side -- against the fp32 oracle chain on the original weights, with
    the tolerance CALIBRATED (round 5): the same loop with the oracle's graph evaluated by stock torch ops in bf16 runs beside it, and the HIP loop may be at
    most 1.5 x as far from the fp32 chain as that one is.  The fp32 chain runs on the GPU (the oracle's code, stock ops, seconds per step); its first two
    steps also run on the CPU (~25 s per CFG step on 32 host threads) and the two evaluations must agree."""
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import SDXLDenoiser, synthetic_request
    ocfg, P, _held, net = full_width_sdxl
    den = SDXLDenoiser(net, guidance_scale=5.0)
    r = synthetic_request(0, 1024, 50, UNetConfig.sdxl_base(), den, "cuda:0")
    P32 = {k: v.float() for k, v in P.items()}
    P32g = {k: v.cuda() for k, v in P32.items()}
    P16g = {k: v.to(torch.bfloat16).cuda() for k, v in P.items()}
    m_cpu = lambda x, t, e, te, ti: ref.unet_forward(P32, ocfg, x, t, e, te, ti)
    m_g32 = lambda x, t, e, te, ti: ref.unet_forward(P32g, ocfg, x, t, e, te, ti, device="cuda").cpu()
    m_b16 = lambda x, t, e, te, ti: ref.unet_forward(P16g, ocfg, x, t, e, te, ti, compute_dtype=torch.bfloat16, device="cuda", sdpa=True).float().cpu()
    with torch.inference_mode():
        _loop_calibrated("sdxl-base 1024 px loop", den.denoising_step, r, ((_mirror_sdxl(r), m_cpu), (_mirror_sdxl(r), m_g32), (_mirror_sdxl(r), m_b16)),
                         n_steps=8, cpu_steps=2, law=SDXL_BASE_LAW, kind="sdxl", guidance=5.0)
    del P32g, P16g
    torch.cuda.empty_cache()


# SD3.5-medium at 1024 px: one forward is 1.2 % rel L2 from the oracle (tests/test_headline_shapes_gpu.py); flow matching spends its sigma range uniformly, so the first
# steps of the 28-step schedule move the latents by 1/28 each: the loop error starts far below the forward's and grows slowly (profiles/r04_y_loop_parity_1024_sd3.txt)
SD3_MEDIUM_LAW = (0.01, 0.05)


def test_sd35_medium_1024_loop_first_steps(cuda_device):
    """configs[2] as a LOOP (round 5; a log only in round 4): SD3.5-medium, ONE 1024 x 1024 request of the 28-step flow-match schedule under CFG 7, the first
    two steps against the fp32 oracle chain, calibrated against the stock bf16 loop.  The chain is the oracle's code evaluated in fp32 on the GPU: one CFG step
    of the CPU evaluation costs 130 s at this size, and tests/test_headline_shapes_gpu.py::test_sd35_medium_forward_1024_tolerance_calibrated asserts on this very
    model that the two evaluations agree to 1e-3."""
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    ocfg = sd3_mmdit_ref.MMDiTConfig.sd35_medium()
    P = sd3_mmdit_ref.init_params(ocfg)
    net = MxSD3Transformer(MMDiTConfig.sd35_medium(), P, device="cuda:0")
    den = SD3Denoiser(net, guidance_scale=7.0)
    r = synthetic_sd3_request(0, 1024, 28, MMDiTConfig.sd35_medium(), den, "cuda:0", ctx_len=333)
    P32g = {k: v.float().cuda() for k, v in P.items()}
    P16g = {k: v.to(torch.bfloat16).cuda() for k, v in P.items()}
    m_cpu = lambda x, t, e, p: sd3_mmdit_ref.mmdit_forward(P, ocfg, x, t, e, p)
    m_g32 = lambda x, t, e, p: sd3_mmdit_ref.mmdit_forward(P32g, ocfg, x, t, e, p, device="cuda").cpu()
    m_b16 = lambda x, t, e, p: sd3_mmdit_ref.mmdit_forward(P16g, ocfg, x, t, e, p, compute_dtype=torch.bfloat16, device="cuda", sdpa=True).float().cpu()
    with torch.inference_mode():
        _loop_calibrated("sd3.5-medium 1024 px loop", den.denoising_step, r, ((_mirror_sd3(r), m_cpu), (_mirror_sd3(r), m_g32), (_mirror_sd3(r), m_b16)),
                         n_steps=2, cpu_steps=0, law=SD3_MEDIUM_LAW, kind="sd3", guidance=7.0)
    del net, P32g, P16g
    torch.cuda.empty_cache()
