#!/usr/bin/env python3
"""Headline benchmark: SDXL-base 1024x1024, 50-step, CFG, bf16 -- images/s of the denoising hot path, one replica per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one ``denoising_step`` (scale+CFG dup -> UNet -> CFG combine -> Euler) over one batch of B=4 synthetic
1024^2 requests (UNet batch 8), i.e. BASELINE.json configs[1].  images/s = N * B / (50 * step time).  Inputs and
weights are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra legs (outside the timed region):
  roofline     one more step with every GEMM/conv/attention launch bracketed by hipEvents on its stream
               (mx_profile_enable): achieved = algorithmic FLOPs / summed launch time of the dominant kernel
  stream       the BASELINE.md section 4 procedure: fixed-prompt Poisson streams at the reference's offered loads
               (1.0 req/s per GPU x 80 requests, short legs at 0.8 and 1.2), p50 / p90 latency and throughput per load
  mixed_stream BASELINE configs[4] shape: a mixed-resolution Poisson stream (512 / 768 / 1024 px, 30-50 steps), continuous batching with
               the resolutions of a step in ONE launch sequence, the reference's metrics (SLO rate, goodput)
  sd3          BASELINE configs[2]: SD3.5-medium 1024^2 28-step, the same timed-step protocol, with its own roofline
  cpu_baseline the CPU oracle (torch fp32, ORACLE_THREADS = 32 host threads: the count at which it runs fastest on the GPU box) timed on ONE UNet sample-forward at 1024^2 = 1/100 image -- on the inputs and
               weights of one row of the bench's own batch, so the same run also checks that row of the HIP forward (parity_check)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODELS = {   # steps per image, algorithmic FLOP per sample-forward at 1024^2, guidance
    "sdxl": {"steps": 50, "flop": 6.76e12, "name": "SDXL-base-1.0 UNet", "sched": "Euler", "cfg": 5.0, "params": "2.57 B"},
    "sd3": {"steps": 28, "flop": 11.25e12, "name": "SD3.5-medium MMDiT", "sched": "flow-match Euler", "cfg": 7.0, "params": "2.47 B"},
}
STEPS_PER_IMAGE = 50
MFMA_PEAK_BF16 = 2.5e15                    # dense, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12
KIND_NAMES = ["gemm_kernel<128,false> / gemm_small_m_kernel (M <= 16)", "gemm_kernel<64,false> / gemm_small_m_kernel (M <= 16)", "gemm_kernel<128,true> (conv3x3)",
              "gemm_kernel<64,true> / conv3x3_small_n_kernel (conv_in / conv_out)", "attn_fwd_kernel", "groupnorm (3 kernels)", "gemm_v5/v2_kernel<160,false>",
              "gemm_v5/v2_kernel<160,true> (conv3x3)", "gemm_v5/v2_kernel<128,false>", "gemm_v5/v2_kernel<128,true> (conv3x3)",
              "gemm_v4_kernel (256x256 ping-pong)", "attn_cross_kernel (Lk<=96)",
              "attn_tail_kernel (to_out + to_q + cross-attention + to_out, one chained launch)"]


KIND_SYMBOLS = {  # bench kernel label -> regex over the symbols in the rocprofv3 summaries (all instantiations of the kind)
    # (the 160- and 128-feature kinds are served by gemm_v5_kernel for 256-row tiles and gemm_v2_kernel for 128-row tiles: one kind, both symbols)
    "gemm_v5/v2_kernel<160,false>": r"mx::gemm_v[25]_kernel<160, \d+, false", "gemm_v5/v2_kernel<160,true> (conv3x3)": r"mx::gemm_v[25]_kernel<160, \d+, true",
    "gemm_v5/v2_kernel<128,false>": r"mx::gemm_v[25]_kernel<128, \d+, false", "gemm_v5/v2_kernel<128,true> (conv3x3)": r"mx::gemm_v[25]_kernel<128, \d+, true",
    "gemm_v4_kernel (256x256 ping-pong)": r"mx::gemm_v[34]_kernel<", "attn_fwd_kernel": r"mx::attn_fwd", "attn_cross_kernel (Lk<=96)": r"mx::attn_cross_kernel",
}


def pmc_traffic_bytes(kernel_label, model):
    """HBM bytes per launch of a kernel kind from the committed rocprofv3 --pmc passes (profiles/*pmc_traffic_<model>*;
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, tools/pmc_traffic.py), launch-weighted over the kind's
    instantiations.  PMC cannot be collected inside the timed run."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_traffic_{model}*.txt")))
    pat = KIND_SYMBOLS.get(kernel_label)
    if not files or not pat:
        return None, None
    n, mb = 0, 0.0
    for line in open(files[-1]):
        if re.match(pat, line):
            f = line.split()
            n += int(f[-4]); mb += int(f[-4]) * float(f[-1])
    return (mb / n * 1e6, os.path.basename(files[-1])) if n else (None, None)


_T0 = time.perf_counter()


def progress(msg):
    """one line per finished leg on stderr (the JSON line on stdout comes last): long runs stay visibly alive"""
    print(f"[bench {time.perf_counter() - _T0:6.1f} s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="requests per step (UNet batch is 2x under CFG)")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--model", choices=["sdxl", "sd3"], default="sdxl", help="sdxl = BASELINE configs[1] (the headline metric); sd3 = configs[2]")
    ap.add_argument("--sliced", action="store_true", help="is_sliced=True, patch_size=256 (the reference's mixed-policy setting)")
    ap.add_argument("--stream-requests", type=int, default=80, help="requests per GPU of the main Poisson leg (0 = skip the stream legs)")
    ap.add_argument("--stream-rates", type=str, default="1.0,0.8,1.2",
                    help="offered loads in requests/s PER GPU (the reference sweeps {0.8..1.2} x N_gpu req/s, scripts/paper/scalibility.sh:12-13); "
                         "the first is the main leg, the others run stream-requests/5 requests each")
    ap.add_argument("--mix", type=int, default=None, help="(default: 28 on the default SDXL run, else 0) configs[4] leg: this many mixed-resolution requests per GPU (512/768/1024 uniform, steps 30..50 as the "
                                                       "reference traces exp/<model>/qps_*.csv) per offered load of --mix-rates; reports the reference's metrics "
                                                       "(scripts/draw/get_metric.py: SLO rate, average latency, goodput, throughput)")
    ap.add_argument("--mix-rates", type=str, default="1.0,2.0")
    ap.add_argument("--mix-max-batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the informative timing of the stages either side of the loop (text encoders, VAE decode)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sd3", action="store_true", help="skip the SD3.5-medium block (configs[2]) of the default SDXL run")
    ap.add_argument("--no-parity", action="store_true", help="skip the check of one row of the bench batch against the oracle")
    ap.add_argument("--pp", type=int, default=None, help="configs[3] leg: ONE 1024^2 request row-split over this many ranks (distrifuser-style patch parallelism, "
                                                         "stale-asynchronous mode after its warm-up); default = --gpus when --gpus > 1 (0 = skip); must divide --gpus")
    ap.add_argument("--pp-steps", type=int, default=10, help="timed stale steps of the --pp leg (after the 5 synchronous warm-up steps, which are timed too)")
    ap.add_argument("--no-cached-mix", action="store_true", help="skip the mixed-stream leg with the block-skip cache on")
    ap.add_argument("--no-two-model", action="store_true", help="skip the configs[4] leg with SDXL and SD3.5 requests interleaved in one stream")
    a = ap.parse_args()
    if a.mix is None:
        a.mix = 28 if (a.model == "sdxl" and a.res == 1024) else 0
    if a.pp is None:
        a.pp = a.gpus if (a.gpus > 1 and a.model == "sdxl") else 0
    if a.pp and (a.pp < 2 or a.gpus % a.pp):
        ap.error(f"--pp {a.pp} must be >= 2 and divide --gpus {a.gpus}")
    return a


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves -- one process per GPU, the reference's
    worker layout (sduss/executor/mp_executor.py:54-158; distrifuser/scalibility.sh:31 uses torchrun) -- as a CHILD `python -m torch.distributed.run`
    of this process, which has not touched the GPU (no HIP call, no torch.cuda.is_available()), and hand its exit code back.  Never an exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def make_batch(den, cfg, n, res, device, shared, base_id=0):
    if hasattr(cfg, "joint_attention_dim"):
        from sduss_amd.pipeline_sd3 import synthetic_sd3_request
        return [synthetic_sd3_request(base_id + i, res, STEPS_PER_IMAGE, cfg, den, device, shared=shared) for i in range(n)]
    from sduss_amd.pipeline import synthetic_request
    return [synthetic_request(base_id + i, res, STEPS_PER_IMAGE, cfg, den, device, shared=shared) for i in range(n)]


def run_stream(den, cfg, args, device, shared, rate_per_gpu, n_per_gpu, rank, world):
    """Fixed-prompt Poisson stream: exponential inter-arrivals at `rate_per_gpu` x world requests/s (numpy
    seed 10086), 100 % 1024^2, 50 steps, continuous batching (FCFS, at most --batch requests per step) on every replica,
    requests placed by replaying the arrivals through the reference's dispatcher bookkeeping (dp.replay_placement: GreedyPlacer add on arrival /
    finish on completion == greedy.py:16-36 + request_pool.py, with a deterministic service model so that every rank derives the same table
    without a collective; for one resolution and equal loads this is round-robin, what north_star names).
    Returns this rank's per-request latencies (finish - arrival, entrypoints/wrappers.py:31) and its (first arrival, last finish)."""
    n_total = n_per_gpu * world
    rng = np.random.RandomState(10086)                      # reference seed (arg_utils.py:20)
    arrivals = np.cumsum(rng.exponential(1.0 / (rate_per_gpu * world), size=n_total))
    from sduss_amd import dp
    place = dp.replay_placement(arrivals, [args.res] * n_total, [STEPS_PER_IMAGE] * n_total, world, STEP_SECONDS[args.model], args.batch)
    mine = [(i, arrivals[i]) for i in range(n_total) if place[i] == rank]
    pending = make_batch(den, cfg, len(mine), args.res, device, shared, base_id=1000)
    for r, (_i, a) in zip(pending, mine):
        r.arrival = float(a)
    active, done = [], []
    t0 = time.perf_counter()
    last_note = 0.0
    while pending or active:
        now = time.perf_counter() - t0
        if rank == 0 and now - last_note > 60.0:          # a full-protocol leg (500 requests) runs for minutes: stay visibly alive
            last_note = now
            progress(f"stream {rate_per_gpu} req/s: {len(done)} of {len(mine)} requests finished, {len(active)} active")
        while pending and len(active) < args.batch and pending[0].arrival <= now:
            r = pending.pop(0)
            r.start = now
            active.append(r)
        if not active:
            time.sleep(max(0.0, pending[0].arrival - now))
            continue
        den.denoising_step({str(args.res): active}, is_sliced=args.sliced, patch_size=256)
        torch.cuda.synchronize()
        now = time.perf_counter() - t0
        for r in [r for r in active if r.done()]:
            r.finish = now
            done.append(r)
        active = [r for r in active if not r.done()]
    lat = [r.finish - r.arrival for r in done]
    return lat, (min(r.arrival for r in done), max(r.finish for r in done))


REF_DEADLINES_S = {  # SLO = 5 deadlines of the reference's metric script (scripts/draw/get_metric.py:45-57), seconds
    "sdxl": {512: 16.35, 768: 17.5, 1024: 19.31}, "sd3": {512: 11.0, 768: 18.0, 1024: 30.0}}
REF_STEP_MIX = ((30, 0.054), (35, 0.178), (40, 0.460), (45, 0.228), (50, 0.080))   # step histogram of exp/sdxl/qps_1.0.csv
STEP_SECONDS = {  # single-request seconds per step on MI355X (profiles/predictor_{sdxl,sd3}_mi355x.txt): the service model of dp.replay_placement
    "sdxl": {512: 0.0140, 768: 0.0185, 1024: 0.0247}, "sd3": {512: 0.0099, 768: 0.0165, 1024: 0.0260}}


def run_mix(den, cfg, args, device, shared, rate_per_gpu, n_per_gpu, rank, world, model, policy="continuous"):
    """configs[4]: a mixed-resolution Poisson stream with the column shape of the reference traces (arrival ms, resolution, steps;
    tests/server/direct_test.py replays them): resolutions uniform over 512 / 768 / 1024, steps by the traces' histogram, exponential
    inter-arrivals (numpy seed 10086), placed by replaying arrivals and completions through the greedy least-outstanding-pixels dispatcher
    (dp.replay_placement), at most --mix-max-batch requests per step, is_sliced=True / patch 256 as the reference's mixed policies force
    (policy/FCFS_Mixed.py:69-70); the resolutions of a step run in ONE launch sequence (pipeline.py _step_mixed).
    policy "fcfs_mixed": the per-cycle decisions of the reference's worker scheduler under FCFS_Mixed (dp.FcfsMixed, pinned cycle by cycle against
    the reference classes: tests/golden/ref_fcfs_mixed.json) -- the oldest request names the stage, a batch drains before the waiting requests are
    prepared together, a finished request leaves when it is the oldest; PREPARE costs nothing here (fixed prompt: the embeddings are cached,
    SURVEY.md 8d) and POSTPROCESSING (the VAE decode) is outside this bench's loop as everywhere else.
    policy "continuous": a request joins the running batch as soon as a slot is free (FCFS admission, sticky until done) -- what the denoiser
    allows and the reference's ESyMReD policy approaches; reported beside the pinned one.
    Metrics as scripts/draw/get_metric.py computes them (deadlines: its SLO = 5 table, measured on H100 by the reference)."""
    from sduss_amd import dp
    n_total = n_per_gpu * world
    rng = np.random.RandomState(10086)
    arrivals = np.cumsum(rng.exponential(1.0 / (rate_per_gpu * world), size=n_total))
    res_all = rng.choice([512, 768, 1024], size=n_total)
    steps_all = rng.choice([s for s, _ in REF_STEP_MIX], size=n_total, p=[p for _, p in REF_STEP_MIX])
    place = dp.replay_placement(arrivals, [int(r) for r in res_all], [int(v) for v in steps_all], world, STEP_SECONDS[model], args.mix_max_batch)
    mine = [i for i in range(n_total) if place[i] == rank]
    pending = []
    for i in mine:
        if hasattr(cfg, "joint_attention_dim"):
            from sduss_amd.pipeline_sd3 import synthetic_sd3_request
            r = synthetic_sd3_request(5000 + i, int(res_all[i]), int(steps_all[i]), cfg, den, device, shared=shared)
        else:
            from sduss_amd.pipeline import synthetic_request
            r = synthetic_request(5000 + i, int(res_all[i]), int(steps_all[i]), cfg, den, device, shared=shared)
        r.arrival = float(arrivals[i])
        pending.append(r)
    active, done = [], []
    t0 = time.perf_counter()
    if policy == "fcfs_mixed":
        sch, byid = dp.FcfsMixed(args.mix_max_batch), {}
        while pending or sch.has_unfinished():
            now = time.perf_counter() - t0
            while pending and pending[0].arrival <= now:
                r = pending.pop(0)
                byid[r.request_id] = r
                sch.add(r.request_id, r.arrival, r.resolution, r.num_inference_steps)
            if not sch.has_unfinished():
                time.sleep(max(0.0, pending[0].arrival - now))
                continue
            status, chosen, sliced, patch = sch.schedule()
            if status == dp.FcfsMixed.DENOISING:
                den.denoising_step({str(res): [byid[i] for i in ids] for res, ids in sorted(chosen.items())}, is_sliced=sliced, patch_size=patch)
                torch.cuda.synchronize()
            for rid in sch.update((status, chosen)):
                byid[rid].finish = time.perf_counter() - t0
                done.append(byid[rid])
        rows = [(r.resolution, r.finish - r.arrival) for r in done]
        return rows, (min(r.arrival for r in done), max(r.finish for r in done))
    while pending or active:
        now = time.perf_counter() - t0
        while pending and len(active) < args.mix_max_batch and pending[0].arrival <= now:
            active.append(pending.pop(0))
        if not active:
            time.sleep(max(0.0, pending[0].arrival - now))
            continue
        by_res = {}
        for r in active:
            by_res.setdefault(str(r.resolution), []).append(r)
        den.denoising_step(by_res, is_sliced=True, patch_size=256)
        torch.cuda.synchronize()
        now = time.perf_counter() - t0
        for r in [r for r in active if r.done()]:
            r.finish = now
            done.append(r)
        active = [r for r in active if not r.done()]
    rows = [(r.resolution, r.finish - r.arrival) for r in done]
    return rows, (min(r.arrival for r in done), max(r.finish for r in done))


def run_two_model(dens, args, device, rate_per_gpu, n_per_gpu, rank, world):
    """configs[4] as BASELINE.json writes it: ONE arrival-ordered Poisson stream of SDXL *and* SD3.5 requests against replicas that hold BOTH
    denoisers resident (the reference traces exp/sdxl/qps_*.csv and exp/sd3/qps_*.csv share their column shape and step histogram; the model of a
    request is drawn uniformly here).  Placement: arrivals and completions replayed through the greedy least-outstanding-pixels dispatcher
    (dp.replay_placement).  On a replica each model keeps its own FCFS continuous batch (<= --mix-max-batch requests, all resolutions of a step in
    ONE launch sequence, is_sliced=True / patch 256 as policy/FCFS_Mixed.py:69-70 forces); when both models have work their step batches
    ALTERNATE, the model holding the oldest request first.  Metrics per model as scripts/draw/get_metric.py (its SLO = 5 deadlines)."""
    from sduss_amd import dp
    from sduss_amd.pipeline import synthetic_request
    from sduss_amd.pipeline_sd3 import synthetic_sd3_request
    n_total = n_per_gpu * world
    rng = np.random.RandomState(10086)
    arrivals = np.cumsum(rng.exponential(1.0 / (rate_per_gpu * world), size=n_total))
    res_all = rng.choice([512, 768, 1024], size=n_total)
    steps_all = rng.choice([s for s, _ in REF_STEP_MIX], size=n_total, p=[p for _, p in REF_STEP_MIX])
    model_all = rng.choice(["sdxl", "sd3"], size=n_total)
    svc = {r: max(STEP_SECONDS["sdxl"][r], STEP_SECONDS["sd3"][r]) for r in (512, 768, 1024)}
    place = dp.replay_placement(arrivals, [int(r) for r in res_all], [int(v) for v in steps_all], world, svc, args.mix_max_batch)
    pending, shared = [], {"sdxl": {}, "sd3": {}}
    for i in range(n_total):
        if place[i] != rank:
            continue
        m = str(model_all[i])
        den, cfg = dens[m]
        mk = synthetic_request if m == "sdxl" else synthetic_sd3_request
        r = mk(7000 + i, int(res_all[i]), int(steps_all[i]), cfg, den, device, shared=shared[m])
        r.arrival, r.model = float(arrivals[i]), m
        pending.append(r)
    active = {"sdxl": [], "sd3": []}
    done, last = [], None
    t0 = time.perf_counter()
    while pending or active["sdxl"] or active["sd3"]:
        now = time.perf_counter() - t0
        keep = []
        for r in pending:                                   # FCFS admission per model: a full batch of one model does not block the other
            if r.arrival <= now and len(active[r.model]) < args.mix_max_batch:
                active[r.model].append(r)
            else:
                keep.append(r)
        pending = keep
        ready = [m for m in ("sdxl", "sd3") if active[m]]
        if not ready:
            time.sleep(max(0.0, min(r.arrival for r in pending) - now))
            continue
        if len(ready) == 2:                                 # alternate; the model of the oldest request goes first
            m = ("sd3" if last == "sdxl" else "sdxl") if last else min(ready, key=lambda k: min(r.arrival for r in active[k]))
        else:
            m = ready[0]
        last = m
        by_res = {}
        for r in active[m]:
            by_res.setdefault(str(r.resolution), []).append(r)
        dens[m][0].denoising_step(by_res, is_sliced=True, patch_size=256)
        torch.cuda.synchronize()
        now = time.perf_counter() - t0
        for r in [r for r in active[m] if r.done()]:
            r.finish = now
            done.append(r)
        active[m] = [r for r in active[m] if not r.done()]
    rows = [(r.model, r.resolution, r.finish - r.arrival) for r in done]
    return rows, (min(r.arrival for r in done), max(r.finish for r in done))


def two_model_summary(rows, window, rate):
    span = window[1] - window[0]
    out = {"offered_req_per_s_per_gpu": rate, "requests": len(rows), "throughput_req_per_s": len(rows) / span, "per_model": {}}
    ok_all = 0
    for m in ("sdxl", "sd3"):
        mine = [(r, l) for mm, r, l in rows if mm == m]
        if not mine:
            continue
        ok = sum(1 for r, l in mine if l <= REF_DEADLINES_S[m][int(r)])
        ok_all += ok
        lat = [l for _r, l in mine]
        out["per_model"][m] = {"requests": len(mine), "slo_rate": ok / len(mine), "p50_latency_s": float(np.percentile(lat, 50)),
                               "p90_latency_s": float(np.percentile(lat, 90)), "avg_latency_s": float(np.mean(lat)), "goodput_req_per_s": ok / span,
                               "deadlines_s": REF_DEADLINES_S[m]}
    out["slo_rate"], out["goodput_req_per_s"] = ok_all / len(rows), ok_all / span
    out["policy"] = ("both denoisers resident on every replica; one arrival-ordered stream, the model of a request drawn uniformly; per model an FCFS continuous batch "
                     "(all resolutions of a step in ONE launch sequence, is_sliced=True / patch 256); the two models' step batches alternate")
    return out


def run_pp(net, args, dist, device, rank, world):
    """configs[3]: ONE 1024^2 SDXL request (CFG: UNet batch 2) with its latent rows split over --pp ranks, distrifuser's DistriUNetPP
    (distrifuser/distrifuser/distrifuser/models/distri_sdxl_unet_pp.py:15-216; launched there by scalibility.sh:31) in its default mode: 5 synchronous
    warm-up steps, then stale-asynchronous steps ("corrected_async_gn").  Every group of --pp consecutive ranks serves its own request.  Reports the
    per-step time of both modes (max over ranks, fenced), the bytes a rank sends / receives per stale step and the collectives it issues, beside the
    same request on ONE GPU."""
    from sduss_amd import dp
    from sduss_amd.patch_parallel import CommLog, PatchParallelUNet
    pp = args.pp
    group = None
    if pp != world:
        groups = [dist.new_group(list(range(g * pp, (g + 1) * pp))) for g in range(world // pp)]
        group = groups[rank // pp]
    log = CommLog()
    ppnet = PatchParallelUNet(net, group=group, log=log, mode="corrected_async_gn")
    cfg = net.cfg
    g = torch.Generator().manual_seed(10086)                       # the same request on every rank of a group
    h = args.res // 8
    lat = torch.randn(2, cfg.in_channels, h, h, generator=g).to(device=device, dtype=torch.bfloat16)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=g).to(device=device, dtype=torch.bfloat16)
    text = torch.randn(2, cfg.text_embed_dim, generator=g).to(device=device, dtype=torch.bfloat16)
    tids = torch.tensor([[args.res, args.res, 0, 0, args.res, args.res]] * 2, dtype=torch.float32, device=device)
    ts = torch.full((2,), 801.0, device=device)
    use_dev = None if dist.get_backend() == "gloo" else device

    def fence():
        dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        ppnet.wait_pending()
        fence()
        return dp.max_over_ranks((time.perf_counter() - t0) / n, dist, use_dev), out

    one = lambda: net.forward_one(lat, ts, ehs, text, tids)
    one(); one()
    single_s, want = timed(one, max(3, args.pp_steps))
    f = lambda: ppnet.forward(lat, ts, ehs, text, tids)
    sync_s, out_sync = timed(f, ppnet.warmup_steps + 1)            # the warm-up steps ARE the synchronous mode
    sync_calls = len(log.calls) // (ppnet.warmup_steps + 1)
    sync_bytes = sum(nb for _s, _r, nb in log.calls) // (ppnet.warmup_steps + 1)
    log.calls.clear()
    stale_s, out_stale = timed(f, args.pp_steps)
    assert ppnet.last_step_mode == 2, "the timed steps of the --pp leg must be stale-asynchronous ones"
    stale_calls = len(log.calls) // args.pp_steps
    stale_bytes = sum(nb for _s, _r, nb in log.calls) // args.pp_steps
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    return {"ranks_per_request": pp, "request_groups": world // pp, "resolution": args.res, "unet_batch": 2,
            "mode": "distrifuser default: 5 synchronous warm-up steps, then stale-asynchronous steps (corrected_async_gn)",
            "ms_per_step_one_gpu": 1e3 * single_s, "ms_per_step_sync": 1e3 * sync_s, "ms_per_step_stale": 1e3 * stale_s,
            "speedup_stale_vs_one_gpu": single_s / stale_s,
            "collectives_per_step": {"sync": sync_calls, "stale": stale_calls},
            "sent_MB_per_rank_per_step": {"sync": sync_bytes / 1e6, "stale": stale_bytes / 1e6},
            "received_MB_per_rank_per_step": {"sync": sync_bytes * (pp - 1) / 1e6, "stale": stale_bytes * (pp - 1) / 1e6},
            "rel_l2_vs_one_gpu": {"sync": rel(out_sync, want), "stale_on_unchanged_inputs": rel(out_stale, want)},
            "backend": dist.get_backend(),
            "note": "UNet forward incl. the final gather of the output rows; the element-wise scheduler step either side (microseconds) is not in it"}


def probe_diffusers():
    """the preferred CPU baseline is the stock third-party `diffusers` pipeline with the public HF weights.
    Probed at run time, never assumed.  Returns (usable, note)."""
    try:
        import diffusers  # noqa: F401
    except Exception as e:  # noqa: BLE001
        return False, f"diffusers not importable on this box ({type(e).__name__})"
    import glob
    roots = [os.environ.get("HF_HOME"), os.environ.get("HF_HUB_CACHE"), os.path.expanduser("~/.cache/huggingface"), "/workspace/huggingface"]
    for r in roots:
        if r and os.path.isdir(r) and glob.glob(os.path.join(r, "**", "unet", "diffusion_pytorch_model*.safetensors"), recursive=True):
            return True, f"diffusers {diffusers.__version__} and a UNet snapshot under {r}"
    return False, f"diffusers {diffusers.__version__} importable but no model snapshot on disk (no network)"


def time_side_stages(device, batch, step_s):
    """prepare_inference's text encoders and post_inference's VAE decode for one step's worth of requests (random-init weights of the real SDXL
    architectures: CLIP ViT-L, OpenCLIP bigG, the 1024^2 VAE decoder), as a share of a request's 50 steps"""
    from sduss_amd.clip import CLIPTextConfig, MxCLIPTextEncoder, encode_prompt_sdxl
    from sduss_amd.vae import MxVAEDecoder, VAEConfig
    g = torch.Generator().manual_seed(10086)

    def rand_clip(c):
        h, i = c.hidden_size, c.intermediate_size
        P = {"text_model.embeddings.token_embedding.weight": torch.randn(c.vocab_size, h, generator=g) * 0.02,
             "text_model.embeddings.position_embedding.weight": torch.randn(77, h, generator=g) * 0.02,
             "text_model.final_layer_norm.weight": torch.ones(h), "text_model.final_layer_norm.bias": torch.zeros(h)}
        for k in range(c.num_hidden_layers):
            p = f"text_model.encoder.layers.{k}"
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                P[f"{p}.self_attn.{n}.weight"] = torch.randn(h, h, generator=g) * h ** -0.5; P[f"{p}.self_attn.{n}.bias"] = torch.zeros(h)
            for n in ("layer_norm1", "layer_norm2"):
                P[f"{p}.{n}.weight"] = torch.ones(h); P[f"{p}.{n}.bias"] = torch.zeros(h)
            P[f"{p}.mlp.fc1.weight"] = torch.randn(i, h, generator=g) * h ** -0.5; P[f"{p}.mlp.fc1.bias"] = torch.zeros(i)
            P[f"{p}.mlp.fc2.weight"] = torch.randn(h, i, generator=g) * i ** -0.5; P[f"{p}.mlp.fc2.bias"] = torch.zeros(h)
        if c.projection_dim:
            P["text_projection.weight"] = torch.randn(c.projection_dim, h, generator=g) * h ** -0.5
        return P

    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    c1, c2 = CLIPTextConfig.sdxl_text_encoder(), CLIPTextConfig.sdxl_text_encoder_2()
    e1, e2 = MxCLIPTextEncoder(c1, rand_clip(c1), device), MxCLIPTextEncoder(c2, rand_clip(c2), device)
    ids = torch.randint(0, 49000, (2 * batch, 77), generator=g)          # prompt + negative prompt per request
    t_text = timed(lambda: encode_prompt_sdxl(e1, e2, ids, ids))
    del e1, e2
    vcfg = VAEConfig.sdxl()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from vae_bench import shapes as vae_shapes
    VP = {k: (torch.randn(s_, generator=g) * (float(np.prod(s_[1:])) ** -0.5) if len(s_) > 1 else torch.ones(s_) if k.endswith("weight") else torch.zeros(s_))
          for k, s_ in vae_shapes(vcfg).items()}
    vae = MxVAEDecoder(vcfg, VP, device, out_dtype=torch.bfloat16)
    lat = torch.randn(batch, 4, 128, 128, device=device, dtype=torch.bfloat16)
    t_vae = timed(lambda: vae.decode(lat), n=3)
    loop_s = STEPS_PER_IMAGE * step_s                                     # one step batch's requests share their 50 steps
    return {"text_encoders_ms_per_request": 1e3 * t_text / batch, "vae_decode_ms_per_image": 1e3 * t_vae / batch,
            "denoising_loop_ms_per_request": 1e3 * loop_s / batch,
            "share_of_request_time": (t_text + t_vae) / (t_text + t_vae + loop_s),
            "note": "CLIP ViT-L + OpenCLIP bigG on prompt and negative prompt (77 tokens each), SDXL VAE decoder at 1024^2; tokenizers / PIL conversion stay "
                    "on the host; NOT included in `value`"}


ORACLE_THREADS = 32   # host threads of the fp32 torch oracles.  Measured on the GPU box (tools/exp/oracle_threads.py, profiles/r04_oracle_threads.txt: 256 CPUs visible,
                      # torch's default 128 threads): one SD3.5-medium sample-forward 77 s at 128 threads, 55 at 64, 39.6 at 32, 39.4 at 16; SDXL-base 57 / 23 / 11.5 / 12.8 s


def oracle_threads():
    """context: the oracle legs run on ORACLE_THREADS host threads (or fewer where torch's default is lower)"""
    import contextlib

    @contextlib.contextmanager
    def ctx():
        n = torch.get_num_threads()
        torch.set_num_threads(min(n, ORACLE_THREADS))
        try:
            yield torch.get_num_threads()
        finally:
            torch.set_num_threads(n)
    return ctx()


def cpu_baseline(res, model, row=None):
    """The CPU baseline on the host cores: one sample-forward of the denoiser at full width.  Probes for stock diffusers + weights
    first (kind 'diffusers'); on this pool the probe has always come back negative (profiles/r02_probe_env_gpubox.json), so the
    oracle restatement (kind 'port') is timed.  `row` (SDXL): the weights and the inputs of ONE row of the bench's own step batch --
    (params fp32 on the CPU, sample, timestep, ehs, text_embeds, time_ids) -- so that the oracle's answer for the row also checks the
    HIP forward of the headline batch (returned as the second value)."""
    usable, probe_note = probe_diffusers()
    threads = min(torch.get_num_threads(), ORACLE_THREADS)     # (the caller holds oracle_threads())
    if model == "sd3":
        from oracle import sd3_mmdit_ref as mref
        cfg = mref.MMDiTConfig.sd35_medium()
        P = mref.init_params(cfg)
        lat, t, ehs, pooled = mref.make_inputs(cfg, 1, res // 8)
        with torch.inference_mode():
            t0 = time.perf_counter()
            out = mref.mmdit_forward(P, cfg, lat, t, ehs, pooled)
            dt = time.perf_counter() - t0
    else:
        from oracle import sdxl_unet_ref as ref
        cfg = ref.UNetConfig.sdxl_base()
        if row is not None:
            P, sample, t, ehs, text, tids = row
        else:
            P = ref.fast_params(cfg)      # timing-equivalent weights without 2.6e9 RNG draws
            sample, t, ehs, text, tids = ref.make_inputs(cfg, 1, res // 8)
        with torch.inference_mode():
            t0 = time.perf_counter()
            out = ref.unet_forward(P, cfg, sample, t, ehs, text, tids)
            dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return {"value": 1.0 / (dt * 2 * STEPS_PER_IMAGE), "unit": "images/s", "cores": threads, "kind": "port", "diffusers_probe": probe_note,
            "sample": f"1 {MODELS[model]['name']} sample-forward (batch 1, {res}x{res}, fp32 torch oracle) = 1/{2 * STEPS_PER_IMAGE} image, "
                      f"{dt:.1f} s on {threads} threads of {os.cpu_count()} host CPUs; extrapolated x{2 * STEPS_PER_IMAGE}"}, out


def build_model(model, device):
    """(cfg, net, denoiser, params) with random-init weights of the real architecture (no checkpoint on the box)"""
    mdl = MODELS[model]
    if model == "sd3":
        from sduss_amd.config import MMDiTConfig
        from sduss_amd.pipeline_sd3 import SD3Denoiser
        from sduss_amd.transformer_sd3 import MxSD3Transformer
        from sduss_amd.weights import synthetic_mmdit_params
        cfg = MMDiTConfig.sd35_medium()
        P = synthetic_mmdit_params(cfg, device=device)
        net = MxSD3Transformer(cfg, P, device=device)
        return cfg, net, SD3Denoiser(net, guidance_scale=mdl["cfg"]), P
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import SDXLDenoiser
    from sduss_amd.unet import MxUNet
    from sduss_amd.weights import synthetic_params
    cfg = UNetConfig.sdxl_base()
    P = synthetic_params(cfg, device=device)
    net = MxUNet(cfg, P, device=device)
    return cfg, net, SDXLDenoiser(net, guidance_scale=mdl["cfg"]), P


def timed_steps(den, reqs, key, args, dist, device, rehearse):
    """W untimed warm-up steps, then exactly K steps between two fences (barrier + device synchronise), max over ranks: seconds per step"""
    from sduss_amd import dp

    def step():
        den.denoising_step({key: reqs}, is_sliced=args.sliced, patch_size=256)
        for r in reqs:
            if r.done():
                r.step_index = 0        # steady state: keep the batch full

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0, dist, None if rehearse else device)
    return elapsed / args.steps, step


def roofline_leg(step, model):
    """one more step with every GEMM / conv / attention / norm launch bracketed by hipEvents on its stream; returns (kernels, roofline)"""
    from sduss_amd import lib
    l = lib.load()
    l.mx_profile_enable(1)
    step()
    torch.cuda.synchronize()
    buf = (C.c_double * 64)()
    lib.check(l.mx_profile_collect(buf), "mx_profile_collect")
    l.mx_profile_enable(0)
    kinds = []
    for k, name in enumerate(KIND_NAMES):
        n, ms, fl, by = buf[4 * k], buf[4 * k + 1], buf[4 * k + 2], buf[4 * k + 3]
        if n > 0:
            kinds.append({"kernel": name, "launches": int(n), "ms_total": ms, "avg_us": 1e3 * ms / n,
                          "tflops": fl / (ms * 1e-3) / 1e12 if fl else None, "gbps": by / (ms * 1e-3) / 1e9,
                          "frac_of_mfma_peak": fl / (ms * 1e-3) / MFMA_PEAK_BF16 if fl else None})
    roof = None
    mf = [k for k in kinds if k["tflops"]]
    if mf:
        dom = max(mf, key=lambda k: k["ms_total"])
        traffic, src = pmc_traffic_bytes(dom["kernel"], model)
        roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"], "peak": MFMA_PEAK_BF16 / 1e12,
                "unit": "TFLOP/s", "frac": dom["tflops"] / (MFMA_PEAK_BF16 / 1e12), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE; separate rocprofv3 --pmc passes over the same step, committed "
                                "under profiles/: counters cannot be collected inside the timed process)", "traffic_source": src,
                "algorithmic_bytes_per_launch": dom["gbps"] * 1e9 * dom["avg_us"] * 1e-6,
                "launches_per_step": dom["launches"], "avg_launch_us": dom["avg_us"]}
    return kinds, roof


def parity_row_inputs(net, reqs, P, guidance_rows=True):
    """The UNet inputs of the headline batch's first step, assembled exactly as SDXLDenoiser does ([uncond..., cond...]), the HIP forward of
    the WHOLE batch on them, and the last row's inputs for the oracle (the conditional row of the last request)."""
    from sduss_amd import ops
    n = len(reqs)
    lat = torch.cat([r.latents for r in reqs], dim=0)
    sig = torch.tensor([float(r.sigmas[0]) for r in reqs], device=lat.device)
    ts = torch.tensor([float(r.timesteps[0]) for r in reqs], device=lat.device)
    x = ops.euler_scale_input(lat, sig, 2 * n)
    ehs = torch.cat([r.negative_prompt_embeds for r in reqs] + [r.prompt_embeds for r in reqs], dim=0)
    pooled = torch.cat([r.negative_pooled_prompt_embeds for r in reqs] + [r.pooled_prompt_embeds for r in reqs], dim=0)
    tids = torch.cat([t for r in reqs for t in (r.negative_add_time_ids, r.add_time_ids)], dim=0)
    ts2 = torch.cat([ts, ts])
    got = net.forward_one(x, ts2, ehs, pooled, tids)
    k = 2 * n - 1
    f = lambda t: t[k:k + 1].float().cpu()
    P32 = {name: v.float().cpu() for name, v in P.items()}
    return got[k:k + 1].float().cpu(), (P32, f(x), f(ts2).reshape(1), f(ehs), f(pooled), f(tids)), k


def dry_rehearsal(args, rank, world):
    """MX_BENCH_REHEARSE=dry (tests/test_dp_gloo.py, no GPU): the launch / rendezvous / fence / max-over-ranks / single-line control flow of an
    N-rank run with the denoising step replaced by a 1-ms sleep.  Nothing is computed and the line says so; it exists so that the self-spawn path
    and the rank accounting can be exercised where no GPU is present."""
    from sduss_amd import dp
    dist = None
    seen = [(rank, "cpu")]
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, (rank, "cpu"))
    if len({s_[0] for s_ in seen}) != args.gpus:
        raise SystemExit(f"bench.py: {len(seen)} ranks answered, --gpus {args.gpus}")
    fence = (lambda: dist.barrier()) if dist is not None else (lambda: None)
    for _ in range(args.warmup):
        time.sleep(1e-3)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    fence()
    step_s = dp.max_over_ranks(time.perf_counter() - t0, dist, None) / args.steps
    if rank == 0:
        print(json.dumps({"metric": "images/sec (node), SDXL 1024^2 50-step, fixed prompt, CFG", "value": world * args.batch / (MODELS[args.model]["steps"] * step_s),
                          "unit": "images/s", "rehearsal": "MX_BENCH_REHEARSE=dry: NO GPU WORK -- the step is a 1-ms sleep; control-flow check of the N-rank launch only",
                          "n_gpus": world, "ranks_seen": len({s_[0] for s_ in seen}), "rank_devices": [s_[1] for s_ in sorted(seen)],
                          "backend": "gloo" if dist is not None else "none (single process)", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * step_s, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none"}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def sd3_parity_start(net3, reqs3, P3):
    """The SD3.5 block's parity check (configs[2] at the batch the bench times): the HIP forward of the WHOLE batch of 2 x requests on its
    first-step inputs is taken now; the fp32 oracle's answer for its last row (the conditional row of the last request) is computed when the
    returned closure is called -- after every latency-measured leg (round 4 first ran it on a host thread beside the stream legs: 128 oracle
    threads starved the launch path and the stream's p50 went from 2.5 s to 13 s, so it runs alone now, ~75 s of host time).
    Returns (hip row, oracle closure, row index)."""
    from oracle import sd3_mmdit_ref as mref
    n = len(reqs3)
    lat = torch.cat([r.latents for r in reqs3], dim=0)
    x = torch.cat([lat, lat], dim=0)
    ts = torch.tensor([float(r.timesteps[0]) for r in reqs3], device=lat.device)
    ts2 = torch.cat([ts, ts])
    ehs = torch.cat([r.negative_prompt_embeds for r in reqs3] + [r.prompt_embeds for r in reqs3], dim=0)
    pooled = torch.cat([r.negative_pooled_prompt_embeds for r in reqs3] + [r.pooled_prompt_embeds for r in reqs3], dim=0)
    got = net3.forward_one(x, ts2, ehs, pooled)
    k = 2 * n - 1
    f = lambda t: t[k:k + 1].float().cpu()
    row = (f(x), f(ts2).reshape(1), f(ehs), f(pooled))
    P32 = {name: v.float().cpu() for name, v in P3.items()}

    def oracle():
        with torch.inference_mode():
            return mref.mmdit_forward(P32, mref.MMDiTConfig.sd35_medium(), *row)
    return got[k:k + 1].float().cpu(), oracle, k


def config1_first_step(net, cfg, den, device):
    """BASELINE.json configs[0] / BASELINE.md section 3 "always-run plumbing baseline": SDXL, one prompt, 4 denoise steps at 512 x 512.  The HIP
    forward of its first step (UNet batch 2 under CFG) is taken here; cpu_baseline() times the oracle on the same inputs (one of the 4 steps) and
    compares.  Returns (hip noise prediction [2, 4, 64, 64] fp32 on the CPU, oracle inputs)."""
    from sduss_amd import ops
    from sduss_amd.pipeline import synthetic_request
    r = synthetic_request(424242, 512, 4, cfg, den, device, shared={})
    sig = torch.tensor([float(r.sigmas[0])], device=device)
    ts2 = torch.tensor([float(r.timesteps[0])] * 2, device=device)
    x = ops.euler_scale_input(r.latents, sig, 2)
    ehs = torch.cat([r.negative_prompt_embeds, r.prompt_embeds], dim=0)
    pooled = torch.cat([r.negative_pooled_prompt_embeds, r.pooled_prompt_embeds], dim=0)
    tids = torch.cat([r.negative_add_time_ids, r.add_time_ids], dim=0)
    got = net.forward_one(x, ts2, ehs, pooled, tids)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):                                      # the same launch sequence eight times: the UNet share of two 4-step images
        net.forward_one(x, ts2, ehs, pooled, tids)
    e1.record()
    e1.synchronize()
    f = lambda t: t.float().cpu()
    return f(got), (f(x), f(ts2), f(ehs), f(pooled), f(tids)), e0.elapsed_time(e1) / 8e3


def main():
    global STEPS_PER_IMAGE
    args = parse()
    mdl = MODELS[args.model]
    STEPS_PER_IMAGE = mdl["steps"]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                    # before any GPU call in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined (WORLD_SIZE={world}): refusing to report a line for fewer ranks than asked")
    if os.environ.get("MX_BENCH_REHEARSE") == "dry":
        return dry_rehearsal(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # MX_BENCH_REHEARSE=1: all ranks on cuda:0 over gloo -- walks the N > 1 control flow (barriers, max over ranks, gathers, rank-0 line)
    # on a one-GPU box; the numbers of such a run mean nothing and the line says so
    rehearse = os.environ.get("MX_BENCH_REHEARSE") == "1" and world > 1
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # RCCL; used for the barrier + max-over-ranks only: replicas share nothing
    ranks_seen, rank_devices, backend = 1, [dev_index], None
    if dist is not None:
        seen = [None] * world
        dist.all_gather_object(seen, (rank, dev_index, torch.cuda.get_device_properties(dev_index).name))
        ranks_seen, rank_devices, backend = len({s_[0] for s_ in seen}), [s_[1] for s_ in sorted(seen)], dist.get_backend()
        if ranks_seen != args.gpus:
            raise SystemExit(f"bench.py: {ranks_seen} distinct ranks answered, --gpus {args.gpus}")

    from sduss_amd import dp
    cfg, net, den, P = build_model(args.model, device)
    shared = {}
    reqs = make_batch(den, cfg, args.batch, args.res, device, shared)
    key = str(args.res)

    # ---- parity of the bench's OWN batch, before anything is timed: the HIP forward of the whole headline batch on its first-step inputs;
    #      the oracle's answer for one row comes from the cpu_baseline leg below (same run, same weights) ----
    parity_hip = parity_row = None
    want_parity = rank == 0 and world == 1 and args.model == "sdxl" and not args.no_parity and not args.no_cpu_baseline
    config1 = None
    if want_parity:
        parity_hip, parity_row, parity_k = parity_row_inputs(net, reqs, P)
        config1 = config1_first_step(net, cfg, den, device)
    del P

    kv0 = net.context_stats() if hasattr(net, "context_stats") else None
    step_s, step = timed_steps(den, reqs, key, args, dist, device, rehearse)
    kv1 = net.context_stats() if kv0 is not None else None
    images_per_s = world * args.batch / (STEPS_PER_IMAGE * step_s)
    finite = all(torch.isfinite(r.latents.float()).all().item() for r in reqs)

    def headline(model, m, res, step_seconds, ips, batch, fin):
        return {
            "metric": f"images/sec (node), {'SDXL' if model == 'sdxl' else 'SD3.5-medium'} {res}^2 {m['steps']}-step, fixed prompt, CFG",
            "value": ips, "unit": "images/s", "ms_per_step": 1e3 * step_seconds,
            "config": {"workload": f"{m['name']} {res}x{res} {m['steps']}-step {m['sched']}, CFG {m['cfg']}, {batch} requests/step "
                                   f"(denoiser batch {2 * batch}) per GPU, is_sliced={args.sliced}, random-init weights of the "
                                   f"real architecture ({m['params']} params), one data-parallel replica per GPU, no collective",
                       "requests_per_step": batch, "resolution": res, "steps_per_image": m["steps"]},
            "outputs_finite": fin,
            "achieved_tflops_whole_step": 2 * batch * m["flop"] / step_seconds / 1e12 if res == 1024 else None,
            "frac_of_mfma_peak_whole_step": 2 * batch * m["flop"] / step_seconds / MFMA_PEAK_BF16 if res == 1024 else None,
        }
    progress(f"timed region: {1e3 * step_s:.2f} ms/step")
    h = headline(args.model, mdl, args.res, step_s, images_per_s, args.batch, finite)
    kv_note = None
    if kv1 is not None and args.model == "sdxl":
        # Per-composition K / V^T store (mx_unet_set_context_key): the text projection of all 70 cross-attention layers (to_k | to_v on 77 tokens per row: 2 * 77 *
        # 166 400 * 2048 FLOP per UNet row at SDXL-base width) runs once per batch composition instead of once per step -- the closed-loop batch never changes, so
        # the timed steps do not execute it.  The whole-step figure below counts only what ran; `value` is unaffected (it counts images).
        steps_run = args.warmup + args.steps
        hits, misses = kv1[0] - kv0[0], kv1[1] - kv0[1]
        kv_flop_row = 2.0 * 77 * 166400 * 2048
        skipped = kv_flop_row * 2 * args.batch * hits / max(steps_run, 1)
        if args.res == 1024 and hits > 0:
            h["achieved_tflops_whole_step"] = (2 * args.batch * mdl["flop"] - skipped) / step_s / 1e12
            h["frac_of_mfma_peak_whole_step"] = (2 * args.batch * mdl["flop"] - skipped) / step_s / MFMA_PEAK_BF16
        kv_note = {"forwards_served_from_store": hits, "forwards_that_projected": misses, "flop_not_executed_per_step": skipped,
                   "note": "cross-attention K / V^T of encoder_hidden_states kept per batch composition (bit-identical to re-projecting; the reference re-projects the same "
                           "embeddings at every step); achieved_tflops_whole_step excludes the projections that did not run"}
    result = {
        "metric": h["metric"], "value": h["value"], "unit": "images/s",
        **({"rehearsal": "MX_BENCH_REHEARSE=1: every rank on cuda:0 over gloo -- control-flow check only, the numbers mean nothing"} if rehearse else {}),
        "n_gpus": world, "ranks_seen": ranks_seen, "rank_devices": rank_devices, "backend": backend or "none (single process)", "steps": args.steps, "warmup": args.warmup, "ms_per_step": h["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": h["config"], "outputs_finite": finite,
        "achieved_tflops_whole_step": h["achieved_tflops_whole_step"], "frac_of_mfma_peak_whole_step": h["frac_of_mfma_peak_whole_step"],
    }
    if kv_note:
        result["context_kv_store"] = kv_note

    # ---- roofline leg: per-launch hipEvents on the launch stream ----
    if not args.no_roofline and rank == 0:
        result["kernels"], roof = roofline_leg(step, args.model)
        if roof:
            result["roofline"] = roof
    if dist is not None:
        dist.barrier()

    # ---- the second model (configs[2] / configs[4]) is built now: both denoisers stay resident from here (SDXL 5.1 GB + SD3.5 5 GB of weights)
    #      and the HIP side of the SD3.5 parity check is taken on the block's own first-step batch ----
    sd3_built = sd3_parity = None
    want_sd3_block = rank == 0 and world == 1 and args.model == "sdxl" and args.res == 1024 and not args.no_sd3
    want_two_model = args.model == "sdxl" and args.res == 1024 and args.mix > 0 and not args.no_two_model and not args.no_sd3
    if want_sd3_block or want_two_model:
        try:
            cfg3, net3, den3, P3 = build_model("sd3", device)
            sd3_built = (cfg3, net3, den3)
            if want_sd3_block and not args.no_parity and not args.no_cpu_baseline:
                STEPS_PER_IMAGE = MODELS["sd3"]["steps"]
                sd3_parity = sd3_parity_start(net3, make_batch(den3, cfg3, args.batch, 1024, device, {}), P3)
                STEPS_PER_IMAGE = mdl["steps"]
            del P3
        except Exception as e:
            sd3_built = None
            if rank == 0:
                result["sd3"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- stream legs: request latency under Poisson arrivals at the reference's offered loads ----
    if args.stream_requests > 0:
        legs = []
        rates = [float(x) for x in args.stream_rates.split(",") if x]
        for li, rate in enumerate(rates):
            n_req = args.stream_requests if li == 0 else max(8, args.stream_requests // 5)
            lat, window = run_stream(den, cfg, args, device, shared, rate, n_req, rank, world)
            lat, window = dp.gather_stream_stats(lat, window, dist)
            progress(f"stream leg {rate} req/s done")
            if rank == 0:
                legs.append({"offered_req_per_s_per_gpu": rate, "offered_frac_of_closed_loop_capacity": rate * world / images_per_s,
                             "requests": len(lat), "p50_latency_s": float(np.percentile(lat, 50)), "p90_latency_s": float(np.percentile(lat, 90)),
                             "mean_latency_s": float(np.mean(lat)), "throughput_images_per_s": len(lat) / (window[1] - window[0])})
            if dist is not None:
                dist.barrier()
        if rank == 0:
            result["stream"] = {"legs": legs, "arrivals": f"exponential inter-arrival, numpy seed 10086, 100% {args.res}^2 {STEPS_PER_IMAGE}-step, continuous batching "
                                                           f"(<= {args.batch} requests/step), placement: arrivals and completions replayed through the greedy "
                                                           "least-outstanding-pixels dispatcher (round-robin at one resolution)",
                                "latency": "finish - arrival per request; throughput = requests / (last finish - first arrival)"}
            result["p50_latency_s"] = legs[0]["p50_latency_s"]
            result["p90_latency_s"] = legs[0]["p90_latency_s"]
            result["stream_throughput_images_per_s"] = legs[0]["throughput_images_per_s"]

    # ---- configs[4] leg: mixed-resolution stream, the reference's metrics ----
    if args.mix > 0:
        legs = []
        for rate, policy in [(float(x), pol) for x in args.mix_rates.split(",") if x for pol in ("fcfs_mixed", "continuous")]:
            rows, window = run_mix(den, cfg, args, device, shared, rate, args.mix, rank, world, args.model, policy)
            rows_all, window = dp.gather_stream_stats(rows, window, dist)
            progress(f"mixed leg {rate} req/s {policy} done")
            if rank == 0:
                ddl = REF_DEADLINES_S[args.model]
                lat = [l for _r, l in rows_all]
                ok = sum(1 for r, l in rows_all if l <= ddl[int(r)])
                span = window[1] - window[0]
                legs.append({"offered_req_per_s_per_gpu": rate, "policy": policy, "requests": len(rows_all), "slo_rate": ok / len(rows_all), "avg_latency_s": float(np.mean(lat)),
                             "p50_latency_s": float(np.percentile(lat, 50)), "p90_latency_s": float(np.percentile(lat, 90)),
                             "goodput_req_per_s": ok / span, "throughput_req_per_s": len(rows_all) / span,
                             "p50_by_resolution_s": {str(rr): float(np.percentile([l for r, l in rows_all if int(r) == rr], 50)) for rr in (512, 768, 1024)
                                                     if any(int(r) == rr for r, _ in rows_all)}})
            if dist is not None:
                dist.barrier()
        # the same stream with the block-skip cache ON (ESYMRED_USE_CACHE=TRUE at its reference unit, the 256-px patch; one launch sequence per step,
        # one host decision per block).  Random-init weights: a quantile rule asks a fixed half of the patches -- the mechanism's cost, not its quality
        cached_leg = None
        if args.model == "sdxl" and not args.no_cached_mix:
            try:
                from sduss_amd.block_cache import QuantilePredictor
                net.enable_block_cache(QuantilePredictor(0.5))
                rows, window = run_mix(den, cfg, args, device, shared, 1.0, args.mix, rank, world, args.model, "continuous")
                progress("mixed leg with the block cache on done")
                pc = net._patch_cache
                frac = (pc.patches_asked / pc.patches_total) if pc is not None and pc.patches_total else None
                skipped = (sum(bin(h ^ 0x7f).count("1") for h in pc.history) / (7.0 * len(pc.history))) if pc is not None and pc.history else None
                net.disable_block_cache()
                torch.cuda.empty_cache()
                rows_all, window = dp.gather_stream_stats(rows, window, dist)
                if rank == 0:
                    lat = [l for _r, l in rows_all]
                    ok = sum(1 for r, l in rows_all if l <= REF_DEADLINES_S[args.model][int(r)])
                    span = window[1] - window[0]
                    cached_leg = {"offered_req_per_s_per_gpu": 1.0, "policy": "continuous", "requests": len(rows_all), "slo_rate": ok / len(rows_all),
                                  "p50_latency_s": float(np.percentile(lat, 50)), "p90_latency_s": float(np.percentile(lat, 90)),
                                  "goodput_req_per_s": ok / span, "throughput_req_per_s": len(rows_all) / span,
                                  "patch_blocks_asked_frac_rank0": frac, "blocks_skipped_frac_rank0": skipped,
                                  "predictor": "QuantilePredictor(0.5): asks the half of the cached patches whose inputs moved most (random-init weights: timing "
                                               "of the mechanism, approximate outputs by design; the exact path never consults the cache)"}
            except Exception as e:                               # never fatal for the headline line
                net.disable_block_cache()
                if rank == 0:
                    cached_leg = {"error": f"{type(e).__name__}: {e}"}
            if dist is not None:
                dist.barrier()
        if rank == 0:
            result["mixed_stream"] = {"legs": legs, **({"block_cache_on": cached_leg} if cached_leg is not None else {}), "trace": "synthetic, shape of exp/<model>/qps_*.csv: resolutions uniform over 512/768/1024, steps 30-50 by the "
                                                              "traces' histogram, exponential arrivals seed 10086", "deadlines_s": REF_DEADLINES_S[args.model],
                                      "max_batch": args.mix_max_batch, "is_sliced": True, "patch_size": 256,
                                      "policies": {"fcfs_mixed": "the reference worker scheduler's FCFS_Mixed decisions, cycle by cycle (policy/FCFS_Mixed.py:25-76; mirror "
                                                                 "sduss_amd/dp.py FcfsMixed pinned by tests/golden/ref_fcfs_mixed.json): batches drain before waiting requests join",
                                                   "continuous": "a request joins the running batch as soon as a slot is free (FCFS admission)"},
                                      "launches": "the resolutions of a step run in ONE launch sequence (grouped launches)"}

    # ---- configs[3] leg: one request row-split over --pp ranks (patch parallelism), every rank takes part ----
    if args.pp and dist is not None and args.model == "sdxl":
        try:
            pp_res = run_pp(net, args, dist, device, rank, world)
        except Exception as e:                                  # never fatal for the headline line (all ranks fail or pass together: same code path)
            pp_res = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            result["patch_parallel"] = pp_res
        dist.barrier()

    # ---- configs[4] as written: SDXL and SD3.5 requests interleaved in one stream, both denoisers resident (every rank) ----
    if want_two_model and sd3_built is not None:
        try:
            cfg3, net3, den3 = sd3_built
            rows2, win2 = run_two_model({"sdxl": (den, cfg), "sd3": (den3, cfg3)}, args, device, 1.0, args.mix, rank, world)
            rows2, win2 = dp.gather_stream_stats(rows2, win2, dist)
            progress("two-model leg done")
            if rank == 0:
                result.setdefault("mixed_stream", {})["two_model"] = two_model_summary(rows2, win2, 1.0)
        except Exception as e:
            if rank == 0:
                result.setdefault("mixed_stream", {})["two_model"] = {"error": f"{type(e).__name__}: {e}"}
        if dist is not None:
            dist.barrier()

    # ---- informative: the stages either side of the denoising loop (not part of `value`, which is the loop as BASELINE.json defines it) ----
    if rank == 0 and not args.no_stages and args.model == "sdxl" and args.res == 1024:
        try:
            result["stages_either_side"] = time_side_stages(device, args.batch, step_s)
        except Exception as e:                                  # never fatal for the headline line
            result["stages_either_side"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- configs[2]: SD3.5-medium 1024^2 28-step on this GPU, the same timed-step protocol (rank 0 of a one-GPU run only) ----
    if want_sd3_block and sd3_built is not None:
        try:
            del step, den, net, reqs
            torch.cuda.empty_cache()
            m3 = MODELS["sd3"]
            STEPS_PER_IMAGE = m3["steps"]
            cfg3, net3, den3 = sd3_built
            sd3_built = None
            reqs3 = make_batch(den3, cfg3, args.batch, 1024, device, {})
            s3, step3 = timed_steps(den3, reqs3, "1024", args, None, device, False)
            progress(f"sd3 timed region: {1e3 * s3:.2f} ms/step")
            fin3 = all(torch.isfinite(r.latents.float()).all().item() for r in reqs3)
            blk = headline("sd3", m3, 1024, s3, args.batch / (m3["steps"] * s3), args.batch, fin3)
            blk.update({"steps": args.steps, "warmup": args.warmup, "dtype": "bf16", "data": "synthetic", "n_gpus": 1})
            if not args.no_roofline:
                blk["kernels"], roof3 = roofline_leg(step3, "sd3")
                if roof3:
                    blk["roofline"] = roof3
            if args.mix > 0:            # configs[4] for the second model: a short mixed-resolution stream through the MMDiT's single launch sequence
                rows3, win3 = run_mix(den3, cfg3, args, device, {}, 1.0, max(8, args.mix // 2), 0, 1, "sd3")
                lat3 = [l for _r, l in rows3]
                ok3 = sum(1 for r, l in rows3 if l <= REF_DEADLINES_S["sd3"][int(r)])
                blk["mixed_stream"] = {"offered_req_per_s_per_gpu": 1.0, "requests": len(rows3), "slo_rate": ok3 / len(rows3),
                                       "p50_latency_s": float(np.percentile(lat3, 50)), "p90_latency_s": float(np.percentile(lat3, 90)),
                                       "goodput_req_per_s": ok3 / (win3[1] - win3[0]), "throughput_req_per_s": len(rows3) / (win3[1] - win3[0]),
                                       "policy": "FCFS mixed batching, is_sliced=True / patch 256, the resolutions of a step in ONE launch sequence"}
            if sd3_parity is not None:
                hip_row, oracle3, k3 = sd3_parity
                progress("sd3 oracle starts")
                try:
                    with oracle_threads():
                        want3 = oracle3()
                    err = hip_row - want3
                    l2, mx = float(err.norm() / want3.norm()), float(err.abs().max() / want3.abs().max())
                    blk["parity_check"] = {"what": f"row {k3} (conditional row of the last request) of the HIP forward of this block's own batch of {2 * args.batch} at its "
                                                   "first step vs the fp32 oracle on the same weights and inputs",
                                           "rel_l2": l2, "max_err_frac_of_range": mx, "bound_rel_l2": 0.03, "bound_max": 0.05, "ok": bool(l2 <= 0.03 and mx <= 0.05)}
                except Exception as e:                          # noqa: BLE001
                    blk["parity_check"] = {"error": f"{type(e).__name__}: {e}"}
                progress("sd3 oracle done")
            result["sd3"] = blk
            del step3, den3, net3, reqs3
            torch.cuda.empty_cache()
        except Exception as e:                                  # never fatal for the headline line
            result["sd3"] = {"error": f"{type(e).__name__}: {e}"}
        STEPS_PER_IMAGE = mdl["steps"]
        pc3 = result.get("sd3", {}).get("parity_check") if isinstance(result.get("sd3"), dict) else None
        if pc3 and pc3.get("ok") is False:
            raise SystemExit(f"bench.py: parity check of the SD3.5 batch failed: {pc3}")

    # ---- CPU baseline leg (+ the parity check of the headline batch's row) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress("cpu baseline starts")
        _oracle_ctx = oracle_threads()
        _oracle_ctx.__enter__()
        result["cpu_baseline"], want = cpu_baseline(args.res, args.model, parity_row)
        if config1 is not None and parity_row is not None and time.perf_counter() - _T0 > 380.0:
            result["cpu_baseline"]["config1"] = {"skipped": "the run had used its wall-clock allowance (380 s) on a slow host before this leg"}
        elif config1 is not None and parity_row is not None:     # configs[0]: one of its four 512^2 steps on the host cores, same weights
            from oracle import sdxl_unet_ref as oref
            hip1, in1, hip_dt1 = config1
            with torch.inference_mode():
                t0 = time.perf_counter()
                want1 = oref.unet_forward(parity_row[0], oref.UNetConfig.sdxl_base(), *in1)
                dt1 = time.perf_counter() - t0
            e1 = hip1 - want1
            result["cpu_baseline"]["config1"] = {
                "what": "BASELINE configs[0]: SDXL-base, one prompt, 4 denoise steps, 512x512, CFG (UNet batch 2): ONE of the four steps timed on the host cores "
                        "(fp32 torch oracle), x4; the HIP forward of the same step compared with it",
                "cpu_seconds_per_step": dt1, "cpu_seconds_4_steps_extrapolated": 4 * dt1, "cpu_images_per_s": 1.0 / (4 * dt1),
                "hip_seconds_per_step": hip_dt1, "hip_images_per_s_unet_only": 1.0 / (4 * hip_dt1),
                "hip_vs_oracle_rel_l2": float(e1.norm() / want1.norm()), "hip_vs_oracle_max_err_frac_of_range": float(e1.abs().max() / want1.abs().max())}
        _oracle_ctx.__exit__(None, None, None)
        progress("cpu baseline done")
        if parity_row is not None:
            err = (parity_hip - want)
            l2 = float(err.norm() / want.norm())
            mx = float(err.abs().max() / want.abs().max())
            result["parity_check"] = {"what": f"row {parity_k} (conditional row of the last request) of the HIP forward of the bench's own batch of "
                                              f"{2 * args.batch} at its first step vs the fp32 oracle on the same weights and inputs",
                                      "rel_l2": l2, "max_err_frac_of_range": mx, "bound_rel_l2": 0.03, "bound_max": 0.05, "ok": bool(l2 <= 0.03 and mx <= 0.05)}
            if not result["parity_check"]["ok"]:
                raise SystemExit(f"bench.py: parity check of the headline batch failed: {result['parity_check']}")

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
