"""Generates fixtures from the REFERENCE ITSELF, in the build container only (it needs /root/reference):

    python tests/golden/make_ref_fixtures.py

The denoiser arithmetic of the reference lives in un-vendored diffusers / xformers and cannot run here (SURVEY.md
section 8c), but three pure-Python pieces on or next to the hot path import cleanly and are run here on seeded inputs:

* ``split_sample_sd3`` / ``concat_sample`` of sduss/model_executor/modules/utils.py:86-136 (loaded by file path: the
  package __init__ pulls diffusers) -- the SD3 token re-chunk either side of PatchSD3Transformer2DModel.forward;
* ``Predictor.predict`` of sduss/worker/scheduler/policy/ESyMReD.py:20-53 -- on the reference's own
  exp/schedule_predictor_{sdxl,sd3}.pkl and on this repo's MI355X re-fit (profiles/schedule_predictor_*_mi355x.pkl);
* ``GreedyDispath.dispatch_requests`` + ``RequestPool`` of sduss/dispatcher/{policy/greedy.py:16-36,request_pool.py} --
  the data-parallel placement that sduss_amd/dp.py mirrors;
* (round 4) ``split_sample`` of sduss/model_executor/modules/utils.py:4-84 -- the index tables of the sliced path (padding_idx, latent_offset,
  resolution_offset, patch_map, the per-patch cache keys) and the halo'd patches; it calls ``.cuda()`` / ``device="cuda"`` on its INTEGER
  tables only, neutralised here for the duration of the call (torch.Tensor.cuda -> identity, torch.tensor(device="cuda") -> CPU);
* (round 4) ``FCFS_Mixed.schedule_requests`` (sduss/worker/scheduler/policy/FCFS_Mixed.py:25-76) inside the reference ``Scheduler``
  (scheduler.py:56-166: schedule / update_reqs_status / process_output) over ``WorkerRequestPool``, driven cycle by cycle through seeded
  arrival traces on a virtual clock -- the per-step batching decisions that bench.py's mixed-stream legs restate (sduss_amd/dp.py FcfsMixed).

Only the resulting DATA travels (tests/golden/ref_*.npz / .json); tests/test_ref_fixtures.py checks the oracle's
restatements and the host mirrors against it.  These fixtures pin the pieces they cover, not the denoiser arithmetic.
"""
import importlib.util
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def load_ref_utils():
    spec = importlib.util.spec_from_file_location("ref_modules_utils", os.path.join(REF, "sduss/model_executor/modules/utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def sd3_split_concat_cases(ru):
    """inputs: per-resolution token tensors [n, L, D] (what pos_embed hands to split_sample_sd3, SD3Transformer.py:82-86)."""
    out = {}
    g = torch.Generator().manual_seed(10086)
    cases = {
        "a": ({"512": 2, "1024": 1}, 256, 2),       # two resolutions, patch 256 -> 4 and 16 chunks per latent
        "b": ({"256": 3}, 256, 4),                  # one patch per latent
        "c": ({"512": 1, "768": 2, "1024": 1}, 256, 2),
        "d": ({"512": 2}, 128, 4),
    }
    for name, (counts, patch, d) in cases.items():
        samples, indices = {}, {}
        rid = 0
        for res, n in counts.items():
            tokens = (int(res) // 16) ** 2          # latent res/8, patch_size 2
            samples[res] = torch.randn(n, tokens, d, generator=g).to(torch.float16).float()
            indices[res] = [f"r{rid + i}" for i in range(n)]
            rid += n
        idx, enc_idx, lat_off, res_off, new_sample = ru.split_sample_sd3(samples, patch, indices)
        back = ru.concat_sample(patch, new_sample, lat_off["cpu"])
        for res, t in samples.items():
            out[f"{name}.in.{res}"] = t.numpy()
        for res, t in back.items():
            out[f"{name}.concat.{res}"] = t.numpy()
        out[f"{name}.new_sample_shape"] = np.array(new_sample.shape) if new_sample.dim() else np.array([0])
        out[f"{name}.latent_offset"] = np.array(lat_off["cpu"], dtype=np.int64)
        out[f"{name}.resolution_offset"] = np.array(res_off["cpu"], dtype=np.int64)
        out[f"{name}.patch"] = np.array([patch])
        out[f"{name}.indices"] = np.array(idx)
        out[f"{name}.encoder_indices"] = np.array(enc_idx)
        out[f"{name}.input_indices"] = np.array(json.dumps(indices))
    return out


def sd3_split_ragged_case(ru):
    """mixed chunk lengths cannot be stacked: the reference raises for mixed resolutions whose per-chunk token counts
    differ (torch.stack of unequal chunks).  Recorded as behaviour, not as a value."""
    samples = {"512": torch.zeros(1, 1024, 4), "768": torch.zeros(1, 2304, 4)}
    try:
        ru.split_sample_sd3(samples, 512, {"512": ["a"], "768": ["b"]})
        return "ok"
    except Exception as e:  # noqa: BLE001
        return type(e).__name__


def predictor_cases():
    os.environ.setdefault("SLO", "5")
    os.environ.setdefault("MODEL", "sdxl")
    sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from sduss.worker.scheduler.policy.ESyMReD import Predictor
        rows = [[a, b, c] for a in range(0, 5) for b in range(0, 4) for c in range(0, 5) if 0 < a + b + c <= 8]
        out = {"task_distribute": np.array(rows, dtype=np.int64)}
        for model in ("sdxl", "sd3"):
            ref_p = Predictor(os.path.join(REF, f"exp/schedule_predictor_{model}.pkl"))
            out[f"{model}.h100.pred"] = np.asarray(ref_p.predict(rows), dtype=np.float64)
            out[f"{model}.latency_table"] = np.array([ref_p.get_latency(r) for r in (512, 768, 1024)])
            mine = os.path.join(ROOT, "profiles", f"schedule_predictor_{model}_mi355x.pkl")
            out[f"{model}.mi355x.pred"] = np.asarray(Predictor(mine).predict(rows), dtype=np.float64)
    return out


def greedy_dispatch_cases():
    sys.path.insert(0, REF)
    from sduss.dispatcher.policy.greedy import GreedyDispath
    from sduss.dispatcher.request_pool import RequestPool
    from sduss.dispatcher.wrappers import ReqStatus, Request

    class _SP:   # the only attribute the dispatcher reads (greedy.py:32, request_pool.py:47)
        def __init__(self, resolution):
            self.resolution = resolution

    scenarios = []
    rng = np.random.RandomState(10086)
    for dp_size, n_events in ((1, 12), (2, 40), (4, 60), (8, 120), (8, 60)):
        pool = RequestPool(dp_size)
        pol = GreedyDispath(request_pool=pool, dp_size=dp_size)
        events, next_id, running = [], 0, []
        fixed = len(scenarios) == 4           # last scenario: fixed 1024 px (north_star's "round-robin")
        for _ in range(n_events):
            kind = rng.rand()
            if kind < 0.6 or not running:
                k = int(rng.randint(1, 4))
                reqs = []
                for _i in range(k):
                    res = 1024 if fixed else int(rng.choice([512, 768, 1024]))
                    r = Request(next_id, _SP(res), arrival_time=0.0)
                    r.status = ReqStatus.WAITING
                    r.dp_rank = None
                    reqs.append(r)
                    next_id += 1
                pool.add_requests(reqs)
                d = pol.dispatch_requests()
                if d:
                    pool.update_requests(sum(d.values(), []))
                assign = {int(r.request_id): int(r.dp_rank) for r in reqs}
                running.extend(r.request_id for r in reqs)
                events.append({"op": "add", "ids": [int(r.request_id) for r in reqs],
                               "resolutions": [int(r.sampling_params.resolution) for r in reqs], "dp_rank": [assign[int(r.request_id)] for r in reqs]})
            else:
                k = int(rng.randint(1, min(3, len(running)) + 1))
                done = [int(x) for x in rng.choice(running, size=k, replace=False)]
                pool.remove_requests(done)
                running = [x for x in running if x not in done]
                events.append({"op": "finish", "ids": done})
        scenarios.append({"dp_size": dp_size, "events": events})
    return scenarios


def split_sample_cases(ru):
    """PatchUNet's split (modules/utils.py:4-84) on small seeded latents: integer tables + the halo'd patches, bit for bit."""
    import contextlib

    @contextlib.contextmanager
    def no_cuda():
        real_cuda, real_tensor = torch.Tensor.cuda, torch.tensor
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: v for kk, v in k.items() if not (kk == "device" and str(v).startswith("cuda"))})
        try:
            yield
        finally:
            torch.Tensor.cuda, torch.tensor = real_cuda, real_tensor

    out = {}
    g = torch.Generator().manual_seed(10086)
    cases = {
        "a": ({"128": 2, "256": 1}, 64, 2),             # scaled-down resolutions: the tables depend on resolution // patch_size only
        "b": ({"256": 3}, 256, 1),                      # one patch per latent: all neighbours -1
        "c": ({"512": 1, "768": 2, "1024": 1}, 256, 1),  # the reference's serving mix at its patch size
        "d": ({"128": 2, "192": 0, "256": 1}, 32, 1),   # an empty resolution in the dict is skipped
        "e": ({"256": 2}, 128, 3),
    }
    for name, (counts, patch, ch) in cases.items():
        samples, indices = {}, {}
        rid = 0
        for res, n in counts.items():
            samples[res] = torch.randn(n, ch, int(res) // 8, int(res) // 8, generator=g)
            indices[res] = [f"req{rid + i}" for i in range(n)]
            rid += n
        with no_cuda():
            idx, pad, lat_off, res_off, new_sample, pmap = ru.split_sample(samples, patch, indices)
        for res, t in samples.items():
            out[f"{name}.in.{res}"] = t.numpy()
        out[f"{name}.patch"] = np.array([patch])
        out[f"{name}.input_indices"] = np.array(json.dumps(indices))
        out[f"{name}.indices"] = np.array(idx)
        out[f"{name}.padding_idx"] = pad["cuda"].numpy().astype(np.int32)
        out[f"{name}.latent_offset"] = np.array(lat_off["cpu"], dtype=np.int64)
        out[f"{name}.resolution_offset"] = np.array(res_off["cpu"], dtype=np.int64)
        out[f"{name}.patch_map"] = np.array(pmap["cpu"], dtype=np.int64)
        out[f"{name}.new_sample"] = new_sample.numpy()
    return out


def fcfs_mixed_cases():
    """The reference Scheduler + FCFS_Mixed policy, cycle by cycle, on a virtual clock.  Per cycle (worker.py:92-140): requests whose arrival time
    has passed are added; schedule(); the virtual clock advances by a service time that depends only on the decision (so the mirror can replay
    it); update_reqs_status(); process_output() of the PREVIOUS cycle is folded into the same cycle (blocking execution: the decisions do not
    depend on the overlap).  Recorded per cycle: [status, {resolution: [request ids]}, is_sliced, patch_size]."""
    os.environ.setdefault("SLO", "5")
    os.environ.setdefault("MODEL", "sdxl")
    sys.path.insert(0, REF)
    from sduss.worker.scheduler.policy.FCFS_Mixed import FCFS_Mixed
    from sduss.worker.scheduler.request_pool import WorkerRequestPool
    from sduss.worker.scheduler.scheduler import Scheduler
    from sduss.worker.wrappers import WorkerReqStatus, WorkerRequest

    class _SP:
        def __init__(self, resolution, steps):
            self.resolution, self.num_inference_steps = resolution, steps

    class _Out:
        def __init__(self, ids):
            self.req_output_dict = {i: None for i in ids}

    scenarios = []
    rng = np.random.RandomState(10086)
    for n_req, rate, max_num in ((30, 1.0, 8), (40, 2.0, 8), (25, 0.5, 4), (30, 4.0, 3)):
        arrivals = np.cumsum(rng.exponential(1.0 / rate, size=n_req))
        res = rng.choice([512, 768, 1024], size=n_req)
        steps = rng.choice([3, 4, 5, 6], size=n_req)            # short loops: the decisions, not the arithmetic
        sched = object.__new__(Scheduler)                        # Scheduler.__init__ only unpacks config objects into these fields
        sched.support_resolutions = [512, 768, 1024]
        sched.max_batchsize = max_num
        sched.request_pool = WorkerRequestPool(sched.support_resolutions)
        sched.policy = FCFS_Mixed(request_pool=sched.request_pool, support_resolutions=sched.support_resolutions)
        sched.cycle_counter = 0
        svc = {"PREPARE": 0.02, "DENOISING": {512: 0.12, 768: 0.2, 1024: 0.35}, "POSTPROCESSING": 0.05}   # slow enough for queues to build
        clock, nxt, cycles, finished = 0.0, 0, [], {}
        while nxt < n_req or sched.has_unfinished_requests():
            while nxt < n_req and arrivals[nxt] <= clock:
                r = WorkerRequest(int(nxt), _SP(int(res[nxt]), int(steps[nxt])))
                r.arrival_time = float(arrivals[nxt])
                sched.add_requests([r])
                nxt += 1
            if not sched.has_unfinished_requests():
                clock = float(arrivals[nxt])
                continue
            out = sched.schedule()
            ids = {int(rr): [int(i) for i in d] for rr, d in out.scheduled_requests.items()}
            name = WorkerReqStatus(out.status).name
            cycles.append([name, ids, out.is_sliced, out.patch_size])
            clock += (max(svc["DENOISING"][rr] for rr in ids) if name == "DENOISING" else svc[name] * sum(len(v) for v in ids.values()))
            sched.update_reqs_status(out)
            done = sched.process_output(out, _Out(out.get_req_ids()))
            for r in done:
                finished[int(r.request_id)] = clock
        scenarios.append({"max_num": max_num, "arrivals": [float(a) for a in arrivals], "resolutions": [int(r) for r in res],
                          "steps": [int(v) for v in steps], "service": {"PREPARE": svc["PREPARE"], "POSTPROCESSING": svc["POSTPROCESSING"],
                                                                       "DENOISING": {str(k): v for k, v in svc["DENOISING"].items()}},
                          "cycles": [[c[0], {str(k): v for k, v in c[1].items()}, c[2], c[3]] for c in cycles],
                          "finish_clock": {str(k): v for k, v in finished.items()}})
    return scenarios


def main():
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    ru = load_ref_utils()
    cases = sd3_split_concat_cases(ru)
    cases["ragged_behaviour"] = np.array(sd3_split_ragged_case(ru))
    np.savez_compressed(os.path.join(HERE, "ref_utils_sd3.npz"), **cases)
    np.savez_compressed(os.path.join(HERE, "ref_predictor.npz"), **predictor_cases())
    with open(os.path.join(HERE, "ref_greedy_dispatch.json"), "w") as f:
        json.dump(greedy_dispatch_cases(), f)
    np.savez_compressed(os.path.join(HERE, "ref_split_sample.npz"), **split_sample_cases(ru))
    with open(os.path.join(HERE, "ref_fcfs_mixed.json"), "w") as f:
        json.dump(fcfs_mixed_cases(), f)
    for n in ("ref_utils_sd3.npz", "ref_predictor.npz", "ref_greedy_dispatch.json", "ref_split_sample.npz", "ref_fcfs_mixed.json"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
