"""Generates fixtures from the REFERENCE ITSELF, in the build container only (it needs /root/reference):

    python tests/golden/make_ref_fixtures.py

The denoiser arithmetic of the reference lives in un-vendored diffusers / xformers and cannot run here (SURVEY.md
section 8c), but three pure-Python pieces on or next to the hot path import cleanly and are run here on seeded inputs:

* ``split_sample_sd3`` / ``concat_sample`` of sduss/model_executor/modules/utils.py:86-136 (loaded by file path: the
  package __init__ pulls diffusers) -- the SD3 token re-chunk either side of PatchSD3Transformer2DModel.forward;
* ``Predictor.predict`` of sduss/worker/scheduler/policy/ESyMReD.py:20-53 -- on the reference's own
  exp/schedule_predictor_{sdxl,sd3}.pkl and on this repo's MI355X re-fit (profiles/schedule_predictor_*_mi355x.pkl);
* ``GreedyDispath.dispatch_requests`` + ``RequestPool`` of sduss/dispatcher/{policy/greedy.py:16-36,request_pool.py} --
  the data-parallel placement that sduss_amd/dp.py mirrors.

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


def main():
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    ru = load_ref_utils()
    cases = sd3_split_concat_cases(ru)
    cases["ragged_behaviour"] = np.array(sd3_split_ragged_case(ru))
    np.savez_compressed(os.path.join(HERE, "ref_utils_sd3.npz"), **cases)
    np.savez_compressed(os.path.join(HERE, "ref_predictor.npz"), **predictor_cases())
    with open(os.path.join(HERE, "ref_greedy_dispatch.json"), "w") as f:
        json.dump(greedy_dispatch_cases(), f)
    for n in ("ref_utils_sd3.npz", "ref_predictor.npz", "ref_greedy_dispatch.json"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
