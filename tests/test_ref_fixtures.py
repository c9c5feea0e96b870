"""Fixtures produced by RUNNING THE REFERENCE (tests/golden/make_ref_fixtures.py, build container only) against the
oracle's restatements and the host-side mirrors.  CPU only; /root/reference is not needed at test time."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_sd3_split_concat_matches_reference(case):
    """oracle.sd3_mmdit_ref.split_sample_sd3 / concat_sample_tokens vs modules/utils.py:86-136 run on the same inputs."""
    from oracle import sd3_mmdit_ref as sd3
    z = np.load(os.path.join(GOLD, "ref_utils_sd3.npz"))
    indices = json.loads(str(z[f"{case}.input_indices"]))
    patch = int(z[f"{case}.patch"][0])
    samples = {res: torch.from_numpy(z[f"{case}.in.{res}"]) for res in indices}
    idx, enc, lat_off, res_off, chunks = sd3.split_sample_sd3(samples, patch, indices)
    assert idx == [str(s) for s in z[f"{case}.indices"]]
    assert enc == [str(s) for s in z[f"{case}.encoder_indices"]]
    assert lat_off == z[f"{case}.latent_offset"].tolist() and res_off == z[f"{case}.resolution_offset"].tolist()
    assert list(chunks.shape) == z[f"{case}.new_sample_shape"].tolist()
    back = sd3.concat_sample_tokens(patch, chunks, lat_off)
    want_keys = sorted(k.split(".")[-1] for k in z.files if k.startswith(f"{case}.concat."))
    assert sorted(back) == want_keys
    for k in want_keys:
        assert torch.equal(back[k], torch.from_numpy(z[f"{case}.concat.{k}"])), f"concat {k}"


def test_sd3_split_concat_key_collision_is_reproduced():
    """the reference names a re-assembled latent by sqrt(chunks) * patch_size; with patch 256 that is the resolution again,
    and with patch 128 on 512 px (case d) it is 4 * 128 = 512 as well -- the fixture holds whatever the reference produced."""
    z = np.load(os.path.join(GOLD, "ref_utils_sd3.npz"))
    for case in "abcd":
        ins = {k.split(".")[-1]: z[k] for k in z.files if k.startswith(f"{case}.in.")}
        outs = {k.split(".")[-1]: z[k] for k in z.files if k.startswith(f"{case}.concat.")}
        assert sorted(ins) == sorted(outs)
        for k in ins:     # split -> concat is the identity on the token axis
            assert np.array_equal(ins[k], outs[k])
    assert str(z["ragged_behaviour"]) != "ok", "the reference cannot stack chunks of unequal length (mixed 512/768 at patch 512)"


@pytest.mark.parametrize("model", ["sdxl", "sd3"])
def test_predictor_features_and_mi355x_refit(model):
    """(a) oracle.predictor_ref reproduces what the reference's Predictor.predict returned, on the reference's own H100
    pickle and on this repo's MI355X re-fit; (b) the re-fit loads, is finite, positive and monotone in the 1024 px count;
    (c) tools/fit_predictor.features is the same feature map."""
    import joblib
    from oracle import predictor_ref
    z = np.load(os.path.join(GOLD, "ref_predictor.npz"))
    rows = z["task_distribute"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mine = joblib.load(os.path.join(ROOT, "profiles", f"schedule_predictor_{model}_mi355x.pkl"))
    got = predictor_ref.predict_step_seconds(mine, rows)
    assert np.allclose(got, z[f"{model}.mi355x.pred"], rtol=1e-9, atol=1e-12)
    assert np.isfinite(got).all() and (got > 0).all()
    only1024 = predictor_ref.predict_step_seconds(mine, [[0, 0, k] for k in range(1, 6)])
    assert (np.diff(only1024) > 0).all(), only1024
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fit_predictor
    assert np.array_equal(fit_predictor.features(rows), predictor_ref.predictor_features(rows))
    # the MI355X table is faster than the H100 one the reference ships, for every composition of the fixture
    assert (z[f"{model}.mi355x.pred"] < z[f"{model}.h100.pred"]).mean() > 0.9


def test_greedy_placer_matches_reference_dispatcher():
    """sduss_amd.dp.GreedyPlacer vs GreedyDispath + RequestPool driven through seeded add / finish scenarios at
    dp_size 1, 2, 4, 8 (mixed resolutions) and 8 (fixed 1024 px: round-robin while loads are equal)."""
    from sduss_amd.dp import GreedyPlacer
    with open(os.path.join(GOLD, "ref_greedy_dispatch.json")) as f:
        scenarios = json.load(f)
    assert [s["dp_size"] for s in scenarios] == [1, 2, 4, 8, 8]
    n_add = 0
    for s in scenarios:
        pl = GreedyPlacer(s["dp_size"])
        for ev in s["events"]:
            if ev["op"] == "add":
                assert pl.add(ev["ids"], ev["resolutions"]) == ev["dp_rank"], ev
                n_add += len(ev["ids"])
            else:
                pl.finish(ev["ids"])
    assert n_add > 200


def test_fcfs_mixed_mirror_reproduces_the_reference_scheduler_cycle_by_cycle():
    """sduss_amd.dp.FcfsMixed vs the reference Scheduler + FCFS_Mixed + WorkerRequestPool (run by make_ref_fixtures.py on a virtual clock):
    the same stage, the same requests grouped by resolution in the same order, every cycle, and the same finish clocks."""
    from sduss_amd.dp import FcfsMixed
    with open(os.path.join(GOLD, "ref_fcfs_mixed.json")) as f:
        scenarios = json.load(f)
    n_cycles = n_mixed = 0
    for s in scenarios:
        sch = FcfsMixed(s["max_num"])
        arr, res, steps, svc = s["arrivals"], s["resolutions"], s["steps"], s["service"]
        clock, nxt, finished = 0.0, 0, {}
        for want in s["cycles"]:
            while True:
                while nxt < len(arr) and arr[nxt] <= clock:
                    sch.add(nxt, arr[nxt], res[nxt], steps[nxt])
                    nxt += 1
                if sch.has_unfinished():
                    break
                clock = arr[nxt]
            status, chosen, sliced, patch = sch.schedule()
            assert [status, {str(k): v for k, v in chosen.items()}, sliced, patch] == want
            assert list(map(str, chosen)) == list(want[1])                      # resolution groups in age order, as the reference's dict
            n_req = sum(len(v) for v in chosen.values())
            clock += max(svc["DENOISING"][str(k)] for k in chosen) if status == "DENOISING" else svc[status] * n_req
            for rid in sch.update((status, chosen)):
                finished[str(rid)] = clock
            n_cycles += 1
            n_mixed += len(chosen) > 1
        assert not sch.has_unfinished() and nxt == len(arr)
        assert finished.keys() == s["finish_clock"].keys()
        assert all(abs(finished[k] - v) < 1e-9 for k, v in s["finish_clock"].items())
    assert n_cycles > 300 and n_mixed > 50


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_split_sample_tables_and_patches_match_reference_bit_for_bit(case):
    """oracle.patch_ref.split_sample vs modules/utils.py:4-84 run on the same latents: padding_idx (top, left, bottom, right), latent_offset,
    resolution_offset, patch_map, the per-patch cache keys "<request id>-<h>-<w>" and the halo'd patches themselves -- integer / copy work,
    bit-exact."""
    from oracle import patch_ref
    z = np.load(os.path.join(GOLD, "ref_split_sample.npz"))
    indices = json.loads(str(z[f"{case}.input_indices"]))
    patch = int(z[f"{case}.patch"][0])
    samples = {res: torch.from_numpy(z[f"{case}.in.{res}"]) for res in indices}
    pad, lat_off, res_off, patches, pmap, keys = patch_ref.split_sample(samples, patch, indices)
    assert np.array_equal(pad.numpy(), z[f"{case}.padding_idx"]) and pad.dtype == torch.int32
    assert lat_off == z[f"{case}.latent_offset"].tolist() and res_off == z[f"{case}.resolution_offset"].tolist()
    assert pmap.tolist() == z[f"{case}.patch_map"].tolist()
    assert keys == [str(k) for k in z[f"{case}.indices"]]
    assert np.array_equal(patches.numpy(), z[f"{case}.new_sample"])
    # and back: concat_sample drops the halos again
    inner = patches[:, :, 1:-1, 1:-1]
    back = patch_ref.concat_sample(patch, inner, lat_off)
    for res, t in samples.items():
        if t.shape[0]:
            assert torch.equal(back[res], t)


def test_split_sample_adjacency_is_symmetric_and_stays_inside_a_latent():
    """The property mx_groupnorm_halo's receiver-driven gather relies on (include/mxdenoise.h): in the tables split_sample produces, b's right
    neighbour has b as its left neighbour (likewise top / bottom) and neighbours always belong to the same latent."""
    z = np.load(os.path.join(GOLD, "ref_split_sample.npz"))
    for case in "abcde":
        pad = z[f"{case}.padding_idx"].reshape(-1, 4)
        pmap = z[f"{case}.patch_map"]
        for b, (top, left, bottom, right) in enumerate(pad):
            for nb, back in ((top, 2), (bottom, 0), (left, 3), (right, 1)):
                if nb >= 0:
                    assert pad[nb][back] == b and pmap[nb] == pmap[b]
