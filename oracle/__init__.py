"""CPU oracle for the sduss/Mixfusion denoiser hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is shipped or measured as the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.

PARITY UNPINNED: the reference holds no golden vectors for this path (SURVEY.md §4,
§8c) and its arithmetic lives in un-vendored ``diffusers==0.32.1`` (conda.yml:50),
which is absent from this image together with the model weights.  The oracle is a
restatement of the published diffusers SDXL ``UNet2DConditionModel`` forward built
from the torch functional primitives that package bottoms out in, plus a literal
restatement of the reference's own patch/halo/scheduler code; it is anchored on the
reference's call sites (file:line cited per function), not on reference outputs.

Pinned by running the reference itself (round 2, tests/golden/make_ref_fixtures.py -> tests/golden/ref_*): the SD3 token
re-chunk (sd3_mmdit_ref.split_sample_sd3 / concat_sample_tokens vs modules/utils.py:86-136) and the latency predictor's
feature map (predictor_ref.py vs policy/ESyMReD.py:46-53).  The UNet / MMDiT arithmetic itself stays unpinned.
"""
