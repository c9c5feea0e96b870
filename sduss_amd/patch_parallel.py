"""Patch parallelism for ONE request across the GPUs of a node (BASELINE.json configs[3]): the distrifuser baseline the
reference bundles (distrifuser/distrifuser/distrifuser/models/distri_sdxl_unet_pp.py:15-216, utils.py:119-214), synchronous
and stale-asynchronous modes, behind the C ABI (``mx_unet_forward_pp`` / ``mx_unet_forward_pp_stale``, include/mxdenoise.h).

One process per GPU.  Every rank holds the whole UNet (weights are not sharded, as in distrifuser) and the latent ROWS
[rank * H / world, (rank + 1) * H / world).  The step plan itself decides what is exchanged (conv boundary rows, GroupNorm
sums, self-attention K / V^T); this module only supplies the collective: ``torch.distributed.all_gather_into_tensor`` on
views of the plan's workspace -- backend "nccl" is RCCL over xGMI on MI355X -- or, for the CPU-side tests, gloo through
host memory.  The final all-gather of the output rows mirrors distri_sdxl_unet_pp.py:193-195.

Buffer bookkeeping (what ``PatchParallelismCommManager`` does with its flat registered buffer, utils.py:119-214): the plan
allocates every send / receive region from the ONE workspace tensor, so a region is a byte range [offset, offset + n) of that
tensor and the collective needs no registration step; ``CommLog`` records the ranges so the tests can check them.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import lib as _lib
from .unet import MxUNet


def split_rows(latents: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """this rank's rows of [B, C, H, W] (modules/pp/conv2d.py:20-40 slices the same way)"""
    h = latents.shape[2]
    assert h % world == 0, "latent height must divide evenly over the ranks"
    hl = h // world
    return latents[:, :, rank * hl:(rank + 1) * hl].contiguous()


class CommLog:
    """byte ranges of the workspace that went through the collective, per call: (send_off, recv_off, bytes_per_rank)"""

    def __init__(self):
        self.calls: List[Tuple[int, int, int]] = []

    def check(self, ws_bytes: int, world: int) -> None:
        for so, ro, nb in self.calls:
            assert 0 <= so and so + nb <= ws_bytes, "send region outside the workspace"
            assert 0 <= ro and ro + nb * world <= ws_bytes, "receive region outside the workspace"
            assert so + nb <= ro or ro + nb * world <= so, "send and receive regions overlap"
            assert so % 16 == 0 and ro % 16 == 0 and nb % 16 == 0, "regions must be 16-byte aligned"


class PatchParallelUNet:
    """``forward_local(latents_local, ...)`` -> this rank's output rows; ``forward(latents, ...)`` -> the whole output on every
    rank (rows all-gathered).  ``group`` is the torch.distributed group of the ranks sharing the request (distrifuser's
    batch_group, utils.py:93-97)."""

    def __init__(self, unet, group=None, log: Optional[CommLog] = None, mode: str = "sync", warmup_steps: int = 4):
        """mode: "sync" (every step exchanges fresh tensors; distrifuser "full_sync"), "stale_gn" or "corrected_async_gn" (distrifuser's
        default, utils.py:30-32): `warmup_steps` synchronous steps, then stale-asynchronous ones (mx_unet_forward_pp_stale).  Call
        ``reset()`` when a new request starts (distrifuser resets its counters per generation, models/base_model.py)."""
        import torch.distributed as dist
        assert mode in ("sync", "stale_gn", "corrected_async_gn")
        self.mode, self.warmup_steps = mode, warmup_steps
        self.counter = 0
        self._state: Optional[torch.Tensor] = None
        self._pending = []                                   # collectives of the last stale step still in flight
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._cb_async = _lib.ALLGATHER_INPLACE_FN(self._all_gather_async)
        self.unet = unet
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.log = log
        self._ws: Optional[torch.Tensor] = None
        self._cb = _lib.ALLGATHER_FN(self._all_gather)     # keep the callback object alive
        self._err: Optional[BaseException] = None

    # called from inside mx_unet_forward_pp (C -> Python through ctypes); returns 0 on success
    def _all_gather(self, _ctx, _stream, send, recv, nbytes) -> int:
        try:
            ws = self._ws
            so, ro = send - ws.data_ptr(), recv - ws.data_ptr()
            if self.log is not None:
                self.log.calls.append((so, ro, nbytes))
            s = ws[so:so + nbytes]
            r = ws[ro:ro + nbytes * self.world]
            if self.backend == "gloo":               # tests: through host memory (gloo moves CPU tensors)
                host = s.cpu()
                parts = [torch.empty_like(host) for _ in range(self.world)]
                self.dist.all_gather(parts, host, group=self.group)
                r.copy_(torch.cat(parts))
            else:                                    # RCCL: device to device, ordered with the current stream by torch
                self.dist.all_gather_into_tensor(r, s, group=self.group)
            return 0
        except BaseException as e:  # noqa: BLE001  (must not propagate through the C frame)
            self._err = e
            return 1

    # stale steps: in-place all-gather over the slots of one state region; must start after what is queued on the compute stream and
    # finish before the next forward (wait_pending)
    def _all_gather_async(self, _ctx, _stream, region, nbytes) -> int:
        try:
            st = self._state
            off = region - st.data_ptr()
            assert 0 <= off and off + nbytes * self.world <= st.numel() and off % 256 == 0
            if self.log is not None:
                self.log.calls.append((-1, off, nbytes))
            r = st[off:off + nbytes * self.world]
            own = r[self.rank * nbytes:(self.rank + 1) * nbytes]
            if self.backend == "gloo":               # tests: through host memory, completes at once (a legal schedule of the async contract)
                host = own.cpu()
                parts = [torch.empty_like(host) for _ in range(self.world)]
                self.dist.all_gather(parts, host, group=self.group)
                r.copy_(torch.cat(parts))
            else:                                    # RCCL on a side stream: the compute stream runs on while the slots travel
                cur = torch.cuda.current_stream()
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream(device=st.device)
                ev = torch.cuda.Event()
                ev.record(cur)
                self._comm_stream.wait_event(ev)
                with torch.cuda.stream(self._comm_stream):
                    self._pending.append(self.dist.all_gather_into_tensor(r, own, group=self.group, async_op=True))
            return 0
        except BaseException as e:  # noqa: BLE001
            self._err = e
            return 1

    def wait_pending(self) -> None:
        """the compute stream waits for the collectives the last stale step left in flight"""
        for w in self._pending:
            w.wait()
        self._pending = []
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def reset(self) -> None:
        self.wait_pending()
        self.counter = 0

    def _stale_desc(self, sneed: int, shape_key: tuple, dev, corrected: int):
        """The state buffer and the mode of this step.  The state's layout (region offsets, bytes per rank) belongs to ONE problem shape: any
        change of (batch, local rows, width, context length, world) -- even to an equal or smaller footprint -- makes the next step a warm-up
        on a fresh layout.  In-flight collectives of the previous step are waited for BEFORE the old buffer is dropped.
        Warm-up length as distrifuser: exchanges stay synchronous while ``counter <= warmup_steps`` (modules/pp/conv2d.py:97, attn.py:136,
        groupnorm.py:46, distri_sdxl_unet_pp.py:109), i.e. warmup_steps + 1 = 5 synchronous steps at the default of 4."""
        self.wait_pending()
        if self._state is None or self._state.numel() < sneed or getattr(self, "_state_key", None) != shape_key:
            if self._state is None or self._state.numel() < sneed:
                self._state = None
                self._state = torch.empty(sneed, dtype=torch.uint8, device=dev)
            self._state_key = shape_key
            self.counter = 0                         # nothing to be stale about yet
        mode = _lib.PP_WARMUP if self.counter <= self.warmup_steps else _lib.PP_STALE
        self.last_step_mode = mode
        return _lib.PPStale(self._state.data_ptr(), self._state.numel(), mode, corrected, self._cb_async)

    def forward_local(self, latents_local: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                      text_embeds: torch.Tensor, time_ids: torch.Tensor) -> torch.Tensor:
        u = self.unet
        x = latents_local.contiguous()
        b, _c, hl, w = x.shape
        dev = u.device
        ctx_len = encoder_hidden_states.shape[1]
        ts = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        ts = (ts.expand(b) if ts.numel() == 1 else ts).contiguous()
        ehs = encoder_hidden_states.to(device=dev, dtype=torch.bfloat16).contiguous()
        te = text_embeds.to(device=dev, dtype=torch.bfloat16).contiguous()
        ti = time_ids.to(device=dev, dtype=torch.float32).contiguous()
        need = u._lib.mx_unet_workspace_bytes_pp(u._handle, b, hl, w, ctx_len, self.world)
        if need == 0:
            raise _lib.MxError("mx_unet_workspace_bytes_pp: " + u._lib.mx_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((b, u.cfg.out_channels, hl, w), dtype=x.dtype, device=dev)
        comm = _lib.PPComm(self.rank, self.world, self._cb, None)
        self._err = None
        if self.mode == "sync":
            rc = u._lib.mx_unet_forward_pp(u._handle, _lib.current_stream(), x.data_ptr(), _lib.torch_dtype_code(x.dtype), ts.data_ptr(),
                                           ehs.data_ptr(), te.data_ptr(), ti.data_ptr(), out.data_ptr(), b, hl, w, ctx_len, C.byref(comm),
                                           self._ws.data_ptr(), self._ws.numel())
        else:
            sneed = u._lib.mx_unet_pp_state_bytes(u._handle, b, hl, w, ctx_len, self.world)
            if sneed == 0:
                raise _lib.MxError("mx_unet_pp_state_bytes: " + u._lib.mx_last_error().decode())
            stale = self._stale_desc(sneed, ("unet", b, hl, w, ctx_len, self.world), dev, int(self.mode == "corrected_async_gn"))
            rc = u._lib.mx_unet_forward_pp_stale(u._handle, _lib.current_stream(), x.data_ptr(), _lib.torch_dtype_code(x.dtype), ts.data_ptr(),
                                                 ehs.data_ptr(), te.data_ptr(), ti.data_ptr(), out.data_ptr(), b, hl, w, ctx_len,
                                                 C.byref(comm), C.byref(stale), self._ws.data_ptr(), self._ws.numel())
            self.counter += 1
        if self._err is not None:
            raise self._err
        _lib.check(rc, "mx_unet_forward_pp")
        return out

    def forward(self, latents: torch.Tensor, timestep, encoder_hidden_states, text_embeds, time_ids) -> torch.Tensor:
        """whole latent in, whole noise prediction out on every rank (distri_sdxl_unet_pp.py:167-195)."""
        local = self.forward_local(split_rows(latents, self.rank, self.world), timestep, encoder_hidden_states, text_embeds, time_ids)
        if self.backend == "gloo":
            parts = [torch.empty_like(local.cpu()) for _ in range(self.world)]
            self.dist.all_gather(parts, local.cpu(), group=self.group)
            return torch.cat([p.to(local.device) for p in parts], dim=2)
        parts = [torch.empty_like(local) for _ in range(self.world)]
        self.dist.all_gather(parts, local, group=self.group)
        return torch.cat(parts, dim=2)


class PatchParallelSD3(PatchParallelUNet):
    """The same for the SD3 / SD3.5 transformer (``mx_mmdit_forward_pp``; distrifuser models/distri_sd3_transformer_pp.py:87-97,
    modules/pp/attn.py:202-277): rank r owns the image tokens of its latent rows, every rank computes the text stream, each joint block
    all-gathers the image K / V^T.  ``forward(latents, timestep, encoder_hidden_states, pooled)``."""

    def forward_local(self, latents_local: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                      pooled: torch.Tensor) -> torch.Tensor:
        u = self.unet                                        # an MxSD3Transformer
        x = latents_local.contiguous()
        b, _c, hl, w = x.shape
        dev = u.device
        lt = encoder_hidden_states.shape[1]
        ts = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        ts = (ts.expand(b) if ts.numel() == 1 else ts).contiguous()
        ehs = encoder_hidden_states.to(device=dev, dtype=torch.bfloat16).contiguous()
        pp = pooled.to(device=dev, dtype=torch.bfloat16).contiguous()
        need = u._lib.mx_mmdit_workspace_bytes_pp(u._handle, b, hl, w, lt, self.world)
        if need == 0:
            raise _lib.MxError("mx_mmdit_workspace_bytes_pp: " + u._lib.mx_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((b, u.cfg.out_channels, hl, w), dtype=x.dtype, device=dev)
        comm = _lib.PPComm(self.rank, self.world, self._cb, None)
        self._err = None
        stale_ref = None
        if self.mode != "sync":
            sneed = u._lib.mx_mmdit_pp_state_bytes(u._handle, b, hl, w, lt, self.world)
            if sneed == 0:
                raise _lib.MxError("mx_mmdit_pp_state_bytes: " + u._lib.mx_last_error().decode())
            stale = self._stale_desc(sneed, ("mmdit", b, hl, w, lt, self.world), dev, 0)
            stale_ref = C.byref(stale)
            self.counter += 1
        rc = u._lib.mx_mmdit_forward_pp(u._handle, _lib.current_stream(), x.data_ptr(), _lib.torch_dtype_code(x.dtype), ts.data_ptr(),
                                        ehs.data_ptr(), pp.data_ptr(), out.data_ptr(), b, hl, w, lt, C.byref(comm), stale_ref,
                                        self._ws.data_ptr(), self._ws.numel())
        if self._err is not None:
            raise self._err
        _lib.check(rc, "mx_mmdit_forward_pp")
        return out

    def forward(self, latents: torch.Tensor, timestep, encoder_hidden_states, pooled) -> torch.Tensor:
        local = self.forward_local(split_rows(latents, self.rank, self.world), timestep, encoder_hidden_states, pooled)
        host = self.backend == "gloo"
        mine = local.cpu() if host else local
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        return torch.cat([p.to(local.device) for p in parts], dim=2)
