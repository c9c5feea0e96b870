// The one exchange of the patch-parallel step plans (unet_sdxl.cpp, mmdit_sd3.cpp): an all-gather of `bytes_per_rank` from every rank of the
// request's group, in its synchronous, warm-up and stale-asynchronous forms (include/mxdenoise.h, "Stale-asynchronous steps").
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "../../include/mxdenoise.h"

namespace mx {

struct PPExchange {
  int rank = 0, world = 1;
  mx_allgather_fn ag = nullptr;
  void* ctx = nullptr;
  int mode = MX_PP_SYNC;                 // MX_PP_SYNC / MX_PP_WARMUP / MX_PP_STALE
  bool corrected_gn = false;
  char* state = nullptr;                 // exchange k of a forward owns region k: the exchanges come in a fixed order
  size_t state_bytes = 0, state_top = 0;
  mx_allgather_inplace_fn ag_async = nullptr;

  void set(const mx_pp_comm* c, const mx_pp_stale* s) {
    rank = c->rank; world = c->world; ag = c->all_gather; ctx = c->ctx;
    if (s) { mode = s->mode; corrected_gn = s->corrected_gn != 0; state = (char*)s->state; state_bytes = s->state_bytes; ag_async = s->all_gather_async; }
  }
  // nullptr on success, else what failed.  keep_stale_own (stale steps only): recv keeps the stale copy of this rank's slot too (the
  // corrected GroupNorm needs it).  dry: no launches; with a callback (the comm-plan walk) it sees the arena's placeholder addresses.
  const char* all_gather(hipStream_t stream, bool dry, const void* send, void* recv, size_t bytes_per_rank, bool keep_stale_own = false) {
    char* region = nullptr;
    if (mode != MX_PP_SYNC) {
      region = state + state_top;
      state_top += ((size_t)world * bytes_per_rank + 255) & ~(size_t)255;
      if (!dry && state_top > state_bytes) return "patch-parallel: state buffer too small (mx_*_pp_state_bytes)";
    }
    if (dry && !ag) return nullptr;        // sizing pass
    if (dry || mode != MX_PP_STALE) {
      if (ag(ctx, stream, send, recv, bytes_per_rank)) return "patch-parallel all_gather failed";
      // a warm-up step leaves what it gathered behind for the first stale step (distrifuser: the buffers registered during warm-up)
      if (!dry && mode == MX_PP_WARMUP &&
          hipMemcpyAsync(region, recv, (size_t)world * bytes_per_rank, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return "patch-parallel: state copy failed";
      return nullptr;
    }
    // stale step (utils.py:180-214, modules/pp/*.py `counter > warmup_steps`): the other ranks' slots are what they sent LAST step, this
    // rank's slot is fresh; the fresh slot goes out through the asynchronous collective and is read by the others NEXT step
    const size_t own = (size_t)rank * bytes_per_rank;
    bool e = hipMemcpyAsync(recv, region, (size_t)world * bytes_per_rank, hipMemcpyDeviceToDevice, stream) != hipSuccess;
    if (!keep_stale_own) e |= hipMemcpyAsync((char*)recv + own, send, bytes_per_rank, hipMemcpyDeviceToDevice, stream) != hipSuccess;
    e |= hipMemcpyAsync(region + own, send, bytes_per_rank, hipMemcpyDeviceToDevice, stream) != hipSuccess;
    if (e) return "patch-parallel: stale assembly failed";
    if (ag_async(ctx, stream, region, bytes_per_rank)) return "patch-parallel asynchronous all_gather failed";
    return nullptr;
  }
};

}  // namespace mx
