// The one exchange of the patch-parallel step plans (unet_sdxl.cpp, mmdit_sd3.cpp): an all-gather of `bytes_per_rank` from every rank of the
// request's group, in its synchronous, warm-up and stale-asynchronous forms (include/mxdenoise.h, "Stale-asynchronous steps").
//
// Stale steps are COALESCED (round 3): distrifuser enqueues the tensors of up to comm_checkpoint (= 60) consecutive exchanges into one slice of
// its flat buffer and flushes them as ONE asynchronous all_gather (distrifuser/distrifuser/distrifuser/utils.py:184-205).  Here the exchanges of
// a forward are dealt, in order, into at most kMaxChunks chunks; chunk c of the state buffer is laid out [world][L_c] -- rank r's slot holds that
// rank's contributions to the chunk's exchanges back to back -- so that ONE in-place all-gather over the chunk (bytes_per_rank = L_c) moves the
// fresh slots of all its exchanges.  A stale forward therefore issues at most kMaxChunks collectives (plus the caller's gather of the output
// rows) instead of one per exchange; each goes out as soon as the last exchange of its chunk has been produced, so the early chunks travel
// while the rest of the forward computes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include "../../include/mxdenoise.h"

namespace mx {

struct PPExchange {
  static constexpr int kMaxChunks = 8;
  int rank = 0, world = 1;
  mx_allgather_fn ag = nullptr;
  void* ctx = nullptr;
  int mode = MX_PP_SYNC;                 // MX_PP_SYNC / MX_PP_WARMUP / MX_PP_STALE
  bool corrected_gn = false;
  char* state = nullptr;
  size_t state_bytes = 0, state_top = 0; // state_top: bytes the layout needs (mx_*_pp_state_bytes)
  mx_allgather_inplace_fn ag_async = nullptr;
  // recording walk: the byte counts of a forward's exchanges, in order (host only)
  std::vector<size_t>* record = nullptr;
  // layout of the state (built from a recording walk): exchange k -> (chunk, offset inside a rank's slot of the chunk)
  struct Slot { int chunk; size_t off, bytes; bool last; };
  struct Chunk { size_t base, per_rank; };
  std::vector<Slot> slots;
  std::vector<Chunk> chunks;
  size_t next = 0;                       // index of the next exchange of this forward

  void set(const mx_pp_comm* c, const mx_pp_stale* s) {
    rank = c->rank; world = c->world; ag = c->all_gather; ctx = c->ctx;
    if (s) { mode = s->mode; corrected_gn = s->corrected_gn != 0; state = (char*)s->state; state_bytes = s->state_bytes; ag_async = s->all_gather_async; }
  }
  // exchanges -> chunks of (about) equal counts; a rank's slot of a chunk holds its exchanges' bytes back to back (each 256-byte aligned)
  void build_layout(const std::vector<size_t>& sizes) {
    slots.clear(); chunks.clear(); state_top = 0;
    const size_t n = sizes.size();
    if (n == 0) return;
    const size_t per = (n + kMaxChunks - 1) / kMaxChunks;
    for (size_t k0 = 0; k0 < n; k0 += per) {
      Chunk c; c.base = state_top; c.per_rank = 0;
      const size_t k1 = k0 + per < n ? k0 + per : n;
      for (size_t k = k0; k < k1; ++k) {
        slots.push_back({(int)chunks.size(), c.per_rank, sizes[k], k + 1 == k1});
        c.per_rank += (sizes[k] + 255) & ~(size_t)255;
      }
      state_top += (size_t)world * c.per_rank;
      chunks.push_back(c);
    }
  }
  // nullptr on success, else what failed.  keep_stale_own (stale steps only): recv keeps the stale copy of this rank's slot too (the
  // corrected GroupNorm needs it).  dry: no launches; with a callback (the comm-plan walk) it sees the arena's placeholder addresses.
  const char* all_gather(hipStream_t stream, bool dry, const void* send, void* recv, size_t bytes_per_rank, bool keep_stale_own = false) {
    if (record) { record->push_back(bytes_per_rank); return nullptr; }
    if (dry && !ag) return nullptr;        // sizing pass
    if (dry || mode == MX_PP_SYNC) {
      if (ag(ctx, stream, send, recv, bytes_per_rank)) return "patch-parallel all_gather failed";
      return nullptr;
    }
    if (next >= slots.size() || slots[next].bytes != bytes_per_rank) return "patch-parallel: the exchange sequence differs from the state layout";
    const Slot sl = slots[next++];
    const Chunk& ch = chunks[sl.chunk];
    if (ch.base + (size_t)world * ch.per_rank > state_bytes) return "patch-parallel: state buffer too small (mx_*_pp_state_bytes)";
    char* slot0 = state + ch.base + sl.off;                  // rank 0's slot of this exchange; rank r's is per_rank further per rank
    if (mode == MX_PP_WARMUP) {
      if (ag(ctx, stream, send, recv, bytes_per_rank)) return "patch-parallel all_gather failed";
      // a warm-up step leaves what it gathered behind for the first stale step (distrifuser: the buffers registered during warm-up)
      if (hipMemcpy2DAsync(slot0, ch.per_rank, recv, bytes_per_rank, bytes_per_rank, (size_t)world, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return "patch-parallel: state copy failed";
      return nullptr;
    }
    // stale step (utils.py:180-214, modules/pp/*.py `counter > warmup_steps`): the other ranks' slots are what they sent LAST step, this
    // rank's slot is fresh; the fresh slot travels with its chunk's asynchronous collective and is read by the others NEXT step
    const size_t own = (size_t)rank * bytes_per_rank;
    bool e = hipMemcpy2DAsync(recv, bytes_per_rank, slot0, ch.per_rank, bytes_per_rank, (size_t)world, hipMemcpyDeviceToDevice, stream) != hipSuccess;
    if (!keep_stale_own) e |= hipMemcpyAsync((char*)recv + own, send, bytes_per_rank, hipMemcpyDeviceToDevice, stream) != hipSuccess;
    e |= hipMemcpyAsync(slot0 + (size_t)rank * ch.per_rank, send, bytes_per_rank, hipMemcpyDeviceToDevice, stream) != hipSuccess;
    if (e) return "patch-parallel: stale assembly failed";
    if (sl.last && ag_async(ctx, stream, state + ch.base, ch.per_rank)) return "patch-parallel asynchronous all_gather failed";
    return nullptr;
  }
};

}  // namespace mx
