// Tables and launchers of the patch-unit block cache (patch_cache.hip; used by unet_sdxl.cpp).
#pragma once
#include <hip/hip_runtime.h>

namespace mx {
// level-0 description of a sample of the batch: first row of its image in the concatenated level-0 activations, latent size, its row of the state
// tensors (one per request), patches per image row.  At level l: image (h >> l) x (w >> l), first row row0 >> 2l, patch edge p0 >> l.
struct PcSample { long long row0; int h, w, slot, npx; };
struct PcPatch { int b, py, px, pad; };
struct PcRange { long long row0; int rows, slot, srow0; };     // rows [row0, row0 + rows) of a batch tensor <-> rows [srow0, ...) of state row `slot`

int launch_pc_image_copy(hipStream_t st, void* batch, void* state, const void* samp, int B, int level, int C, long state_row_elems, int to_batch,
                         const float* vec, int ldvec, const void* residual, long max_image_elems, int gate = 0);
int launch_pc_rows_load_stats(hipStream_t st, void* batch, const void* state, const void* samp, int B, int level, int C, long state_row_elems, const void* residual,
                              float* stats1, float* fin, float eps, long max_image_rows);
int launch_pc_range_copy(hipStream_t st, void* batch, void* state, long state_row_elems, int C, const void* ranges, int n, int to_batch, long max_range_elems);
int launch_pc_range_sq_diff(hipStream_t st, const void* x, const void* state, long state_row_elems, int C, const void* ranges, int n, double* partial);
int launch_pc_gather(hipStream_t st, const void* src, int ld_src, int C, void* dst, const void* list, int n, const void* samp, int level, int p, int halo_lo,
                     int halo_hi, int up);
int launch_pc_scatter(hipStream_t st, const void* src, int Ps, int o0, int C, void* state, long state_row_elems, const void* list, int n, const void* samp,
                      int level, int p);
int launch_pc_patch_store(hipStream_t st, const void* batch, void* state, long state_row_elems, int C, const void* list, int n, const void* samp, int level, int p);
int launch_pc_patch_sq_diff(hipStream_t st, const void* x, const void* state, long state_row_elems, int C, const void* list, int n, const void* samp, int level,
                            int p, double* partial);
}  // namespace mx
