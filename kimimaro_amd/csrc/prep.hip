// prep.hip -- whole-volume streaming kernels around the per-label searches (gfx950).
//
//   kh_label_stats    fastremap.unique counts (kimimaro/intake.py:198), np.max(DBF) per label
//                     (kimimaro/trace.py:100), first_label (skeletontricks.pyx:307-326) and the
//                     x extent of find_objects (kimimaro/utility.py:85-102), in ONE pass.
//   kh_scatter_lists  per-label voxel index lists (the flatnonzero of skeletontricks.pyx:1001).
//   kh_neighbor_mask  26-bit same-label connectivity (replaces the per-crop bounds + mask tests
//                     of dijkstra_invalidation.hpp:60-124 / dijkstra3d's neighbourhood code).
//   kh_pdrf           zero2inf + inf2zero + compute_pdrf fused (kimimaro/trace.py:138,146,315-356).
//
// All are HBM-bound sweeps: lanes along x, wave-aggregated atomics (one atomic per run of equal
// labels inside a wave instead of one per voxel).
#include "common.h"

namespace kh {

template <typename LT>
__device__ __forceinline__ uint32_t ld_label(const LT* p, int64_t i) { return (uint32_t)p[i]; }

// run structure of a wave: lanes hold consecutive voxels; a run = maximal lane interval with the
// same label that does not cross an x == 0 row start.
struct WaveRun {
  int start;   // first lane of my run
  int end;     // last lane of my run
};

__device__ __forceinline__ WaveRun wave_runs(uint32_t label, bool valid, bool row_start) {
  const int lane = threadIdx.x & 63;
  const uint32_t prev = __shfl_up(label, 1);
  const bool pvalid = __shfl_up((int)valid, 1) != 0;
  const bool leader = (lane == 0) || (label != prev) || row_start || (valid != pvalid);
  const unsigned long long lead = __ballot(leader);
  // start = highest leader bit <= lane ; end = (next leader bit > lane) - 1
  const unsigned long long below = lead & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  WaveRun r;
  r.start = 63 - __clzll((long long)below);
  const unsigned long long above = (lane == 63) ? 0ull : (lead & ~((2ull << lane) - 1ull));
  r.end = above ? (__ffsll((long long)above) - 2) : 63;
  return r;
}

template <typename LT>
__global__ __launch_bounds__(256) void label_stats_kernel(const LT* __restrict__ lab, const float* __restrict__ dbf,
                                                          int64_t nvox, int sx, int sy, uint32_t* counts, uint32_t* dbf_max_bits,
                                                          uint32_t* first_index, uint32_t* xmin, uint32_t* xmax,
                                                          uint32_t* yzext) {
  const int lane = threadIdx.x & 63;
  const int64_t nchunks = (nvox + 255) / 256;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t i = c * 256 + threadIdx.x;
    const bool valid = i < nvox;
    const uint32_t L = valid ? ld_label(lab, i) : 0u;
    const int x = valid ? (int)(i % sx) : 0;
    float v = valid ? dbf[i] : 0.0f;
    const WaveRun r = wave_runs(L, valid, x == 0);
    // segmented max scan of v over the run
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(v, d);
      if (lane - d >= r.start) v = fmaxf(v, o);
    }
    if (valid && L != 0 && lane == r.end) {
      const int len = r.end - r.start + 1;
      atomicAdd(&counts[L], (uint32_t)len);
      atomicMax(&dbf_max_bits[L], __float_as_uint(v));  // v >= 0
      atomicMin(&first_index[L], (uint32_t)(i - (len - 1)));
      atomicMin(&xmin[L], (uint32_t)(x - (len - 1)));
      atomicMax(&xmax[L], (uint32_t)x);
      if (yzext) {  // bounding box in y and z: [ymin, ymax, zmin, zmax] per label
        const int64_t row = i / sx;
        const uint32_t y = (uint32_t)(row % sy), z = (uint32_t)(row / sy);
        atomicMin(&yzext[4 * (size_t)L + 0], y);
        atomicMax(&yzext[4 * (size_t)L + 1], y);
        atomicMin(&yzext[4 * (size_t)L + 2], z);
        atomicMax(&yzext[4 * (size_t)L + 3], z);
      }
    }
  }
}

template <typename LT>
__global__ __launch_bounds__(256) void scatter_lists_kernel(const LT* __restrict__ lab, int64_t nvox,
                                                            const int32_t* __restrict__ slot_of_label,
                                                            const uint32_t* __restrict__ offsets, uint32_t* cursors,
                                                            uint32_t* __restrict__ lists) {
  const int lane = threadIdx.x & 63;
  const int64_t nchunks = (nvox + 255) / 256;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t i = c * 256 + threadIdx.x;
    const bool valid = i < nvox;
    const uint32_t L = valid ? ld_label(lab, i) : 0u;
    const int slot = (valid && L != 0) ? slot_of_label[L] : -1;
    const WaveRun r = wave_runs(L, valid, false);
    uint32_t base = 0;
    if (slot >= 0 && lane == r.end) base = offsets[slot] + atomicAdd(&cursors[slot], (uint32_t)(r.end - r.start + 1));
    base = __shfl(base, r.end);
    if (slot >= 0) lists[base + (uint32_t)(lane - r.start)] = (uint32_t)i;
  }
}

template <typename LT>
__global__ __launch_bounds__(256) void neighbor_mask_kernel(const LT* __restrict__ lab, int sx, int sy, int sz,
                                                            uint32_t* __restrict__ out) {
  // grid: x tiles of 256, y, z folded into blockIdx.x (1-D grid, row major)
  const int xt = (sx + 255) >> 8;
  const int64_t ntiles = (int64_t)xt * sy * sz;
  const int64_t sxy = (int64_t)sx * sy;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int x = (int)(t % xt) * 256 + threadIdx.x;
    const int64_t r = t / xt;
    const int y = (int)(r % sy), z = (int)(r / sy);
    if (x >= sx) continue;
    const int64_t i = x + (int64_t)sx * y + sxy * z;
    const uint32_t L = ld_label(lab, i);
    uint32_t m = 0;
    if (L != 0) {
#pragma unroll
      for (int k = 0; k < 26; k++) {
        int dx, dy, dz;
        dir_delta(k, dx, dy, dz);
        const int nx = x + dx, ny = y + dy, nz = z + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
        if (ld_label(lab, i + dx + (int64_t)sx * dy + sxy * dz) == L) m |= (1u << k);
      }
    }
    out[i] = m;
  }
}

template <typename LT>
__global__ __launch_bounds__(256) void pdrf_kernel(const LT* __restrict__ lab, int64_t nvox,
                                                   const int32_t* __restrict__ slot_of_label,
                                                   const kh_label_t* __restrict__ tasks, const float* __restrict__ dbf,
                                                   float* __restrict__ daf, int nsq, float scale, float* __restrict__ pdrf, int keep) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * 256) {
    const uint32_t L = ld_label(lab, i);
    const int slot = L ? slot_of_label[L] : -1;
    float p = KH_INF;
    if (slot < 0 && keep) continue;        // KH_PDRF_KEEP_OTHERS: not this call's voxel
    if (slot >= 0) {
      const float M = tasks[slot].M;
      const float max_daf = tasks[slot].max_val;
      if (nsq != KH_PDRF_FINISH) {
        p = dbf[i] * M;        // np.multiply(DBF, M)            trace.py:341
        p = 1.0f - p;          // np.subtract(f(1), PDRF)        trace.py:342
      } else {
        p = pdrf[i];           // the base raised to the exponent by np.power on the host  trace.py:346-347
      }
      if (nsq == KH_PDRF_BASE) { pdrf[i] = p; continue; }
      for (int s = 0; s < nsq; s++) p = p * p;  //             trace.py:343-345
      p = p * scale;         // PDRF *= f(pdrf_scale)          trace.py:349
      float d = daf[i];
      if (d == KH_INF) d = 0.0f;  // inf2zero                  trace.py:146
      if (max_daf != 0.0f) {
        const float inv = 1.0f / max_daf;  // (1 / max_daf) in float32 (numpy 2 scalar)  trace.py:353
        d = d * inv;
        p = p + d;           //                                trace.py:354
      }
      daf[i] = d;
    }
    pdrf[i] = p;
  }
}

template <typename LT>
__global__ __launch_bounds__(256) void init_alive_kernel(const LT* __restrict__ lab, int64_t nvox,
                                                         const int32_t* __restrict__ slot_of_label, uint8_t* __restrict__ alive) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * 256) {
    const uint32_t L = ld_label(lab, i);
    alive[i] = (L != 0 && slot_of_label[L] >= 0) ? 1 : 0;
  }
}

// a3 / a5 as plain element-wise kernels (the function-level mirrors; the path loop uses the fused pdrf_kernel)
__global__ void zero2inf_kernel(float* f, int64_t n) {   // skeletontricks.pyx:203-224
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (f[i] == 0.0f) f[i] = KH_INF;
}
__global__ void inf2zero_kernel(float* f, int64_t n) {   // skeletontricks.pyx:177-198
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (f[i] == KH_INF) f[i] = 0.0f;
}
// compute_pdrf, kimimaro/trace.py:315-356, on every element (DBF already through zero2inf: background -> +-inf)
__global__ void pdrf_field_kernel(const float* __restrict__ dbf, float* __restrict__ daf, int64_t n, float M, int nsq, float scale,
                                  float max_daf, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float p;
    if (nsq != KH_PDRF_FINISH) {
      p = dbf[i] * M;
      p = 1.0f - p;
    } else {
      p = out[i];
    }
    if (nsq == KH_PDRF_BASE) { out[i] = p; continue; }
    for (int s = 0; s < nsq; s++) p = p * p;
    p = p * scale;
    if (max_daf != 0.0f) {
      const float inv = 1.0f / max_daf;
      const float d = daf[i] * inv;
      daf[i] = d;
      p = p + d;
    }
    out[i] = p;
  }
}
// CachedTargetFinder.find_target (skeletontricks.pyx:1008-1045): the valid voxel with the largest DAF, ties -> largest index
__global__ __launch_bounds__(1024) void target_max_kernel(const uint32_t* __restrict__ list, const float* __restrict__ list_daf,
                                                          const uint8_t* __restrict__ alive, uint32_t n, unsigned long long* out) {
  __shared__ unsigned long long red[16];
  unsigned long long best = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t v = list[i];
    if (!alive[v]) continue;
    // bit 63 marks "found" (DAF >= 0: the sign bit is free).  It is part of the compared key: `best` carries it from the
    // first valid voxel on, so a bare key would never beat it again (lists longer than the block, round-2 advisor finding).
    const unsigned long long key = ((unsigned long long)__float_as_uint(list_daf[i]) << 32) | v | (1ull << 63);
    if (key > best) best = key;                    // equal keys cannot occur: the voxel index is in the key
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long ob = __shfl_xor(best, o); if (ob > best) best = ob; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned w = 1; w < (blockDim.x >> 6); w++) if (red[w] > best) best = red[w];
    *out = best;
  }
}

// legacy skeletontricks.find_target (skeletontricks.pyx:331-367): the first voxel, scanning x outermost / z innermost,
// whose value is the maximum over the mask (strict >, starting from -inf: a -inf or NaN voxel is never chosen).
// key = orderable float bits << 32 | (~scan position): one atomicMax per wave.
__global__ __launch_bounds__(256) void find_target_kernel(const uint8_t* __restrict__ labels, const float* __restrict__ field,
                                                          int64_t sx, int64_t sy, int64_t sz, unsigned long long* out) {
  const int64_t n = sx * sy * sz;
  unsigned long long best = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (!labels[i]) continue;
    const float f = field[i];
    if (!(f > -KH_INF)) continue;                      // PDRF[x,y,z] > maxpdrf fails for -inf and for NaN
    const uint32_t b = __float_as_uint(f);
    const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // monotone map of the floats onto unsigned
    const int64_t z = i / (sx * sy), r = i - z * sx * sy, y = r / sx, x = r - y * sx;
    const uint32_t scan = (uint32_t)((x * sy + y) * sz + z);
    const unsigned long long key = ((unsigned long long)ord << 32) | (0xFFFFFFFFu - scan);
    if (key > best) best = key;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long ob = __shfl_xor(best, o); if (ob > best) best = ob; }
  if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}
// skeletontricks.first_label (skeletontricks.pyx:307-326): the first non-zero voxel of the z / y / x raster = the smallest
// Fortran linear index
__global__ __launch_bounds__(256) void first_label_kernel(const uint8_t* __restrict__ labels, int64_t n, unsigned long long* out) {
  unsigned long long best = ~0ull;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if (labels[i] && (unsigned long long)i < best) best = (unsigned long long)i;
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long ob = __shfl_xor(best, o); if (ob < best) best = ob; }
  if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(out, best);
}

__global__ void fill_f32_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_u32_kernel(uint32_t* p, int64_t n, uint32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_u8_kernel(uint8_t* p, int64_t n, uint8_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void gather_f32_kernel(const float* __restrict__ src, const uint32_t* __restrict__ idx, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = src[idx[i]];
}

static inline unsigned grid_for(int64_t n, int per_block, int64_t cap = 8192) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace kh

using namespace kh;

#define KH_DISPATCH_LT(bytes, CALL)                          \
  switch (bytes) {                                           \
    case 1: { typedef uint8_t LT; CALL; } break;             \
    case 2: { typedef uint16_t LT; CALL; } break;            \
    case 4: { typedef uint32_t LT; CALL; } break;            \
    default: set_error("label_bytes must be 1, 2 or 4"); return KH_EINVAL; \
  }

__global__ void init_yzext_kernel(uint32_t* p, int64_t n1) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (int64_t)gridDim.x * blockDim.x) {
    p[4 * i + 0] = 0xFFFFFFFFu; p[4 * i + 1] = 0u; p[4 * i + 2] = 0xFFFFFFFFu; p[4 * i + 3] = 0u;
  }
}

extern "C" int kh_label_stats(const void* labels, int label_bytes, const float* dbf, int64_t nvox, int64_t sx, int64_t sy,
                              int64_t nlabels, uint32_t* counts, float* dbf_max, uint32_t* first_index, uint32_t* xmin,
                              uint32_t* xmax, uint32_t* yz_extent, void* stream) {
  if (int rc = require_device()) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n1 = nlabels + 1;
  hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, st, counts, n1, 0u);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, st, (uint32_t*)dbf_max, n1, 0u);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, st, first_index, n1, 0xFFFFFFFFu);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, st, xmin, n1, 0xFFFFFFFFu);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, st, xmax, n1, 0u);
  if (yz_extent) hipLaunchKernelGGL(init_yzext_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, st, yz_extent, n1);
  KH_DISPATCH_LT(label_bytes, hipLaunchKernelGGL((label_stats_kernel<LT>), dim3(grid_for(nvox, 256)), dim3(256), 0, st,
                                                 (const LT*)labels, dbf, nvox, (int)sx, (int)sy, counts, (uint32_t*)dbf_max,
                                                 first_index, xmin, xmax, yz_extent));
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_scatter_lists(const void* labels, int label_bytes, int64_t nvox, const int32_t* slot_of_label,
                                int64_t nslots, const uint32_t* offsets, uint32_t* cursors, uint32_t* lists, void* stream) {
  if (int rc = require_device()) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, st, cursors, nslots, 0u);
  KH_DISPATCH_LT(label_bytes, hipLaunchKernelGGL((scatter_lists_kernel<LT>), dim3(grid_for(nvox, 256)), dim3(256), 0, st,
                                                 (const LT*)labels, nvox, slot_of_label, offsets, cursors, lists));
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_neighbor_mask(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz, uint32_t* nbrmask,
                                void* stream) {
  if (int rc = require_device()) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = ((sx + 255) / 256) * sy * sz;
  KH_DISPATCH_LT(label_bytes, hipLaunchKernelGGL((neighbor_mask_kernel<LT>), dim3(grid_for(ntiles, 1, 1 << 20)), dim3(256), 0,
                                                 st, (const LT*)labels, (int)sx, (int)sy, (int)sz, nbrmask));
  KH_LAUNCH_CHECK();
  return KH_OK;
}

namespace kh {
__global__ __launch_bounds__(256) void apply_voxel_graph_kernel(uint32_t* nbrmask, const uint32_t* __restrict__ graph, int64_t nvox,
                                                                uint8_t* corner_gate) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t geo = nbrmask[v];
  const uint32_t allowed = graph_to_directions(graph[v]);
  nbrmask[v] = geo & allowed;
  if (corner_gate) {
    // a corner entry (18..25) of a voxel on an x face of its array degenerates into the yz diagonal with the corner's y / z steps
    // (dijkstra_invalidation.hpp:116-123), and the graph gates it by the CORNER's bit (:182-190): bit j = that diagonal exists
    // (same label, in bounds) and corner 18 + j is allowed.  Both the heap emulation and the sweep (sweep_mask) read it: the
    // diagonal can be ENTERED through such a corner although its own bit is clear.
    uint32_t cg = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int dx, dy, dz;
      dir_delta(18 + j, dx, dy, dz);
      const int kd = 10 + (dy > 0 ? 2 : 0) + (dz > 0 ? 1 : 0);
      cg |= (((geo >> kd) & 1u) & ((allowed >> (18 + j)) & 1u)) << j;
    }
    corner_gate[v] = (uint8_t)cg;
  }
}
}  // namespace kh

extern "C" int kh_apply_voxel_graph(uint32_t* nbrmask, const uint32_t* graph, int64_t nvox, uint8_t* corner_gate, void* stream) {
  if (int rc = kh::require_device()) return rc;
  if (!nbrmask || !graph || nvox < 0) { kh::set_error("kh_apply_voxel_graph: bad arguments"); return KH_EINVAL; }
  if (nvox == 0) return KH_OK;
  hipLaunchKernelGGL(kh::apply_voxel_graph_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nbrmask, graph,
                     nvox, corner_gate);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_pdrf(const void* labels, int label_bytes, int64_t nvox, const int32_t* slot_of_label,
                       const kh_label_t* tasks, const float* dbf, float* daf, int log2_exponent, float scale, float* pdrf,
                       void* stream) {
  if (int rc = require_device()) return rc;
  int keep = 0;
  if (log2_exponent >= 0 && (log2_exponent & KH_PDRF_KEEP_OTHERS)) { keep = 1; log2_exponent &= ~KH_PDRF_KEEP_OTHERS; }
  if ((log2_exponent < 0 && log2_exponent != KH_PDRF_BASE && log2_exponent != KH_PDRF_FINISH) || log2_exponent > 15) {
    set_error("kh_pdrf: log2_exponent must be 0..15 (| KH_PDRF_KEEP_OTHERS), KH_PDRF_BASE or KH_PDRF_FINISH");
    return KH_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  KH_DISPATCH_LT(label_bytes, hipLaunchKernelGGL((pdrf_kernel<LT>), dim3(grid_for(nvox, 256)), dim3(256), 0, st,
                                                 (const LT*)labels, nvox, slot_of_label, tasks, dbf, daf, log2_exponent, scale,
                                                 pdrf, keep));
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_init_alive(const void* labels, int label_bytes, int64_t nvox, const int32_t* slot_of_label, uint8_t* alive,
                             void* stream) {
  if (int rc = require_device()) return rc;
  hipStream_t st = (hipStream_t)stream;
  KH_DISPATCH_LT(label_bytes, hipLaunchKernelGGL((init_alive_kernel<LT>), dim3(grid_for(nvox, 256)), dim3(256), 0, st,
                                                 (const LT*)labels, nvox, slot_of_label, alive));
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_fill_f32(float* p, int64_t n, float v, void* stream) {
  if (int rc = require_device()) return rc;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, n, v);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
extern "C" int kh_fill_u8(uint8_t* p, int64_t n, int v, void* stream) {
  if (int rc = require_device()) return rc;
  hipLaunchKernelGGL(fill_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, n, (uint8_t)v);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
extern "C" int kh_gather_f32(const float* src, const uint32_t* idx, int64_t n, float* out, void* stream) {
  if (int rc = require_device()) return rc;
  hipLaunchKernelGGL(gather_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, n, out);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_zero2inf(float* f, int64_t n, void* stream) {
  if (int rc = require_device()) return rc;
  hipLaunchKernelGGL(zero2inf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, f, n);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
extern "C" int kh_inf2zero(float* f, int64_t n, void* stream) {
  if (int rc = require_device()) return rc;
  hipLaunchKernelGGL(inf2zero_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, f, n);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
extern "C" int kh_pdrf_field(const float* dbf, float* daf, int64_t n, float M, int log2_exponent, float scale, float max_daf,
                             float* out, void* stream) {
  if (int rc = require_device()) return rc;
  if ((log2_exponent < 0 && log2_exponent != KH_PDRF_BASE && log2_exponent != KH_PDRF_FINISH) || log2_exponent > 15) {
    set_error("kh_pdrf_field: log2_exponent must be 0..15, KH_PDRF_BASE or KH_PDRF_FINISH");
    return KH_EINVAL;
  }
  hipLaunchKernelGGL(pdrf_field_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dbf, daf, n, M, log2_exponent,
                     scale, max_daf, out);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
extern "C" int kh_find_target(const uint8_t* labels, const float* field, int64_t sx, int64_t sy, int64_t sz, uint64_t* out,
                              void* stream) {
  if (int rc = require_device()) return rc;
  if (!labels || !field || !out || sx <= 0 || sy <= 0 || sz <= 0 || sx * sy * sz >= (1ll << 32)) {
    set_error("kh_find_target: bad arguments");
    return KH_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  KH_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(uint64_t), st));
  hipLaunchKernelGGL(find_target_kernel, dim3(grid_for(sx * sy * sz, 256)), dim3(256), 0, st, labels, field, sx, sy, sz,
                     (unsigned long long*)out);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
extern "C" int kh_first_label(const uint8_t* labels, int64_t n, uint64_t* out, void* stream) {
  if (int rc = require_device()) return rc;
  if (!labels || !out || n < 0) { set_error("kh_first_label: bad arguments"); return KH_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  KH_HIP_CHECK(hipMemsetAsync(out, 0xFF, sizeof(uint64_t), st));
  if (n > 0) {
    hipLaunchKernelGGL(first_label_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, labels, n, (unsigned long long*)out);
    KH_LAUNCH_CHECK();
  }
  return KH_OK;
}
extern "C" int kh_target_max(const uint32_t* list, const float* list_daf, const uint8_t* alive, int64_t n, uint64_t* out,
                             void* stream) {
  if (int rc = require_device()) return rc;
  if (n < 0 || n >= (1ll << 32)) { set_error("kh_target_max: bad list length"); return KH_EINVAL; }
  hipLaunchKernelGGL(target_max_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, list, list_daf, alive, (uint32_t)n,
                     (unsigned long long*)out);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
