// edt.hip -- a1: multi-label anisotropic exact Euclidean distance transform for gfx950.
//
// Replaces edt.edt as called at kimimaro/intake.py:178-183 / kimimaro/trace.py:112-117
// (third-party package `edt`, source not in the reference tree; semantics restated in
// oracle/kimi_oracle.c:ko_edt which this file matches bit for bit).
//
// MI355X-first design (not the CPU's sequential parabolic-envelope stack):
//   * x pass: one wave ballot per 64 voxels turns "label changes here" into a bit mask
//     staged in LDS; every voxel finds its nearest label change on either side with
//     clz/ctz on those words -- O(1) per voxel, one coalesced read of the labels, one
//     coalesced write of the squared distance.
//   * y and z pass: one thread per voxel, lanes along x (so every access of the pass is a
//     coalesced 256-B row segment).  Each voxel searches outward along the axis,
//     best = min(best, f[j] + (w*k)^2), and stops as soon as (w*k)^2 >= best or the
//     same-label segment ends.  The search window is ~sqrt(best)/w voxels, i.e. the local
//     object radius: thin neurites close it in a handful of steps, and re-reads hit L1/L2
//     (the +-k rows are shared by the 4 neighbouring rows handled by the same workgroup).
//     The minimum is exact over the float expressions, no envelope intersections, no
//     sequential dependency between voxels.
//   * block -> tile mapping is XCD aware: the 8 XCDs (block b runs on XCD b % 8) each get a
//     contiguous 1/8 of the volume so the rows a block re-reads live in its own L2.
// Memory bound: algorithmic bytes = L + 4 (x pass) and L + 8 (y, z pass) per voxel.
#include "common.h"

namespace kh {

template <typename LT>
__global__ __launch_bounds__(256) void edt_x_kernel(const LT* __restrict__ lab, float* __restrict__ out,
                                                    int sx, int64_t nrows, float w, int black_border) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* words = reinterpret_cast<unsigned long long*>(smem);  // ceil(sx/64)+... per row
  const int nwords = (sx + 63) >> 6;
  const int lane = threadIdx.x & 63;
  // XCD-aware row assignment: XCD c (= blockIdx.x % 8) walks rows [c*chunk, (c+1)*chunk)
  const int64_t nblk = gridDim.x;
  const int64_t per_xcd = (nblk + 7) / 8;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t stride = per_xcd * 8;
  for (int64_t row = logical; row < nrows; row += stride) {
    const LT* __restrict__ r = lab + row * sx;
    float* __restrict__ o = out + row * sx;
    // 1. boundary flags -> LDS words
    for (int x0 = 0; x0 < nwords * 64; x0 += 256) {
      const int x = x0 + threadIdx.x;
      bool flag = false;
      if (x < sx && x > 0) flag = (r[x] != r[x - 1]);
      const unsigned long long m = __ballot(flag);
      if (lane == 0 && (x >> 6) < nwords) words[x >> 6] = m;
    }
    __syncthreads();
    // 2. nearest label change on both sides
    for (int x = threadIdx.x; x < sx; x += 256) {
      const LT L = r[x];
      float res = 0.0f;
      if (L != 0) {
        const int wi = x >> 6, bit = x & 63;
        // left: highest flag position p <= x  (run starts at p, differing voxel at p-1)
        int dl = -1;  // -1 = none
        {
          unsigned long long m = words[wi] & ((bit == 63) ? ~0ull : ((2ull << bit) - 1ull));
          int k = wi;
          while (m == 0 && k > 0) { k--; m = words[k]; }
          if (m != 0) {
            const int p = (k << 6) + (63 - __clzll((long long)m));
            dl = x - p + 1;
          } else if (black_border) dl = x + 1;
        }
        int dr = -1;
        {
          unsigned long long m = (bit == 63) ? 0ull : (words[wi] & ~((2ull << bit) - 1ull));
          int k = wi;
          while (m == 0 && k + 1 < nwords) { k++; m = words[k]; }
          if (m != 0) {
            const int q = (k << 6) + (__ffsll((long long)m) - 1);
            dr = q - x;
          } else if (black_border) dr = sx - x;
        }
        int d = dl;
        if (d < 0 || (dr >= 0 && dr < d)) d = dr;
        if (d < 0) res = KH_INF;
        else {
          const float dd = w * (float)d;
          res = dd * dd;
        }
      }
      o[x] = res;
    }
    __syncthreads();
  }
}

// y / z pass.  blockDim = (64, 4): 64 lanes along x; every thread owns R consecutive positions along the
// axis and walks outward from its group, so each row it loads (one coalesced 256-B segment per wave)
// feeds R minima.  For output r of the group, the row at distance s to the left of the group is at
// distance k = s + r, the one to the right at k = s + R-1-r.  A side of an output closes when its
// same-label segment ends or when (w*k)^2 >= best (nothing farther can improve the minimum), so the
// result is the exact minimum over the float expressions in any visiting order (== oracle ko_edt_axis).
template <typename LT, bool LAST, int R>
__global__ __launch_bounds__(256) void edt_axis_kernel(const LT* __restrict__ lab, const float* __restrict__ fin,
                                                       float* __restrict__ fout, int sx, int n, int64_t astride,
                                                       int m, int64_t ostride, float w, int black_border) {
  // volume seen as [sx][n along axis][m others]: index = x + a*astride + o*ostride
  const int xt = (sx + 63) >> 6, at = (n + 4 * R - 1) / (4 * R);
  const int64_t ntiles = (int64_t)xt * at * m;
  const int64_t nblk = gridDim.x;
  const int64_t per_xcd = (nblk + 7) / 8;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t stride = per_xcd * 8;
  for (int64_t t = logical; t < ntiles; t += stride) {
    const int tx = (int)(t % xt);
    const int64_t rr = t / xt;
    const int ta = (int)(rr % at);
    const int o = (int)(rr / at);
    const int x = (tx << 6) + threadIdx.x;
    const int a0 = (ta * 4 + threadIdx.y) * R;
    if (x >= sx || a0 >= n) continue;
    const int64_t base = x + (int64_t)o * ostride;
    LT L[R];
    float f[R], best[R];
    bool lo[R], ro[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const bool valid = a0 + r < n;
      const int64_t i = base + (int64_t)(valid ? a0 + r : a0) * astride;
      L[r] = valid ? lab[i] : (LT)0;
      f[r] = valid ? fin[i] : 0.0f;
      best[r] = (L[r] != 0) ? f[r] : 0.0f;
      lo[r] = ro[r] = (L[r] != 0);
    }
    // candidates inside the group, nearest first
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
      for (int d = 1; d < R; d++) {
        const float dd = w * (float)d;
        const float tt = dd * dd;
        if (r - d >= 0 && lo[r]) {
          if (tt >= best[r]) lo[r] = false;
          else if (L[r - d] != L[r]) { lo[r] = false; best[r] = tt; }
          else { const float c = f[r - d] + tt; if (c < best[r]) best[r] = c; }
        }
        if (r + d < R && ro[r]) {
          if (tt >= best[r]) ro[r] = false;
          else if (a0 + r + d >= n) { ro[r] = false; if (black_border) best[r] = tt; }
          else if (L[r + d] != L[r]) { ro[r] = false; best[r] = tt; }
          else { const float c = f[r + d] + tt; if (c < best[r]) best[r] = c; }
        }
      }
    }
    bool anyl = false, anyr = false;
#pragma unroll
    for (int r = 0; r < R; r++) { anyl |= lo[r]; anyr |= ro[r]; }
    for (int s = 1; anyl || anyr; s++) {
      if (anyl) {
        const int j = a0 - s;
        LT Lj = 0;
        float fj = 0.0f;
        if (j >= 0) { const int64_t q = base + (int64_t)j * astride; Lj = lab[q]; fj = fin[q]; }
        anyl = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (lo[r]) {
            const float dd = w * (float)(s + r);
            const float tt = dd * dd;
            if (tt >= best[r]) lo[r] = false;
            else if (j < 0) { lo[r] = false; if (black_border) best[r] = tt; }
            else if (Lj != L[r]) { lo[r] = false; best[r] = tt; }
            else { const float c = fj + tt; if (c < best[r]) best[r] = c; anyl = true; }
          }
        }
      }
      if (anyr) {
        const int j = a0 + R - 1 + s;
        LT Lj = 0;
        float fj = 0.0f;
        if (j < n) { const int64_t q = base + (int64_t)j * astride; Lj = lab[q]; fj = fin[q]; }
        anyr = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (ro[r]) {
            const float dd = w * (float)(s + R - 1 - r);
            const float tt = dd * dd;
            if (tt >= best[r]) ro[r] = false;
            else if (j >= n) { ro[r] = false; if (black_border) best[r] = tt; }
            else if (Lj != L[r]) { ro[r] = false; best[r] = tt; }
            else { const float c = fj + tt; if (c < best[r]) best[r] = c; anyr = true; }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (a0 + r < n) fout[base + (int64_t)(a0 + r) * astride] = LAST ? sqrtf(best[r]) : best[r];
    }
  }
}

template <bool LAST>
__global__ void edt_finish_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = LAST ? sqrtf(in[i]) : in[i];
}

template <typename LT>
static int edt_impl(const LT* lab, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                    int black_border, float* ws, float* out, hipStream_t st, hipEvent_t* ev = nullptr) {
  const int64_t nrows = sy * sz;
  const int64_t nvox = sx * nrows;
  const bool do_y = (sy > 1) || black_border;
  const bool do_z = (sz > 1) || black_border;
  // ping-pong so that the final pass lands in `out`
  const int npass = 1 + (do_y ? 1 : 0) + (do_z ? 1 : 0);
  float* bufs[2] = {out, ws};
  int cur = (npass % 2 == 1) ? 0 : 1;  // buffer the x pass writes
  {
    const int nwords = (int)((sx + 63) >> 6);
    int64_t grid = nrows < 8192 ? nrows : 8192;
    grid = (grid + 7) & ~7ll;  // the XCD remap needs a multiple of 8 blocks
    if (ev) KH_HIP_CHECK(hipEventRecord(ev[0], st));
    hipLaunchKernelGGL((edt_x_kernel<LT>), dim3((unsigned)grid), dim3(256), nwords * 8, st, lab, bufs[cur],
                       (int)sx, nrows, wx, black_border);
    KH_LAUNCH_CHECK();
    if (ev) KH_HIP_CHECK(hipEventRecord(ev[1], st));
  }
  auto axis = [&](int n, int64_t astride, int m, int64_t ostride, float w, bool last) -> int {
    constexpr int R = 4;
    const int64_t ntiles = ((sx + 63) / 64) * (int64_t)((n + 4 * R - 1) / (4 * R)) * m;
    int64_t grid = ntiles < 16384 ? ntiles : 16384;
    grid = (grid + 7) & ~7ll;  // the XCD remap needs a multiple of 8 blocks
    const float* fin = bufs[cur];
    float* fout = bufs[cur ^ 1];
    if (last)
      hipLaunchKernelGGL((edt_axis_kernel<LT, true, R>), dim3((unsigned)grid), dim3(64, 4), 0, st, lab, fin, fout, (int)sx, n,
                         astride, m, ostride, w, black_border);
    else
      hipLaunchKernelGGL((edt_axis_kernel<LT, false, R>), dim3((unsigned)grid), dim3(64, 4), 0, st, lab, fin, fout, (int)sx, n,
                         astride, m, ostride, w, black_border);
    KH_LAUNCH_CHECK();
    cur ^= 1;
    return KH_OK;
  };
  if (do_y) { int rc = axis((int)sy, sx, (int)sz, sx * sy, wy, !do_z); if (rc) return rc; }
  if (ev) KH_HIP_CHECK(hipEventRecord(ev[2], st));
  if (do_z) { int rc = axis((int)sz, sx * sy, (int)sy, sx, wz, true); if (rc) return rc; }
  if (ev) KH_HIP_CHECK(hipEventRecord(ev[3], st));
  if (!do_y && !do_z) {
    // 1-D input: x pass wrote `out` un-rooted; take the root in place
    hipLaunchKernelGGL((edt_finish_kernel<true>), dim3(1024), dim3(256), 0, st, out, out, nvox);
    KH_LAUNCH_CHECK();
  }
  return KH_OK;
}

}  // namespace kh

extern "C" int kh_edt(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                      float wz, int black_border, float* workspace, float* out, void* stream) {
  if (int rc = kh::require_device()) return rc;
  if (!labels || !out || !workspace || sx <= 0 || sy <= 0 || sz <= 0 || sx * sy * sz >= (1ll << 32)) {
    kh::set_error("kh_edt: bad arguments (null pointer, empty volume or >= 2^32 voxels)");
    return KH_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (label_bytes) {
    case 1: return kh::edt_impl<uint8_t>((const uint8_t*)labels, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st);
    case 2: return kh::edt_impl<uint16_t>((const uint16_t*)labels, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st);
    case 4: return kh::edt_impl<uint32_t>((const uint32_t*)labels, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st);
    default: kh::set_error("kh_edt: label_bytes must be 1, 2 or 4"); return KH_EINVAL;
  }
}

// Same as kh_edt, but brackets each pass with HIP events on `stream` and returns the three pass
// durations in milliseconds (x, y, z; 0 for a skipped pass).  Synchronises the stream.  Used by
// bench.py for the roofline line (the events sit on the stream the kernels are launched on).
extern "C" int kh_edt_timed(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                            float wz, int black_border, float* workspace, float* out, void* stream, float* ms3) {
  if (int rc = kh::require_device()) return rc;
  if (!labels || !out || !workspace || !ms3 || sx <= 0 || sy <= 0 || sz <= 0 || sx * sy * sz >= (1ll << 32)) {
    kh::set_error("kh_edt_timed: bad arguments");
    return KH_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t ev[4];
  for (int i = 0; i < 4; i++) KH_HIP_CHECK(hipEventCreate(&ev[i]));
  int rc;
  switch (label_bytes) {
    case 1: rc = kh::edt_impl<uint8_t>((const uint8_t*)labels, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st, ev); break;
    case 2: rc = kh::edt_impl<uint16_t>((const uint16_t*)labels, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st, ev); break;
    case 4: rc = kh::edt_impl<uint32_t>((const uint32_t*)labels, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st, ev); break;
    default: kh::set_error("kh_edt_timed: label_bytes must be 1, 2 or 4"); rc = KH_EINVAL;
  }
  if (rc == KH_OK) {
    KH_HIP_CHECK(hipEventSynchronize(ev[3]));
    for (int i = 0; i < 3; i++) { ms3[i] = 0.0f; (void)hipEventElapsedTime(&ms3[i], ev[i], ev[i + 1]); }
  }
  for (int i = 0; i < 4; i++) (void)hipEventDestroy(ev[i]);
  return rc;
}
