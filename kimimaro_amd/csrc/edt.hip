// edt.hip -- a1: multi-label anisotropic exact Euclidean distance transform for gfx950.
//
// Replaces edt.edt as called at kimimaro/intake.py:178-183 / kimimaro/trace.py:112-117
// (third-party package `edt`, source not in the reference tree; semantics restated in
// oracle/kimi_oracle.c:ko_edt which this file matches bit for bit).
//
// MI355X-first design (not the CPU's sequential parabolic-envelope stack):
//   * x pass: one wave per row; a ballot per 64 voxels turns "a run starts here" into a bit mask, every voxel
//     finds its nearest label change on either side with clz/ctz on those words -- O(1) per voxel, one
//     coalesced read of the labels, one coalesced write of the squared distance.
//   * y and z pass: lanes along x (every access of the pass is a coalesced 256-B row segment), a tile of rows
//     staged in LDS together with per-column run masks, and an exact window search per voxel:
//     best = min(best, f[j] + (w*k)^2) walking outward until (w*k)^2 >= best or the same-label segment ends.
//     The window is ~sqrt(best)/w voxels, i.e. the local object radius.  The minimum is exact over the float
//     expressions, no envelope intersections, no sequential dependency between voxels.
//   * block -> tile mapping is XCD aware: the 8 XCDs (block b runs on XCD b % 8) each get a
//     contiguous 1/8 of the volume so the rows a block re-reads live in its own L2.
// Algorithmic bytes = L + 4 (x pass) and L + 8 (y, z pass) per voxel; measured HBM traffic matches them, the
// y / z passes are bound by instruction issue.
#include "common.h"

namespace kh {

// x pass: one wave per row, no workgroup barrier.  The wave reads the row once in 64-voxel chunks (the labels of
// the first 8 chunks stay in registers), a ballot per chunk gives the "a run starts here" word, and every voxel
// then finds the nearest run start on either side with clz / ctz on those words.
#define KH_EDT_XCACHE 8
template <typename LT>
__device__ __forceinline__ float edt_x_voxel(const unsigned long long* words, int nwords, int x, int sx, float w,
                                             int black_border) {
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
  if (d < 0) return KH_INF;
  const float dd = w * (float)d;
  return dd * dd;
}

template <typename LT>
__global__ __launch_bounds__(256) void edt_x_kernel(const LT* __restrict__ lab, float* __restrict__ out,
                                                    int sx, int64_t nrows, float w, int black_border) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nwords = (sx + 63) >> 6;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long* words = reinterpret_cast<unsigned long long*>(smem) + wave * nwords;  // this wave's row
  // XCD-aware row assignment: XCD c (= blockIdx.x % 8) walks a contiguous 1/8 of the rows; the 4 waves of a
  // workgroup take 4 consecutive rows
  const int64_t nblk = gridDim.x;
  const int64_t per_xcd = (nblk + 7) / 8;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t stride = per_xcd * 8;
  for (int64_t row = logical * 4 + wave; row < nrows; row += stride * 4) {
    const LT* __restrict__ r = lab + row * sx;
    float* __restrict__ o = out + row * sx;
    // 1. boundary flags -> this wave's words
    uint32_t cached[KH_EDT_XCACHE];
    uint32_t prev_last = 0;  // label of the last voxel of the previous chunk
#pragma unroll
    for (int c = 0; c < KH_EDT_XCACHE; c++) {
      if (c < nwords) {
        const int x = (c << 6) + lane;
        const uint32_t L = x < sx ? (uint32_t)r[x] : 0u;
        uint32_t Lm = (uint32_t)__shfl_up((int)L, 1);
        if (lane == 0) Lm = prev_last;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(x < sx && x > 0 && L != Lm);
        if (lane == 0) words[c] = m;
        prev_last = (uint32_t)__builtin_amdgcn_readlane((int)L, 63);
        cached[c] = L;
      }
    }
    for (int c = KH_EDT_XCACHE; c < nwords; c++) {
      const int x = (c << 6) + lane;
      const uint32_t L = x < sx ? (uint32_t)r[x] : 0u;
      uint32_t Lm = (uint32_t)__shfl_up((int)L, 1);
      if (lane == 0) Lm = prev_last;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(x < sx && x > 0 && L != Lm);
      if (lane == 0) words[c] = m;
      prev_last = (uint32_t)__builtin_amdgcn_readlane((int)L, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // 2. nearest label change on both sides
#pragma unroll
    for (int c = 0; c < KH_EDT_XCACHE; c++) {
      if (c < nwords) {
        const int x = (c << 6) + lane;
        if (x < sx) o[x] = cached[c] != 0u ? edt_x_voxel<LT>(words, nwords, x, sx, w, black_border) : 0.0f;
      }
    }
    for (int c = KH_EDT_XCACHE; c < nwords; c++) {
      const int x = (c << 6) + lane;
      if (x < sx) o[x] = r[x] != 0 ? edt_x_voxel<LT>(words, nwords, x, sx, w, black_border) : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();  // the words are rewritten by the next row
  }
}

// x pass, round 5: rows of up to 64 * NW voxels (NW = 8 or 16: every BASELINE volume) without LDS and without the per-voxel word
// scans.  A ballot is a scalar, so the NW "a run starts here" words of the row live in SGPRs; for every chunk the nearest run
// start BEFORE the chunk and the nearest one AFTER it are scalars too (two short scalar loops per row), and a voxel needs one
// clz and one ctz on its own chunk's word, falling back to those two scalars when its side of the word is empty.  All loads of
// the row are issued before the first is consumed.  Same integers, same float operations as edt_x_voxel: bit identical.
template <typename LT, int NW>
__global__ __launch_bounds__(256) void edt_x_rows_kernel(const LT* __restrict__ lab, float* __restrict__ out,
                                                         int sx, int64_t nrows, float w, int black_border) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nblk = gridDim.x;
  const int64_t per_xcd = (nblk + 7) / 8;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t stride = per_xcd * 8;
  const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);   // bits at or below this lane
  // the labels of the NEXT row are requested before this row is worked on (a wave has one row in flight otherwise: 1 KiB)
  uint32_t Ln[NW];
  {
    const int64_t row0 = logical * 4 + wave;
#pragma unroll
    for (int c = 0; c < NW; c++) {
      const int x = (c << 6) + lane;
      Ln[c] = (row0 < nrows && x < sx) ? (uint32_t)lab[row0 * sx + x] : 0u;
    }
  }
  for (int64_t row = logical * 4 + wave; row < nrows; row += stride * 4) {
    float* __restrict__ o = out + row * sx;
    uint32_t L[NW];
    const int64_t rown = row + stride * 4;
#pragma unroll
    for (int c = 0; c < NW; c++) {
      const int x = (c << 6) + lane;
      L[c] = Ln[c];
      Ln[c] = (rown < nrows && x < sx) ? (uint32_t)lab[rown * sx + x] : 0u;
    }
    unsigned long long word[NW];
    uint32_t prev_last = 0;
#pragma unroll
    for (int c = 0; c < NW; c++) {
      const int x = (c << 6) + lane;
      uint32_t Lm = (uint32_t)__shfl_up((int)L[c], 1);
      if (lane == 0) Lm = prev_last;
      word[c] = __builtin_amdgcn_ballot_w64(x < sx && x > 0 && L[c] != Lm);
      prev_last = (uint32_t)__builtin_amdgcn_readlane((int)L[c], 63);
    }
    // nearest run start before / after every chunk (scalars; -1 = none)
    int before[NW], after[NW];
    int run = -1;
#pragma unroll
    for (int c = 0; c < NW; c++) {
      before[c] = run;
      if (word[c]) run = (c << 6) + 63 - __clzll((long long)word[c]);
    }
    run = -1;
#pragma unroll
    for (int c = NW - 1; c >= 0; c--) {
      after[c] = run;
      if (word[c]) run = (c << 6) + __ffsll((long long)word[c]) - 1;
    }
#pragma unroll
    for (int c = 0; c < NW; c++) {
      const int x = (c << 6) + lane;
      if (x >= sx) continue;
      // left: the highest run start p <= x (the differing voxel is p - 1); right: the lowest run start q > x
      const unsigned long long ml = word[c] & le, mr = word[c] & ~le;
      int dl = -1, dr = -1;
      if (ml) dl = lane - (63 - __clzll((long long)ml)) + 1;
      else if (before[c] >= 0) dl = x - before[c] + 1;
      else if (black_border) dl = x + 1;
      if (mr) dr = (__ffsll((long long)mr) - 1) - lane;
      else if (after[c] >= 0) dr = after[c] - x;
      else if (black_border) dr = sx - x;
      int d = dl;
      if (d < 0 || (dr >= 0 && dr < d)) d = dr;
      float v = 0.0f;
      if (L[c] != 0u) {
        const float dd = w * (float)d;
        v = d < 0 ? KH_INF : dd * dd;
      }
      o[x] = v;
    }
  }
}

// y / z pass: best = min over the same-label segment of f[j] + (w*(i-j))^2, clamped by the segment
// ends (label change, or the volume border when black_border).  The search walks outward and stops as
// soon as (w*k)^2 >= best -- nothing farther can improve the minimum -- so the result is the exact
// minimum over the float expressions in any visiting order (== oracle ko_edt_axis).
//
// A workgroup (64 x 4 threads) owns 64 lanes along x times T = 64 positions along the axis.
//  * stage: rows [A0-H, A0+T+H) of f go to LDS (H = 32, 32 KiB, bank = lane: conflict free); at the same
//    time every column gets a 128-bit "label changes at this row" mask and a "background" mask.
//  * limits: the number of same-label rows below / above an output is a clz / ctz on its column's mask --
//    no label is loaded or compared inside the search loop.
//  * search: probes read LDS only.  Runs or windows that leave the staged rows (objects wider than H
//    voxels) continue in a slow path on global memory.
// Every f element is fetched from L2/HBM (T+2H)/T = 2x instead of 2*window times.
#define KH_EDT_T 64

// 128-bit mask helpers; the bit position p is wave uniform (a row index), so the shifts are by scalars and
// the two cases of each shift are scalar branches; the data dependent part is select-only.
__device__ __forceinline__ int zeros_down(unsigned long long lo, unsigned long long hi, int p) {
  // number of consecutive zero bits at p, p-1, ... ; p+1 when none is set down to bit 0  (0 <= p <= 127)
  const int s = 127 - p;  // bits p..0 moved to the top of a 128-bit word
  unsigned long long xh, xl;
  if (s >= 64) { xh = lo << (s - 64); xl = 0; }
  else if (s == 0) { xh = hi; xl = lo; }
  else { xh = (hi << s) | (lo >> (64 - s)); xl = lo << s; }
  const int nz = xh ? __clzll((long long)xh) : 64 + (xl ? __clzll((long long)xl) : 64);
  return min(nz, p + 1);
}
__device__ __forceinline__ int zeros_up(unsigned long long lo, unsigned long long hi, int p) {
  // number of consecutive zero bits at p, p+1, ... ; 128-p when none is set up to bit 127  (0 <= p <= 128)
  if (p >= 128) return 0;
  unsigned long long xh, xl;
  if (p >= 64) { xl = hi >> (p - 64); xh = 0; }
  else if (p == 0) { xl = lo; xh = hi; }
  else { xl = (lo >> p) | (hi << (64 - p)); xh = hi >> p; }
  const int nz = xl ? __ffsll((long long)xl) - 1 : 64 + (xh ? __ffsll((long long)xh) - 1 : 64);
  return min(nz, 128 - p);
}
// min of two floats that are known to be >= +0 (or +inf): the order of the bit patterns is the order of the values
__device__ __forceinline__ float minpos(float a, float b) {
  return __uint_as_float(min(__float_as_uint(a), __float_as_uint(b)));
}
__device__ __forceinline__ bool bit128(unsigned long long lo, unsigned long long hi, int p) {
  return (((p < 64) ? (lo >> p) : (hi >> (p - 64))) & 1ull) != 0;
}

template <typename LT, bool LAST, int KH_EDT_H>
__global__ __launch_bounds__(256) void edt_axis_kernel(const LT* __restrict__ lab, const float* __restrict__ fin,
                                                       float* __restrict__ fout, int sx, int n, int64_t astride,
                                                       int m, int64_t ostride, float w, int black_border) {
  constexpr int KH_EDT_ROWS = KH_EDT_T + 2 * KH_EDT_H;  // staged rows: <= 128 (column masks), multiple of 4
  static_assert(KH_EDT_ROWS <= 128 && KH_EDT_ROWS % 4 == 0 && KH_EDT_ROWS / 4 <= 32, "tile shape");
  __shared__ float tile[KH_EDT_ROWS * 64];
  __shared__ unsigned int part[2][4][64];  // [label change | background][ly][lx]: ROWS/4 rows of a column mask each
  __shared__ __attribute__((aligned(16))) float tsq[KH_EDT_ROWS + 8];  // [0] = +inf, [k+2] = (w*k)^2 for k >= 0
  // volume seen as [sx][n along axis][m others]: index = x + a*astride + o*ostride ; blockDim = (64, 4)
  const int xt = (sx + 63) >> 6, at = (n + KH_EDT_T - 1) / KH_EDT_T;
  const int64_t ntiles = (int64_t)xt * at * m;
  const int64_t nblk = gridDim.x;
  const int64_t per_xcd = (nblk + 7) / 8;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t stride = per_xcd * 8;
  // blockDim = (64, 4): a wave is one row of the block, so ly is wave uniform -- say so (SGPR), which turns
  // every row-index test and shift below into scalar code
  const int lx = threadIdx.x, ly = __builtin_amdgcn_readfirstlane(threadIdx.y);
  for (int k = threadIdx.y * 64 + threadIdx.x; k < KH_EDT_ROWS + 8; k += 256) {
    const float d = w * (float)(k - 2);
    tsq[k] = k == 0 ? KH_INF : d * d;
  }
  for (int64_t t = logical; t < ntiles; t += stride) {
    const int tx = (int)(t % xt);
    const int64_t rr = t / xt;
    const int ta = (int)(rr % at);
    const int o = (int)(rr / at);
    const int x = (tx << 6) + lx;
    const int A0 = ta * KH_EDT_T;
    const int64_t base = x + (int64_t)o * ostride;
    __syncthreads();  // previous tile fully consumed
    {
      // thread (lx, ly) stages KH_EDT_ROWS / 4 consecutive rows: the label of the previous row
      // is the previous iteration's register, so labels are read once (+1 row per thread).
      const int rbeg = ly * (KH_EDT_ROWS / 4);
      const int pbeg = A0 - KH_EDT_H + rbeg;
      LT Lp = 0;
      if (x < sx && pbeg - 1 >= 0 && pbeg - 1 < n) Lp = lab[base + (int64_t)(pbeg - 1) * astride];
      unsigned int cm = 0, bm = 0;
#ifndef KH_EDT_STAGE_UNROLL
#define KH_EDT_STAGE_UNROLL 4
#endif
#pragma unroll KH_EDT_STAGE_UNROLL
      for (int j = 0; j < KH_EDT_ROWS / 4; j++) {
        const int pos = pbeg + j;
        const bool valid = x < sx && pos >= 0 && pos < n;
        float v = KH_INF;
        LT L = 0;
        if (valid) {
          const int64_t q = base + (int64_t)pos * astride;
          v = fin[q];
          L = lab[q];
        }
        tile[(rbeg + j) * 64 + lx] = v;
        // a row outside the volume, or the first row of the volume, or a label change, ends every run
        const bool chg = !valid || pos == 0 || L != Lp;
        const bool bg = !valid || L == 0;
        cm |= (chg ? 1u : 0u) << j;
        bm |= (bg ? 1u : 0u) << j;
        Lp = L;
      }
      part[0][ly][lx] = cm;
      part[1][ly][lx] = bm;
    }
    __syncthreads();
    if (x >= sx) continue;
    // thread ly contributed rows [RPT*ly, RPT*ly + RPT) of the column masks
    constexpr int RPT = KH_EDT_ROWS / 4;
    unsigned __int128 cmask = 0, bmask = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      cmask |= (unsigned __int128)part[0][j][lx] << (j * RPT);
      bmask |= (unsigned __int128)part[1][j][lx] << (j * RPT);
    }
    const unsigned long long c_lo = (unsigned long long)cmask, c_hi = (unsigned long long)(cmask >> 64);
    const unsigned long long b_lo = (unsigned long long)bmask, b_hi = (unsigned long long)(bmask >> 64);
    for (int al = ly; al < KH_EDT_T; al += 4) {
      const int a = A0 + al;
      if (a >= n) break;
      const int r0 = al + KH_EDT_H;  // my row in the tile (wave uniform)
      const int64_t i = base + (int64_t)a * astride;
      float best = 0.0f;
      if (!bit128(b_lo, b_hi, r0)) {
        const float* __restrict__ pc = &tile[r0 * 64 + lx];
        best = pc[0];
        // same-label rows below / above inside the staged rows
        int nl = zeros_down(c_lo, c_hi, r0);
        int nr = zeros_up(c_lo, c_hi, r0 + 1);
        const bool lunk = nl > r0;                     // no change found down to the first staged row
        const bool runk = r0 + 1 + nr >= KH_EDT_ROWS;  // none up to the last staged row
        nl = min(nl, r0);
        nr = min(nr, KH_EDT_ROWS - 1 - r0);
        // a segment end that is known here is a candidate: a differing voxel, or the volume border when
        // black_border.  (Rows outside the volume are staged as +inf, so probing them is harmless.)
        const bool lc = !lunk && (a - nl - 1 >= 0 || black_border);
        const bool rc = !runk && (a + nr + 1 < n || black_border);
        const float tl = tsq[lc ? nl + 3 : 0], tr = tsq[rc ? nr + 3 : 0];   // tsq[0] = +inf, tsq[k+2] = (w*k)^2
        best = minpos(best, minpos(tl, tr));
        // Both sides are probed together for k <= kb.  A side that ended in a candidate needs nothing beyond
        // its end: the candidate (w*(end+1))^2 already bounds everything farther away.  A side without one
        // (volume border, or a run leaving the staged rows) is probed up to the edge of the staged rows.
        const int kb = min(lc ? nl : r0, rc ? nr : KH_EDT_ROWS - 1 - r0);
        // Groups of 4 steps, run as a wave-uniform loop (k is a scalar): a lane takes part while its next 4 rows
        // are inside kb and (w*k)^2 < best.  (w*k)^2 comes from the LDS table (broadcast reads), the 8 probes are
        // fetched with constant offsets from one base address before any is consumed, no clamps, no per-step
        // branches.  A probe farther than the bound cannot lower `best` (f >= 0), so nothing has to be undone
        // when the bound is crossed inside a group.  All values are >= +0, so min is taken on the bit patterns.
        int kn = 1;  // first step this lane has not probed yet
        float tx = tsq[3];  // (w*k)^2 of the group's first step; the next group's arrives with this group's reads
        for (int k = 1;; k += 4) {
          const bool can = (k + 3 <= kb) && (tx < best);  // once false it stays false
          if (!__builtin_amdgcn_ballot_w64(can)) break;
          if (can) {
            const float4 t = *reinterpret_cast<const float4*>(&tsq[k + 3]);  // steps k+1, k+2, k+3 and k+4
            const float* __restrict__ pl = pc - k * 64;
            const float* __restrict__ pr = pc + k * 64;
            const float l0 = pl[0], l1 = pl[-64], l2 = pl[-128], l3 = pl[-192];
            const float r0v = pr[0], r1 = pr[64], r2 = pr[128], r3 = pr[192];
            // min(l + t, r + t) == min(l, r) + t bit for bit (rounding is monotone): one add per step
            best = minpos(best, minpos(minpos(l0, r0v) + tx, minpos(l1, r1) + t.x));
            best = minpos(best, minpos(minpos(l2, r2) + t.y, minpos(l3, r3) + t.z));
            tx = t.w;
            kn = k + 4;
          }
        }
        // at most 3 steps are left below kb for a lane that was stopped by kb (a lane stopped by the bound gains
        // nothing from them, and loses nothing)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int kq = kn + j;
          if (kq <= kb) best = minpos(best, minpos(pc[-kq * 64], pc[kq * 64]) + tsq[kq + 2]);
        }
        const bool open = tsq[kb + 3] < best;  // would step kb + 1 still be inside the bound?
        int k = kb + 1;
        if (open && !(lc && rc)) {
          // rare: one side has no end inside the staged rows (objects wider than H voxels) or ends at the volume
          // border.  Keep walking: staged rows first, then global memory with the labels checked.
          const LT L = lab[i];
          bool lo = lunk, ro = runk;
          for (;; k++) {
            const bool lin = k <= nl, rin = k <= nr;
            if (!(lin || rin || lo || ro)) break;
            const float d = w * (float)k;
            const float tt = d * d;
            if (tt >= best) break;
            if (lin) best = fminf(best, pc[-k * 64] + tt);
            else if (lo) {
              const int j = a - k;
              if (j < 0) { lo = false; if (black_border) best = tt; }
              else {
                const int64_t q = base + (int64_t)j * astride;
                if (lab[q] != L) { lo = false; best = tt; }
                else best = fminf(best, fin[q] + tt);
              }
            }
            if (rin) best = fminf(best, pc[k * 64] + tt);
            else if (ro) {
              const int j = a + k;
              if (j >= n) { ro = false; if (black_border) best = fminf(best, tt); }
              else {
                const int64_t q = base + (int64_t)j * astride;
                if (lab[q] != L) { ro = false; best = fminf(best, tt); }
                else best = fminf(best, fin[q] + tt);
              }
            }
          }
        }
        if (LAST) best = sqrtf(best);
      }
      fout[i] = best;
    }
  }
}

template <bool LAST>
__global__ void edt_finish_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = LAST ? sqrtf(in[i]) : in[i];
}

template <typename LT>
static int edt_impl(const LT* lab, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                    int black_border, float* ws, float* out, hipStream_t st, hipEvent_t* ev = nullptr) {
  const int64_t nrows = sy * sz;
  const int64_t nvox = sx * nrows;
  // an axis beyond the array's dimensionality is not an axis: edt.edt on a 2-D plane is a 2-D transform, the
  // border of a missing axis does not exist (kimimaro/intake.py:568 calls it on the faces of the volume)
  const bool do_y = ndim >= 2 && ((sy > 1) || black_border);
  const bool do_z = ndim >= 3 && ((sz > 1) || black_border);
  // ping-pong so that the final pass lands in `out`
  const int npass = 1 + (do_y ? 1 : 0) + (do_z ? 1 : 0);
  float* bufs[2] = {out, ws};
  int cur = (npass % 2 == 1) ? 0 : 1;  // buffer the x pass writes
  {
    const int nwords = (int)((sx + 63) >> 6);
    const int64_t need = (nrows + 3) / 4;  // 4 rows (one per wave) per workgroup and step
    int64_t grid = need < 8192 ? need : 8192;
    grid = (grid + 7) & ~7ll;  // the XCD remap needs a multiple of 8 blocks
    if (ev) KH_HIP_CHECK(hipEventRecord(ev[0], st));
    if (nwords <= 8)
      hipLaunchKernelGGL((edt_x_rows_kernel<LT, 8>), dim3((unsigned)grid), dim3(256), 0, st, lab, bufs[cur], (int)sx, nrows, wx, black_border);
    else if (nwords <= 16)
      hipLaunchKernelGGL((edt_x_rows_kernel<LT, 16>), dim3((unsigned)grid), dim3(256), 0, st, lab, bufs[cur], (int)sx, nrows, wx, black_border);
    else
      hipLaunchKernelGGL((edt_x_kernel<LT>), dim3((unsigned)grid), dim3(256), 4 * nwords * 8, st, lab, bufs[cur],
                         (int)sx, nrows, wx, black_border);
    KH_LAUNCH_CHECK();
    if (ev) KH_HIP_CHECK(hipEventRecord(ev[1], st));
  }
  auto axis = [&](int n, int64_t astride, int m, int64_t ostride, float w, bool last) -> int {
    const int64_t ntiles = ((sx + 63) / 64) * (int64_t)((n + KH_EDT_T - 1) / KH_EDT_T) * m;
    int64_t grid = ntiles < 16384 ? ntiles : 16384;
    grid = (grid + 7) & ~7ll;  // the XCD remap needs a multiple of 8 blocks
    const float* fin = bufs[cur];
    float* fout = bufs[cur ^ 1];
    // halo: a search needs k <= sqrt(best)/w rows either side, so the axis with the coarse voxel pitch gets the small halo
    // (less re-reading); anything that does not fit continues on global memory, so this only affects speed.  Measured on
    // the 512^3 bench volume, fine axis: H = 8 / 12 / 16 / 20 / 24 / 28 / 32 -> 1.18 / 1.10 / 1.12 / 1.06 / 1.05 / 1.24 / 1.23 ms;
    // coarse axis: H = 8 / 12 -> 0.62 / 0.69 ms
    const float wmin = fminf(wx, fminf(wy, wz));
    const bool small_halo = w >= 2.0f * wmin;
#define KH_AXIS_LAUNCH(LASTV, HV) hipLaunchKernelGGL((edt_axis_kernel<LT, LASTV, HV>), dim3((unsigned)grid), dim3(64, 4), 0, st, \
                                                     lab, fin, fout, (int)sx, n, astride, m, ostride, w, black_border)
    if (last) { if (small_halo) KH_AXIS_LAUNCH(true, 8); else KH_AXIS_LAUNCH(true, 24); }
    else { if (small_halo) KH_AXIS_LAUNCH(false, 8); else KH_AXIS_LAUNCH(false, 24); }
#undef KH_AXIS_LAUNCH
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

extern "C" int kh_edt_nd(const void* labels, int label_bytes, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                         float wz, int black_border, float* workspace, float* out, void* stream) {
  if (int rc = kh::require_device()) return rc;
  if (!labels || !out || !workspace || sx <= 0 || sy <= 0 || sz <= 0 || sx * sy * sz >= (1ll << 32)) {
    kh::set_error("kh_edt: bad arguments (null pointer, empty volume or >= 2^32 voxels)");
    return KH_EINVAL;
  }
  if (ndim < 1 || ndim > 3 || (ndim < 3 && sz != 1) || (ndim < 2 && sy != 1)) {
    kh::set_error("kh_edt_nd: ndim must be 1..3 and the axes beyond it must have extent 1");
    return KH_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (label_bytes) {
    case 1: return kh::edt_impl<uint8_t>((const uint8_t*)labels, ndim, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st);
    case 2: return kh::edt_impl<uint16_t>((const uint16_t*)labels, ndim, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st);
    case 4: return kh::edt_impl<uint32_t>((const uint32_t*)labels, ndim, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st);
    default: kh::set_error("kh_edt: label_bytes must be 1, 2 or 4"); return KH_EINVAL;
  }
}

extern "C" int kh_edt(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                      float wz, int black_border, float* workspace, float* out, void* stream) {
  return kh_edt_nd(labels, label_bytes, 3, sx, sy, sz, wx, wy, wz, black_border, workspace, out, stream);
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
    case 1: rc = kh::edt_impl<uint8_t>((const uint8_t*)labels, 3, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st, ev); break;
    case 2: rc = kh::edt_impl<uint16_t>((const uint16_t*)labels, 3, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st, ev); break;
    case 4: rc = kh::edt_impl<uint32_t>((const uint32_t*)labels, 3, sx, sy, sz, wx, wy, wz, black_border, workspace, out, st, ev); break;
    default: kh::set_error("kh_edt_timed: label_bytes must be 1, 2 or 4"); rc = KH_EINVAL;
  }
  if (rc == KH_OK) {
    KH_HIP_CHECK(hipEventSynchronize(ev[3]));
    for (int i = 0; i < 3; i++) { ms3[i] = 0.0f; (void)hipEventElapsedTime(&ms3[i], ev[i], ev[i + 1]); }
  }
  for (int i = 0; i < 4; i++) (void)hipEventDestroy(ev[i]);
  return rc;
}
