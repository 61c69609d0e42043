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
//     staged in LDS as two views with the segment ends folded into the data, and an exact window search per voxel:
//     best = min(best, view[j] + (w*k)^2) walking outward until (w*k)^2 >= best -- no limits, masks or labels in the
//     search; windows wider than the halo are served band by band by the whole workgroup (see edt_axis_kernel).
//     The window is ~sqrt(best)/w voxels, i.e. the local object radius.  The minimum is exact over the float
//     expressions, no envelope intersections, no sequential dependency between voxels.
//   * block -> tile mapping is XCD aware: the 8 XCDs (block b runs on XCD b % 8) each get a
//     contiguous 1/8 of the volume so the rows a block re-reads live in its own L2.
// Algorithmic bytes = L + 4 (x pass) and L + 8 (y, z pass) per voxel; measured HBM traffic matches them.
#include "common.h"
#include <cstdlib>

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
// yflags: the sign bit of an output says "the label changes between this voxel and the one above it in y" (never set in the last row
// of a plane): the y pass then needs no labels at all (edt_axis_kernel<.., SIGN = true>).  A wave walks consecutive rows, so the row
// above is the row it reads next anyway.
template <typename LT, int NW>
__global__ __launch_bounds__(256) void edt_x_rows_kernel(const LT* __restrict__ lab, float* __restrict__ out,
                                                         int sx, int64_t nrows, float w, int black_border, int sy, int yflags) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nblk = gridDim.x;
  const int64_t per_xcd = (nblk + 7) / 8;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);   // bits at or below this lane
  // A wave walks CONSECUTIVE rows: the labels of the next row are requested before this row is worked on (a wave has one row in
  // flight otherwise: 1 KiB), and the next row is also the row above in y -- the y flags cost one extra row per wave, not per row.
  const int64_t nwaves = nblk * 4, rpw = (nrows + nwaves - 1) / nwaves;
  const int64_t r0 = (logical * 4 + wave) * rpw, r1 = r0 + rpw < nrows ? r0 + rpw : nrows;
  uint32_t Ln[NW];
#pragma unroll
  for (int c = 0; c < NW; c++) {
    const int x = (c << 6) + lane;
    Ln[c] = (r0 < nrows && x < sx) ? (uint32_t)lab[r0 * sx + x] : 0u;
  }
  int yrow = (r0 < nrows) ? (int)(r0 % sy) : 0;      // y of the row being worked on (kept by counting: no division per row)
  for (int64_t row = r0; row < r1; row++) {
    float* __restrict__ o = out + row * sx;
    uint32_t L[NW];
    const int64_t rown = row + 1;
    const bool above = yflags && yrow != sy - 1;      // wave uniform: there is a row above this one in its plane
    yrow = (yrow + 1 == sy) ? 0 : yrow + 1;
    const bool need = rown < nrows && (rown < r1 || above);      // the next row: this wave's next one, or only the row above
#pragma unroll
    for (int c = 0; c < NW; c++) {
      const int x = (c << 6) + lane;
      L[c] = Ln[c];
      Ln[c] = (need && x < sx) ? (uint32_t)lab[rown * sx + x] : 0u;
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
      if (above && L[c] != Ln[c]) v = __uint_as_float(__float_as_uint(v) | 0x80000000u);
      o[x] = v;
    }
  }
}

// y / z pass: best = min over the same-label segment of f[j] + (w*(i-j))^2, clamped by the segment
// ends (label change, or the volume border when black_border).  The search walks outward and stops as
// soon as (w*k)^2 >= best -- nothing farther can improve the minimum -- so the result is the exact
// minimum over the float expressions in any visiting order (== oracle ko_edt_axis).
//
// Round 6: the segment ends are folded into the DATA, so the search has no limits, no masks and no labels.
// A label change between rows j and j+1 contributes the candidate (w*(i-j))^2 to a voxel i above it -- exactly what row j
// would contribute if its f were 0 -- and the voxels of the lower segment must still see row j's real f.  So a tile is staged
// as TWO views: Fd[j] = "row j as seen by somebody walking DOWN onto it" (0 when label[j] != label[j+1], else f[j]) and
// Fu[j] = "as seen walking UP" (0 when label[j] != label[j-1]).  Rows outside the volume are 0 with black_border, +inf
// without.  A walk that reaches a segment end at step kb now holds best <= (w*kb)^2 and stops by the ordinary bound test;
// whatever a group of probes reads beyond the end is >= (w*k')^2 > (w*kb)^2 >= best and cannot lower it.  Background voxels
// carry f = 0 from the x pass, so their search never starts: no background mask either.  The candidate set that can lower
// `best` is the oracle's, the minimum is over the same float expressions: bit identical.
//
// A workgroup (64 x 4 threads) owns 64 lanes along x times T = 64 positions along the axis; wave ly owns the 16 consecutive
// output rows [16 ly, 16 ly + 16), stages them (their f stays in registers: the own value is never read back) plus a quarter
// of the halo rows; the rows of the NEXT tile are requested before the search of the current one.
//
// BANDS.  The staged rows serve steps k <= H.  A tile in which some voxel's window is wider (fat objects; measured on the 512^3
// bench volume: per-lane walks on global memory for those voxels were 0.96 of the y pass's 1.37 ms) is searched again for
// k in (bH, (b+1)H], b = 1, 2, ...: the same code on the down-walk view of the tile H*b rows lower and the up-walk view of the
// tile H*b rows higher -- the previous band's views moved by H rows inside the LDS plus H new rows each, loaded by the whole
// workgroup with coalesced row loads -- the running minima kept in registers.  No thread ever walks global memory on its own.
#define KH_EDT_T 64

// min of two floats that are known to be >= +0 (or +inf): the order of the bit patterns is the order of the values
__device__ __forceinline__ float minpos(float a, float b) {
  return __uint_as_float(min(__float_as_uint(a), __float_as_uint(b)));
}

template <typename T>
__device__ __forceinline__ T ld_row(const T* __restrict__ row, uint32_t byte_off) {   // scalar row base + the lane's 32-bit byte offset
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + byte_off);
}

// CNT consecutive rows starting at position p0 of one column: f of rows p0 .. p0+CNT-1 and the labels of rows p0-1 .. p0+CNT.
// Positions are wave uniform, so a row's address is a scalar base plus the lane's x (saddr form), and the base moves by one
// scalar add per row.  CHECK: the rows may touch the ends of the axis.  A row outside the volume has f = `outside` and the label
// of the nearest row inside (clamped position): it never differs from its neighbours outside, so both its views read `outside`,
// and the flags of the first / last row inside that involve it belong to views nobody walks onto.
template <typename LT, int CNT, bool CHECK>
__device__ __forceinline__ void edt_load_rows(const LT* __restrict__ lab, const float* __restrict__ fin, int64_t rowbase,
                                              int64_t astride, int n, uint32_t xc, int p0, float outside, float* __restrict__ f,
                                              LT* __restrict__ L) {
  const uint32_t xl = xc * (uint32_t)sizeof(LT), xf = xc * 4u;
  if (!CHECK) {
    const LT* __restrict__ lp = lab + rowbase + (int64_t)(p0 - 1) * astride;
    const float* __restrict__ fp = fin + rowbase + (int64_t)p0 * astride;
#pragma unroll
    for (int j = 0; j < CNT + 2; j++) { L[j] = ld_row(lp, xl); lp += astride; }
#pragma unroll
    for (int j = 0; j < CNT; j++) { f[j] = ld_row(fp, xf); fp += astride; }
  } else {
    // the same walk with the label row clamped into the volume: the pointer moves only while the next row is inside
    const LT* __restrict__ lp = lab + rowbase + (int64_t)min(max(p0 - 1, 0), n - 1) * astride;
    const float* __restrict__ fp = fin + rowbase + (int64_t)p0 * astride;   // dereferenced for rows inside only
#pragma unroll
    for (int j = 0; j < CNT + 2; j++) {
      L[j] = ld_row(lp, xl);
      const int pn = p0 + j;
      if (pn > 0 && pn < n) lp += astride;
    }
#pragma unroll
    for (int j = 0; j < CNT; j++) {
      const int p = p0 + j;
      f[j] = outside;
      if (p >= 0 && p < n) f[j] = ld_row(fp, xf);
      fp += astride;
    }
  }
}

// The same for a pass whose input carries the label changes in its sign bits (SIGN): rows p0-1 .. p0+CNT-1 of f, f[j] = row
// p0-1+j; no labels.  Rows outside the volume read `outside` (positive: no change there).
template <int CNT, bool CHECK>
__device__ __forceinline__ void edt_load_frows(const float* __restrict__ fin, int64_t rowbase, int64_t astride, int n, uint32_t xc,
                                               int p0, float outside, float* __restrict__ f) {
  const uint32_t xf = xc * 4u;
  const float* __restrict__ fp = fin + rowbase + (int64_t)(p0 - 1) * astride;   // dereferenced for rows inside only
#pragma unroll
  for (int j = 0; j < CNT + 1; j++) {
    const int p = p0 - 1 + j;
    f[j] = outside;
    if (!CHECK || (p >= 0 && p < n)) f[j] = ld_row(fp, xf);
    fp += astride;
  }
}
// the two views of row j of such a chunk (j = 0 .. CNT-1 <-> f[j+1]; the row under it is f[j])
__device__ __forceinline__ float sgn_abs(float v) { return __uint_as_float(__float_as_uint(v) & 0x7fffffffu); }
__device__ __forceinline__ bool sgn_set(float v) { return (int)__float_as_uint(v) < 0; }

// The search of one wave's 16 output rows over the steps of one band: FOUR consecutive output rows at a time, steps in groups
// of 4.  Output al0+q at step k+s probes Fd row (al0+H-k) + (q-s) and Fu row (al0+k) + (q+s): the 32 probes of a block-group
// are 7 + 7 distinct rows, fetched with constant offsets from two base addresses before any is consumed -- four independent
// minima in flight per wave instead of one chain of LDS round trips.  The loop is wave uniform and no lane is masked: a lane (or
// row) whose bound is already crossed probes along (what it reads is >= (w*k)^2 > best and changes nothing).  (w*k)^2 comes
// from the LDS table `tq` ([k+3] = the band's step k; broadcast reads).  All values are >= +0: min on the bit patterns.
// Returns whether some lane's bound is still open after the band's last step.
template <bool LAST, int H>
__device__ __forceinline__ bool edt_search_rows(const float* __restrict__ rowd, const float* __restrict__ rowu,
                                                const float* __restrict__ tq, float (&own)[KH_EDT_T / 4], int nrows,
                                                float* __restrict__ orow, int64_t astride, bool xin) {
  bool open = false;
#pragma unroll
  for (int jb = 0; jb < KH_EDT_T / 4; jb += 4) {
    if (jb >= nrows) break;      // wave uniform: rows beyond the end of the axis
    float b0 = own[jb], b1 = own[jb + 1], b2 = own[jb + 2], b3 = own[jb + 3];
    int k = 1;           // first step not probed yet
    float tk = tq[4];    // (w*k)^2 of that step
    for (; k + 3 <= H; k += 4) {
      if (!__builtin_amdgcn_ballot_w64(tk < fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)))) break;
      const float4 t4 = *reinterpret_cast<const float4*>(&tq[k + 4]);  // steps k+1, k+2, k+3 and k+4
      const float* __restrict__ pd = rowd + (jb - k - 3) * 64;   // D[d] = pd[(d + 3) * 64], d = q - s = -3 .. 3
      const float* __restrict__ pu = rowu + (jb + k) * 64;       // U[u] = pu[u * 64],       u = q + s =  0 .. 6
      const float d0 = pd[0], d1 = pd[64], d2 = pd[128], d3 = pd[192], d4 = pd[256], d5 = pd[320], d6 = pd[384];
      const float u0 = pu[0], u1 = pu[64], u2 = pu[128], u3 = pu[192], u4 = pu[256], u5 = pu[320], u6 = pu[384];
      // min(l + t, r + t) == min(l, r) + t bit for bit (rounding is monotone): one add per output and step
      const float t0 = tk, t1 = t4.x, t2 = t4.y, t3 = t4.z;
      b0 = minpos(b0, minpos(minpos(minpos(d3, u0) + t0, minpos(d2, u1) + t1), minpos(minpos(d1, u2) + t2, minpos(d0, u3) + t3)));
      b1 = minpos(b1, minpos(minpos(minpos(d4, u1) + t0, minpos(d3, u2) + t1), minpos(minpos(d2, u3) + t2, minpos(d1, u4) + t3)));
      b2 = minpos(b2, minpos(minpos(minpos(d5, u2) + t0, minpos(d4, u3) + t1), minpos(minpos(d3, u4) + t2, minpos(d2, u5) + t3)));
      b3 = minpos(b3, minpos(minpos(minpos(d6, u3) + t0, minpos(d5, u4) + t1), minpos(minpos(d4, u5) + t2, minpos(d3, u6) + t3)));
      tk = t4.w;
    }
    own[jb] = b0; own[jb + 1] = b1; own[jb + 2] = b2; own[jb + 3] = b3;
    // the band is exhausted (k = H + 1, tk = its (w*k)^2) and somebody could still be improved from farther away
    if (k + 3 > H) open = open || (tk < fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)));
    if (xin && jb != KH_EDT_T / 4 - 4) {   // (the wave's last block is stored by the caller, one tile later: see edt_axis_kernel)
      float* __restrict__ op = orow + (int64_t)jb * astride;
      const float r0 = LAST ? sqrtf(b0) : b0, r1 = LAST ? sqrtf(b1) : b1, r2 = LAST ? sqrtf(b2) : b2, r3 = LAST ? sqrtf(b3) : b3;
      if (jb + 4 <= nrows) {       // (wave uniform; all but the last rows of the axis)
        op[0] = r0; op[astride] = r1; op[2 * astride] = r2; op[3 * astride] = r3;
      } else {
        op[0] = r0;
        if (jb + 1 < nrows) op[astride] = r1;
        if (jb + 2 < nrows) op[2 * astride] = r2;
      }
    }
  }
  return __builtin_amdgcn_ballot_w64(open) != 0;
}

template <typename LT, bool LAST, int KH_EDT_H, bool SIGN>
__global__ __launch_bounds__(256, 3) void edt_axis_kernel(const LT* __restrict__ lab, const float* __restrict__ fin,
                                                       float* __restrict__ fout, int sx, int n, int64_t astride,
                                                       int m, int64_t ostride, float w, int black_border, int chunk) {
  constexpr int T = KH_EDT_T, H = KH_EDT_H, R = T + H, OWN = T / 4, HC = H / 2, HQ = H / 4;
  static_assert(H % 4 == 0 && T == 64, "tile shape");
  __shared__ float Fd[R * 64];   // row r <-> position A0 - H + r  (A0-H .. A0+T-1): the down-walk view
  __shared__ float Fu[R * 64];   // row r <-> position A0 + r      (A0 .. A0+T+H-1): the up-walk view
  __shared__ __attribute__((aligned(16))) float tsq[H + 8];   // [k+3] = (w*k)^2 for k >= 0
  __shared__ __attribute__((aligned(16))) float tsqb[H + 8];  // the same for the band being searched: [k+3] = (w*(bH+k))^2
  __shared__ int open_epoch;   // number of the last search that ended with an open bound
  // volume seen as [sx][n along axis][m others]: index = x + a*astride + o*ostride ; blockDim = (64, 4)
  const int xt = (sx + 63) >> 6, at = (n + T - 1) / T;
  const int ntiles = xt * at * m;
  // a block owns `chunk` consecutive tiles, consecutive ALONG THE AXIS first (then x, then the other axis): a tile's lower halo and
  // the rows under it were staged by the same block one tile earlier and come out of the L2; XCD c (= blockIdx.x % 8) owns a
  // contiguous eighth of the blocks
  const int per_xcd = (int)(gridDim.x >> 3);
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  const int t0 = logical * chunk, t1 = min(t0 + chunk, ntiles);
  // blockDim = (64, 4): a wave is one row of the block, so ly is wave uniform -- say so (SGPR), which turns
  // every row-index test below into scalar code
  const uint32_t lx = threadIdx.x;
  const int ly = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int k = tid; k < H + 8; k += 256) {
    const float d = w * (float)(k - 3);
    tsq[k] = d * d;
  }
  if (tid == 0) open_epoch = 0;
  if (t0 >= t1) return;
  const float outside = black_border ? 0.0f : KH_INF;
  // The block visits its tiles in an order rotated by its own number, so that blocks which start together are not all at the
  // same x offset at the same time.
  const int cnt = t1 - t0, rot = logical % cnt;
  int tcur = t0 + rot;
  int ta = tcur % at, tx = (tcur / at) % xt, o = tcur / (xt * at);
  // the rows this thread stages for a tile: its wave's 16 output rows and a quarter of the halo (waves 0, 1 below the tile, 2, 3
  // above it).  They are requested one tile ahead, before the search of the current tile, and consumed after it.
  // [0] = the row under the chunk (SIGN: its sign bit is the chunk's first "label changes below me"), [1 ..] = the chunk's rows
  float nown[OWN + 1], nhal[HC + 1];
  LT nLo[SIGN ? 1 : OWN + 2], nLh[SIGN ? 1 : HC + 2];
  const int hrow = (ly * HC < H) ? ly * HC - H : T + ly * HC - H;   // first halo row of this wave relative to A0
  auto request = [&](int qx, int qa, int qo) {
    const uint32_t xc = min((uint32_t)((qx << 6) + lx), (uint32_t)(sx - 1));
    const int A0 = qa * T;
    const int64_t rowbase = (int64_t)qo * ostride;
    const bool inner = A0 - H - 1 >= 0 && A0 + T + H + 1 <= n;
    if constexpr (SIGN) {
      if (inner) {
        edt_load_frows<OWN, false>(fin, rowbase, astride, n, xc, A0 + OWN * ly, outside, nown);
        edt_load_frows<HC, false>(fin, rowbase, astride, n, xc, A0 + hrow, outside, nhal);
      } else {
        edt_load_frows<OWN, true>(fin, rowbase, astride, n, xc, A0 + OWN * ly, outside, nown);
        edt_load_frows<HC, true>(fin, rowbase, astride, n, xc, A0 + hrow, outside, nhal);
      }
    } else {
      if (inner) {
        edt_load_rows<LT, OWN, false>(lab, fin, rowbase, astride, n, xc, A0 + OWN * ly, outside, nown + 1, nLo);
        edt_load_rows<LT, HC, false>(lab, fin, rowbase, astride, n, xc, A0 + hrow, outside, nhal + 1, nLh);
      } else {
        edt_load_rows<LT, OWN, true>(lab, fin, rowbase, astride, n, xc, A0 + OWN * ly, outside, nown + 1, nLo);
        edt_load_rows<LT, HC, true>(lab, fin, rowbase, astride, n, xc, A0 + hrow, outside, nhal + 1, nLh);
      }
    }
  };
  // the two views of row j of a chunk held as (f[0 .. CNT], L[0 .. CNT+1]): a row reads 0 for a walker that crosses a label change onto it
  auto view_d = [&](const float* f, const LT* L, int j) -> float {
    if constexpr (SIGN) return sgn_set(f[j + 1]) ? 0.0f : sgn_abs(f[j + 1]);
    else return (L[j + 1] != L[j + 2]) ? 0.0f : f[j + 1];
  };
  auto view_u = [&](const float* f, const LT* L, int j) -> float {
    if constexpr (SIGN) return sgn_set(f[j]) ? 0.0f : sgn_abs(f[j + 1]);
    else return (L[j + 1] != L[j]) ? 0.0f : f[j + 1];
  };
  request(tx, ta, o);
  // The last four output rows of a wave are stored one tile LATER, after the next tile's views are written: the wait for the
  // requested rows at the top of a tile is a wait for every memory operation of the wave (one counter for loads and stores on
  // gfx9), and stores issued just before it would be waited for as well.
  float dfr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float* dptr = nullptr;
  int dn = 0;            // rows of the deferred block inside the axis (0: nothing deferred)
  bool dx = false;
  auto flush = [&]() {
    if (dn > 0 && dx) {
      dptr[0] = LAST ? sqrtf(dfr[0]) : dfr[0];
      if (dn > 1) dptr[astride] = LAST ? sqrtf(dfr[1]) : dfr[1];
      if (dn > 2) dptr[2 * astride] = LAST ? sqrtf(dfr[2]) : dfr[2];
      if (dn > 3) dptr[3 * astride] = LAST ? sqrtf(dfr[3]) : dfr[3];
    }
    dn = 0;
  };
  const float* __restrict__ rowd = &Fd[(OWN * ly + H) * 64 + lx];
  const float* __restrict__ rowu = &Fu[(OWN * ly) * 64 + lx];
  int epoch = 0;
  __syncthreads();   // the table and open_epoch
  for (int it = 0; it < cnt; it++) {
    const int x = (tx << 6) + (int)lx;
    const uint32_t xc = (uint32_t)min(x, sx - 1);
    const int A0 = ta * T;
    const int64_t rowbase = (int64_t)o * ostride;
    const int nrows = min(OWN, n - (A0 + OWN * ly));   // this wave's output rows inside the axis (<= 0: none)
    float own[OWN];
    // the two views: a row reads 0 for a walker that crosses a label change onto it.  (The previous tile's last search ended
    // with a barrier.)
    if (hrow < 0) {
#pragma unroll
      for (int j = 0; j < HC; j++) Fd[(H + hrow + j) * 64 + lx] = view_d(nhal, nLh, j);
    } else {
#pragma unroll
      for (int j = 0; j < HC; j++) Fu[(hrow + j) * 64 + lx] = view_u(nhal, nLh, j);
    }
#pragma unroll
    for (int j = 0; j < OWN; j++) {
      own[j] = (j < nrows) ? (SIGN ? sgn_abs(nown[j + 1]) : nown[j + 1]) : 0.0f;   // rows beyond the axis: nothing to search, never "open"
      Fd[(H + OWN * ly + j) * 64 + lx] = view_d(nown, nLo, j);
      Fu[(OWN * ly + j) * 64 + lx] = view_u(nown, nLo, j);
    }
    __syncthreads();
    flush();       // the previous tile's last block
    {
      int qa = ta + 1, qx = tx, qo = o;      // the next tile along the axis ...
      if (qa == at) { qa = 0; qx++; if (qx == xt) { qx = 0; qo++; } }
      tcur++;
      if (tcur == t1) { tcur = t0; qa = t0 % at; qx = (t0 / at) % xt; qo = t0 / (xt * at); }   // ... or the block's first one (rotation)
      if (it + 1 < cnt) request(qx, qa, qo);
      tx = qx; ta = qa; o = qo;
    }
    float* __restrict__ orow = fout + rowbase + (int64_t)(A0 + OWN * ly) * astride + xc;
    epoch++;
    if (edt_search_rows<LAST, H>(rowd, rowu, tsq, own, nrows, orow, astride, x < sx) && lx == 0) open_epoch = epoch;
    for (int b = 1;; b++) {
      __syncthreads();   // every wave's search is done (the views may be rewritten) and its open mark is visible
      if (open_epoch != epoch) break;
      const int AD = A0 - b * H, AU = A0 + b * H;     // origins of the band's two views
      if (AD + T - 1 < 0 && AU >= n && !black_border) break;   // nothing but +inf out there (with black_border the zeros stop everybody)
      // The band's down-walk view is the previous band's moved H rows up plus H new rows at its lower end, the up-walk view the
      // previous one moved H rows down plus H new rows at its upper end (a view's value belongs to its row alone): T rows of
      // either view move inside the LDS (read, barrier, write: the ranges overlap), and a wave loads H / 4 new rows per view
      // in one round trip instead of a quarter of the whole view in two.
      {
        float mv[OWN], f[HQ + 1];
        LT L[SIGN ? 1 : HQ + 2];
#pragma unroll
        for (int j = 0; j < OWN; j++) mv[j] = Fd[(OWN * ly + j) * 64 + lx];
        if constexpr (SIGN) edt_load_frows<HQ, true>(fin, rowbase, astride, n, xc, AD - H + HQ * ly, outside, f);
        else edt_load_rows<LT, HQ, true>(lab, fin, rowbase, astride, n, xc, AD - H + HQ * ly, outside, f + 1, L);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OWN; j++) Fd[(H + OWN * ly + j) * 64 + lx] = mv[j];
#pragma unroll
        for (int j = 0; j < HQ; j++) Fd[(HQ * ly + j) * 64 + lx] = view_d(f, L, j);
      }
      {
        float mv[OWN], f[HQ + 1];
        LT L[SIGN ? 1 : HQ + 2];
#pragma unroll
        for (int j = 0; j < OWN; j++) mv[j] = Fu[(H + OWN * ly + j) * 64 + lx];
        if constexpr (SIGN) edt_load_frows<HQ, true>(fin, rowbase, astride, n, xc, AU + T + HQ * ly, outside, f);
        else edt_load_rows<LT, HQ, true>(lab, fin, rowbase, astride, n, xc, AU + T + HQ * ly, outside, f + 1, L);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OWN; j++) Fu[(OWN * ly + j) * 64 + lx] = mv[j];
#pragma unroll
        for (int j = 0; j < HQ; j++) Fu[(T + HQ * ly + j) * 64 + lx] = view_u(f, L, j);
      }
      for (int k = tid; k < H + 8; k += 256) {
        const float d = w * (float)(b * H + k - 3);
        tsqb[k] = d * d;
      }
      __syncthreads();
      epoch++;
      if (edt_search_rows<LAST, H>(rowd, rowu, tsqb, own, nrows, orow, astride, x < sx) && lx == 0) open_epoch = epoch;
    }
    // the wave's last block: its final minima (bands included) wait for the next tile
#pragma unroll
    for (int q = 0; q < 4; q++) dfr[q] = own[OWN - 4 + q];
    dptr = orow + (int64_t)(OWN - 4) * astride;
    dn = min(max(nrows - (OWN - 4), 0), 4);
    dx = x < sx;
  }
  flush();
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
  // the x pass marks the label changes along y in the sign bits of its output when a y pass follows (rows of up to 1024 voxels:
  // the register kernels); that pass then reads no labels
  const int nwords = (int)((sx + 63) >> 6);
  const bool signs = do_y && nwords <= 16;
  float* bufs[2] = {out, ws};
  int cur = (npass % 2 == 1) ? 0 : 1;  // buffer the x pass writes
  {
    const int64_t need = (nrows + 3) / 4;  // 4 rows (one per wave) per workgroup and step
    int64_t grid = need < 8192 ? need : 8192;
    grid = (grid + 7) & ~7ll;  // the XCD remap needs a multiple of 8 blocks
    if (ev) KH_HIP_CHECK(hipEventRecord(ev[0], st));
    if (nwords <= 8)
      hipLaunchKernelGGL((edt_x_rows_kernel<LT, 8>), dim3((unsigned)grid), dim3(256), 0, st, lab, bufs[cur], (int)sx, nrows, wx, black_border, (int)sy, (int)signs);
    else if (nwords <= 16)
      hipLaunchKernelGGL((edt_x_rows_kernel<LT, 16>), dim3((unsigned)grid), dim3(256), 0, st, lab, bufs[cur], (int)sx, nrows, wx, black_border, (int)sy, (int)signs);
    else
      hipLaunchKernelGGL((edt_x_kernel<LT>), dim3((unsigned)grid), dim3(256), 4 * nwords * 8, st, lab, bufs[cur],
                         (int)sx, nrows, wx, black_border);
    KH_LAUNCH_CHECK();
    if (ev) KH_HIP_CHECK(hipEventRecord(ev[1], st));
  }
  auto axis = [&](int n, int64_t astride, int m, int64_t ostride, float w, bool last, bool sign) -> int {
    const int64_t ntiles = ((sx + 63) / 64) * (int64_t)((n + KH_EDT_T - 1) / KH_EDT_T) * m;
    // a block walks `chunk` consecutive tiles, requesting a tile's rows while it searches the one before.  Four tiles per block:
    // thousands of blocks for the hardware to balance (a persistent grid would depend on the occupancy it assumes), and a block
    // whose tiles need extra bands (fat objects cluster) holds the tail of the launch up less: y pass 0.67 / 0.62 / 0.58 / 0.57 ms
    // with 16 / 8 / 4 / 2 tiles per block on the 512^3 bench volume.  KH_EDT_CHUNK: developer knob (A/B runs).
    int chunk = ntiles >= 4 * 2048 ? 4 : (ntiles >= 2048 ? (int)(ntiles / 2048) : 1);
    if (const char* e = getenv("KH_EDT_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;
    int64_t grid = (ntiles + chunk - 1) / chunk;
    grid = (grid + 7) & ~7ll;  // the XCD remap needs a multiple of 8 blocks
    const float* fin = bufs[cur];
    float* fout = bufs[cur ^ 1];
    // halo: a search needs k <= sqrt(best)/w rows either side; what does not fit is served by a band (edt_axis_kernel).  32 rows on
    // an axis with the fine voxel pitch, 8 on a coarse one (w >= 2 x the finest).  Measured on the 512^3 bench volume (anisotropy
    // 16, 16, 40) on the final kernels, ms: y pass with H = 8 / 12 / 24 / 28 / 32 / 40: 0.54 / 0.50 / 0.455 / 0.45 / 0.44 / 0.49;
    // z pass with H = 8 / 12 / 16 / 32: 0.327 / 0.331 / 0.338 / 0.36 -- a small halo sends more tiles into a second band, a large
    // one costs staging and occupancy; since the bands move their views inside the LDS the curve is flat around the optimum.
    // KH_EDT_H = 8 | 16 | 24 | 32 | 40: developer knob (A/B runs).
    const float wmin = fminf(wx, fminf(wy, wz));
    int hsel = (w >= 2.0f * wmin) ? 8 : 32;
    if (const char* e = getenv("KH_EDT_H")) hsel = atoi(e);
#define KH_AXIS_LAUNCH(LASTV, HV, SV) hipLaunchKernelGGL((edt_axis_kernel<LT, LASTV, HV, SV>), dim3((unsigned)grid), dim3(64, 4), 0, st, \
                                                         lab, fin, fout, (int)sx, n, astride, m, ostride, w, black_border, chunk)
#define KH_AXIS_H(LASTV, SV) do { if (hsel <= 8) KH_AXIS_LAUNCH(LASTV, 8, SV); else if (hsel <= 16) KH_AXIS_LAUNCH(LASTV, 16, SV); \
                                  else if (hsel <= 24) KH_AXIS_LAUNCH(LASTV, 24, SV); else if (hsel <= 32) KH_AXIS_LAUNCH(LASTV, 32, SV); else KH_AXIS_LAUNCH(LASTV, 40, SV); } while (0)
#define KH_AXIS_PICK(LASTV) do { if (sign) KH_AXIS_H(LASTV, true); else KH_AXIS_H(LASTV, false); } while (0)
    if (last) KH_AXIS_PICK(true); else KH_AXIS_PICK(false);
#undef KH_AXIS_H
#undef KH_AXIS_PICK
#undef KH_AXIS_LAUNCH
    KH_LAUNCH_CHECK();
    cur ^= 1;
    return KH_OK;
  };
  if (do_y) { int rc = axis((int)sy, sx, (int)sz, sx * sy, wy, !do_z, signs); if (rc) return rc; }
  if (ev) KH_HIP_CHECK(hipEventRecord(ev[2], st));
  if (do_z) { int rc = axis((int)sz, sx * sy, (int)sy, sx, wz, true, false); if (rc) return rc; }
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
