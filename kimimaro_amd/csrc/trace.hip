// trace.hip -- the per-label TEASAR searches for gfx950: one workgroup per label (512 threads for the
// distance fields, one 64-lane wave for the path loop).
//
//   kh_edf_batch    a4  dijkstra3d.euclidean_distance_field   (kimimaro/trace.py:139-145, 302-307)
//   kh_trace_paths  a6-a11 compute_paths loop                 (kimimaro/trace.py:196-267):
//                     target finder   skeletontricks.pyx:995-1045
//                     railroad        dijkstra3d.railroad, trace.py:240-242
//                     invalidation    skeletontricks.pyx:373-418 -> dijkstra_invalidation.hpp:239-332
//                     rail edits      trace.py:220, 261-263
//
// Search = label-correcting relaxation with a near/far split (delta-stepping with an adaptive
// threshold): work items are (frontier voxel, direction) pairs spread over the lanes, distances are
// float bit patterns updated with atomicMin, the work lists hold voxel indices only (membership bits in
// `qstate` keep every list bounded by the label size).  The distances converge to the unique Bellman
// fixpoint d[v] = min_u fl(d[u] + w), so they equal the oracle's heap Dijkstra bit for bit regardless of
// the relaxation order; paths are then recovered with the canonical predecessor rule of
// oracle/kimi_oracle.c (ko_pred / ko_walk), 26 lanes looking at the 26 neighbours at once.
//
// The invalidation flood is order dependent down to the tie order of
// std::priority_queue, so it is run as an exact emulation of the libstdc++ binary heap in which the 64
// lanes of the label's wave cooperate on every push and pop (see "The invalidation heap" below), with
// the 26 neighbour tests of each popped voxel evaluated by 26 lanes.
//
// No MFMA: irregular, latency/atomic bound integer+f32 work (north_star).  Labels are independent, so
// the chip is filled by running every label's workgroup concurrently; the largest labels keep two
// chunks of their heap in LDS (kh_trace_paths n_large).
#include <stdlib.h>

#include "common.h"
#include "sweep.h"

namespace kh {

static constexpr uint32_t INF_BITS = 0x7f800000u;
static constexpr unsigned long long NONE64 = ~0ull;

__device__ __forceinline__ float ld_f32_l2(const float* p) {
  // dist words are modified by L2 atomics; read them at agent scope (bypasses the per-CU L1)
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_f32_l2(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pack(float d, uint32_t v) {
  return ((unsigned long long)__float_as_uint(d) << 32) | v;
}
__device__ __forceinline__ float next_up(float x) { return __uint_as_float(__float_as_uint(x) + 1u); }
// read lane `l` (wave-uniform index) of a 32-bit value: v_readlane instead of a ds_bpermute round trip
__device__ __forceinline__ uint32_t rdlane_u32(uint32_t v, int l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l));
}
struct Ctl {
  unsigned long long best_rail;
  unsigned long long red64[16];
  uint32_t n_cur, n_next, n_far, n_far2, n_touched;
  uint32_t status;
  float red_min[16];
  float red_sum[16];
  uint32_t red_cnt[16];
  float T;
  uint32_t u0, u1, u2, u3;
  uint32_t retry[2];     // sssp: a frontier voxel of this batch could not place an offer (parity of the batch)
  uint32_t pool_base;
  unsigned long long cyc3[3];
  Geometry g;  // the 26 offsets / edge lengths are indexed per lane: keep them in LDS, not in SGPRs
};

// work lists hold voxel indices only; membership is tracked by two bits per voxel in `qstate`
// (bit0: queued in the near list being built, bit1: queued in the far list), so a voxel is in each
// list at most once and every list is bounded by the label size Nf.
struct Queues {
  uint32_t* a;
  uint32_t* b;
  uint32_t* c;
  uint32_t* touched;
  uint32_t cap;
};

__device__ __forceinline__ uint32_t flag_or(uint8_t* base, uint32_t v, uint32_t bit) {
  uint32_t* w = reinterpret_cast<uint32_t*>(base + (v & ~3u));
  const int sh = (int)(v & 3u) * 8;
  return (atomicOr(w, bit << sh) >> sh) & 0xFFu;
}
__device__ __forceinline__ void flag_clear(uint8_t* base, uint32_t v, uint32_t bit) {
  uint32_t* w = reinterpret_cast<uint32_t*>(base + (v & ~3u));
  const int sh = (int)(v & 3u) * 8;
  atomicAnd(w, ~(bit << sh));
}

// Address spaces by type inside the searches: the pointers arrive through a function boundary as generic ones, and a
// generic (flat) access counts on the LDS counter as well as on the vector-memory one -- an LDS read between two of them
// would wait for the first to return.  With the address space in the type the loads are global_load / ds_read and the
// stages below really overlap.
#define KH_AS_GLOBAL __attribute__((address_space(1)))
#define KH_AS_LDS __attribute__((address_space(3)))
typedef KH_AS_GLOBAL uint32_t gu32_t;
typedef KH_AS_GLOBAL float gf32_t;
__device__ __forceinline__ uint32_t gld_l2(const gu32_t* p) {       // agent-scope load: the word is modified by L2 atomics
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t gflag_or(gu32_t* qs, uint32_t v, uint32_t bit) {
  const int sh = (int)(v & 3u) * 8;
  return (__hip_atomic_fetch_or(qs + (v >> 2), bit << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> sh) & 0xFFu;
}
__device__ __forceinline__ void gflag_clear(gu32_t* qs, uint32_t v, uint32_t bit) {
  const int sh = (int)(v & 3u) * 8;
  __hip_atomic_fetch_and(qs + (v >> 2), ~(bit << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The 27 words around voxel u of a whole-volume u32 / f32 array as nine rows of three consecutive words:
// w[(dx + 1) + 3 * ((dy + 1) + 3 * (dz + 1))] = a[u + dx + sx * dy + sxy * dz].  A row outside the volume reads u's own row (its
// neighbours are in nobody's mask); a row that starts one word before the array or ends one behind it is read one word further in
// and shifted (the missing word belongs to a neighbour outside the volume).  nvox >= 3.
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef u32x3_t u32x3_a4_t __attribute__((aligned(4)));
__device__ __forceinline__ void rows27_base(uint32_t nvox, int sx, int sxy, int sy, int sz, uint32_t u, int y, int z, uint32_t (&bc)[9],
                                            int (&sh)[9]) {
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const int dy = r % 3 - 1, dz = r / 3 - 1;
    const bool in = (unsigned)(y + dy) < (unsigned)sy && (unsigned)(z + dz) < (unsigned)sz;
    const long long base = (long long)u + (in ? dy * sx + dz * sxy : 0) - 1;
    const long long hi = nvox >= 3u ? (long long)nvox - 3 : 0;       // (arrays of fewer than three words carry padding: include/kimi_hip.h)
    const long long c = base < 0 ? 0 : (base > hi ? hi : base);
    bc[r] = (uint32_t)c;
    sh[r] = (int)(base - c);
  }
}
__device__ __forceinline__ void rows27_shift(const u32x3_t (&t)[9], const int (&sh)[9], uint32_t (&w)[27]) {
#pragma unroll
  for (int r = 0; r < 9; r++) {
    w[3 * r + 0] = sh[r] > 0 ? t[r].y : t[r].x;                       // (sh < 0: the word in front of the array -- never used)
    w[3 * r + 1] = sh[r] > 0 ? t[r].z : (sh[r] < 0 ? t[r].x : t[r].y);
    w[3 * r + 2] = sh[r] < 0 ? t[r].y : t[r].z;                       // (sh > 0: the word behind the array -- never used)
  }
}
// words that memory-side atomics change: past the vector cache (sc1).  The nine loads and their wait are ONE asm statement, so the
// compiler never sees a register that is still in flight; the wait also covers the loads it has issued itself just before.
__device__ __forceinline__ void rows27_sc1(const KH_AS_GLOBAL uint32_t* a, uint32_t nvox, int sx, int sxy, int sy, int sz, uint32_t u,
                                           int y, int z, uint32_t (&w)[27]) {
  uint32_t bc[9];
  int sh[9];
  rows27_base(nvox, sx, sxy, sy, sz, u, y, z, bc, sh);
  u32x3_t t[9];
  asm volatile(
      "global_load_dwordx3 %0, %9, off sc1\n\t"
      "global_load_dwordx3 %1, %10, off sc1\n\t"
      "global_load_dwordx3 %2, %11, off sc1\n\t"
      "global_load_dwordx3 %3, %12, off sc1\n\t"
      "global_load_dwordx3 %4, %13, off sc1\n\t"
      "global_load_dwordx3 %5, %14, off sc1\n\t"
      "global_load_dwordx3 %6, %15, off sc1\n\t"
      "global_load_dwordx3 %7, %16, off sc1\n\t"
      "global_load_dwordx3 %8, %17, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8])
      : "v"(a + bc[0]), "v"(a + bc[1]), "v"(a + bc[2]), "v"(a + bc[3]), "v"(a + bc[4]), "v"(a + bc[5]), "v"(a + bc[6]), "v"(a + bc[7]),
        "v"(a + bc[8])
      : "memory");
  rows27_shift(t, sh, w);
}
// words only plain stores of this workgroup change (the weight field: rails are zeroed between the searches)
__device__ __forceinline__ void rows27_plain(const KH_AS_GLOBAL float* a, uint32_t nvox, int sx, int sxy, int sy, int sz, uint32_t u, int y,
                                             int z, float (&w)[27]) {
  uint32_t bc[9];
  int sh[9];
  rows27_base(nvox, sx, sxy, sy, sz, u, y, z, bc, sh);
  u32x3_t t[9];
#pragma unroll
  for (int r = 0; r < 9; r++) t[r] = *(const KH_AS_GLOBAL u32x3_a4_t*)((const KH_AS_GLOBAL uint32_t*)a + bc[r]);
  uint32_t wu[27];
  rows27_shift(t, sh, wu);
#pragma unroll
  for (int i = 0; i < 27; i++) w[i] = __uint_as_float(wu[i]);
}

// LDS scratch of a search (north_star: "frontier relaxation with LDS bucket queues and wave-ballot compaction"): KH_SSSP_LDS_BYTES
// per thread of the workgroup, laid out as
//   key[H], val[H]   the COMBINE TABLE of a batch of frontier voxels: open addressing on the voxel index, val = the smallest
//                    fl(d[u] + w) any frontier voxel of the batch offers the voxel (LDS atomic min).  A batch relaxes into the
//                    table, a barrier, then every occupied slot is flushed by ONE thread: it is the only writer of dist[v], of
//                    v's membership byte and of v's list entries, so the searches issue NO global atomic (every global atomic
//                    on gfx950 is a fabric round trip that drops its line from the L2: DESIGN.md r6-1).
//   qa[NQ], qb[NQ]   the heads of the two near work lists (bucket "below T" being processed / being built); entries from NQ on
//                    overflow to the label's lists in HBM at the same index.  The far bucket stays in HBM (it is split by a
//                    streaming pass).
// List slots are handed out per wave: a ballot of the lanes that append, ONE LDS atomic add by the first of them, a prefix
// population count for the others.
#define KH_SSSP_LDS_BYTES 48u   /* per thread: H = 4 slots of 8 bytes, NQ = 2 + 2 entries of 4 bytes */
static constexpr uint32_t SSSP_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uintptr_t uni64(uintptr_t v) { return ((uintptr_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); }
typedef KH_AS_GLOBAL uint8_t gu8_t;
__device__ __forceinline__ uint32_t gld8_l2(const gu8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gst8_l2(gu8_t* p, uint32_t v) { __hip_atomic_store(p, (uint8_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// slot of this lane among the lanes of its wave with p set (all lanes of the wave call this): base from one LDS atomic
__device__ __forceinline__ uint32_t wave_slot(bool p, KH_AS_LDS uint32_t* counter, int lane) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
  if (m == 0ull) return 0u;
  const int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0u;
  if (lane == leader) base = __hip_atomic_fetch_add(counter, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ bool sssp_offer(KH_AS_LDS uint32_t* key, KH_AS_LDS uint32_t* val, uint32_t hmask, int hshift, uint32_t v,
                                           uint32_t nb) {
  uint32_t s = (v * 0x9E3779B1u) >> hshift;
#pragma unroll 1
  for (int t = 0; t < 12; t++) {
    uint32_t seen = SSSP_EMPTY;
    __hip_atomic_compare_exchange_strong(key + s, &seen, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (seen == SSSP_EMPTY || seen == v) {
      __hip_atomic_fetch_min(val + s, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return true;
    }
    s = (s + 1u) & hmask;
  }
  return false;       // the neighbourhood of the slot is full: the frontier voxel is relaxed again after the flush
}

// MODE 0: EDF (edge length by direction).  MODE 1: railroad (cost = pdrf of the entered voxel, rails
// absorb, stops once everything at or below the nearest rail is final).  MODE 2: parental field
// (trace.py:155, fix_branching=False): field costs, no rails, runs to completion.
// On return distances below the final threshold are exact (ctl->best_rail set for MODE 1).
// lds / lds_bytes: the workgroup's search scratch, >= KH_SSSP_LDS_BYTES * blockDim.x bytes, 4-byte aligned.
template <int MODE>
__device__ __attribute__((noinline)) void sssp(const Geometry& g_, const uint32_t* __restrict__ nbrmask_, const float* __restrict__ wfield_,
                     float* dist_, uint8_t* qstate_, uint32_t source, Queues q, Ctl* ctl_, float delta_floor,
                     unsigned char* lds_, uint32_t lds_bytes, uint32_t preseeded_far = 0) {
  constexpr bool RAIL = MODE == 1;
  constexpr bool FIELD = MODE != 0;  // MODE 2: dijkstra3d.parental_field -- field weights, no rails, runs to completion
  const int tid = threadIdx.x;
  const int nthr = blockDim.x, nwav = nthr >> 6;
  const int lane = tid & 63, wave = tid >> 6;
  // The arguments of a function that is not a kernel arrive in VECTOR registers, uniform or not: eight 64-bit pointers and every
  // address derived from them would live in VGPR pairs.  Read back through lane 0 they are scalars again.
  const KH_AS_LDS Geometry* g = (const KH_AS_LDS Geometry*)(uintptr_t)uni32((uint32_t)(uintptr_t)(const KH_AS_LDS Geometry*)&g_);
  KH_AS_LDS Ctl* ctl = (KH_AS_LDS Ctl*)(uintptr_t)uni32((uint32_t)(uintptr_t)(KH_AS_LDS Ctl*)ctl_);
  const gu32_t* nbrmask = (const gu32_t*)uni64((uintptr_t)nbrmask_);
  const gf32_t* wfield = (const gf32_t*)uni64((uintptr_t)wfield_);
  gu32_t* dist = (gu32_t*)uni64((uintptr_t)dist_);                   // float bit patterns (non-negative: ordered like unsigned)
  gu8_t* qs8 = (gu8_t*)uni64((uintptr_t)qstate_);                    // membership bytes: bit 0 near list being built, bit 1 far list
  q.cap = uni32(q.cap);
  gu32_t* cur = (gu32_t*)uni64((uintptr_t)q.a);
  gu32_t* next = (gu32_t*)uni64((uintptr_t)q.b);
  gu32_t* far = (gu32_t*)uni64((uintptr_t)q.c);
  gu32_t* touched = (gu32_t*)uni64((uintptr_t)q.touched);
  source = uni32(source);
  lds_bytes = uni32(lds_bytes);
  delta_floor = __uint_as_float(uni32(__float_as_uint(delta_floor)));
  lds_ = (unsigned char*)(uintptr_t)uni32((uint32_t)(uintptr_t)(KH_AS_LDS unsigned char*)lds_);
  uint32_t H = 1u;
  while (H * 24u <= lds_bytes) H <<= 1;                              // the largest power of two with 12 H <= lds_bytes
  const uint32_t NQ = H >> 1, hmask = H - 1u;
  const int hshift = __clz((int)H) + 1;
  KH_AS_LDS uint32_t* key = (KH_AS_LDS uint32_t*)(uintptr_t)lds_;
  KH_AS_LDS uint32_t* val = key + H;
  KH_AS_LDS uint32_t* curL = val + H;
  KH_AS_LDS uint32_t* nextL = curL + NQ;
  float T = RAIL ? 1e-45f : delta_floor;
  for (uint32_t s = tid; s < H; s += nthr) { key[s] = SSSP_EMPTY; val[s] = 0xFFFFFFFFu; }
  if (tid == 0) {
    ctl->n_cur = 1; ctl->n_next = 0; ctl->n_far = preseeded_far; ctl->n_far2 = 0;  // far list q.c may hold seeds
    ctl->n_touched = 0;
    ctl->best_rail = NONE64;
    ctl->retry[0] = ctl->retry[1] = 0u;
    curL[0] = source;
    __hip_atomic_store(dist + source, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (RAIL) { touched[0] = source; ctl->n_touched = 1; }
  }
  __syncthreads();
  const int gsx = g->sx, gsxy = g->sxy, gsy = g->sy, gsz = g->sz;
  const uint32_t nvox = (uint32_t)gsxy * (uint32_t)gsz;
  uint32_t par = 0u;
  for (;;) {
    // ---- near phase: label-correct everything below T
    for (;;) {
      const uint32_t n = ctl->n_cur;
      if (n == 0) break;
      // ONE THREAD PER FRONTIER VOXEL, a batch of blockDim.x voxels at a time.  A thread reads its voxel's mask and the 27 distances
      // around it as nine rows of three consecutive words (past the vector cache) -- with a weight field its 27 weights the same
      // way -- in ONE round trip whose addresses depend on nothing but the voxel; every edge that lowers a distance is offered to
      // the combine table; after the barrier the table is flushed (see above).
      for (uint32_t base = 0; base < n; base += (uint32_t)nthr) {
        const uint32_t i = base + (uint32_t)tid;
        bool pending = i < n;
        const uint32_t u = pending ? (i < NQ ? curL[i] : cur[i]) : source;
        const uint32_t zz = u / (uint32_t)gsxy, rr = u - zz * (uint32_t)gsxy, yy = rr / (uint32_t)gsx;
        if (pending) gst8_l2(qs8 + u, gld8_l2(qs8 + u) & ~1u);       // out of the near list (the only thread that holds u)
        for (;;) {
          if (pending) {
            const uint32_t nm = nbrmask[u];
            float wr[27];
            if (FIELD) rows27_plain(wfield, nvox, gsx, gsxy, gsy, gsz, u, (int)yy, (int)zz, wr);
            uint32_t dr[27];
            rows27_sc1(dist, nvox, gsx, gsxy, gsy, gsz, u, (int)yy, (int)zz, dr);
            const float du = __uint_as_float(dr[13]);
            bool failed = false;
#pragma unroll
            for (int k = 0; k < 26; k++) {
              int dx, dy, dz;
              dir_delta(k, dx, dy, dz);
              const int idx = (dx + 1) + 3 * ((dy + 1) + 3 * (dz + 1));
              const float wn = FIELD ? wr[idx] : g->w[k];
              const uint32_t nb = __float_as_uint(du + wn);
              // distances only go down, so an edge that cannot lower dist[v] now never will
              if (((nm >> k) & 1u) && nb < dr[idx])
                failed |= !sssp_offer(key, val, hmask, hshift, u + (uint32_t)(dx + gsx * dy + gsxy * dz), nb);
            }
            pending = failed;
            if (failed) ctl->retry[par] = 1u;
          }
          __syncthreads();
          // ---- flush: one thread per occupied slot
          for (uint32_t s = (uint32_t)tid; s < H; s += (uint32_t)nthr) {
            const uint32_t v = key[s];
            const bool occ = v != SSSP_EMPTY;
            uint32_t nd = 0u, old = 0u, q8 = 0u;
            float wv = 1.0f;
            if (occ) {
              nd = val[s];
              key[s] = SSSP_EMPTY;
              val[s] = 0xFFFFFFFFu;
              if (RAIL) { old = gld_l2(dist + v); wv = wfield[v]; }
              q8 = gld8_l2(qs8 + v);
              __hip_atomic_store(dist + v, nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const float ndf = __uint_as_float(nd);
            const bool rail = RAIL && occ && wv == 0.0f;            // a rail: absorbing
            if (rail) __hip_atomic_fetch_min(&ctl->best_rail, pack(ndf, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool t_touch = RAIL && occ && old == INF_BITS;
            const bool t_next = occ && !rail && ndf < T && !(q8 & 1u);
            const bool t_far = occ && !rail && !(ndf < T) && !(q8 & 2u);
            if (t_next | t_far) gst8_l2(qs8 + v, q8 | (t_next ? 1u : 2u));
            if (RAIL) {
              const uint32_t p = wave_slot(t_touch, &ctl->n_touched, lane);
              if (t_touch) {
                if (p < q.cap) touched[p] = v;
                else __hip_atomic_fetch_or(&ctl->status, (uint32_t)KH_ST_QUEUE_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
            {
              const uint32_t p = wave_slot(t_next, &ctl->n_next, lane);
              if (t_next) {
                if (p < NQ) nextL[p] = v;
                else if (p < q.cap) next[p] = v;
                else __hip_atomic_fetch_or(&ctl->status, (uint32_t)KH_ST_QUEUE_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
            {
              const uint32_t p = wave_slot(t_far, &ctl->n_far, lane);
              if (t_far) {
                if (p < q.cap) far[p] = v;
                else __hip_atomic_fetch_or(&ctl->status, (uint32_t)KH_ST_QUEUE_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
          const bool again = ctl->retry[par] != 0u;                  // (stable since the barrier above)
          if (tid == 0) ctl->retry[par ^ 1u] = 0u;                   // (nobody reads or sets the other word before the next barrier)
          par ^= 1u;
          __syncthreads();
          if (!again) break;
        }
      }
      if (tid == 0) {
        ctl->n_cur = ctl->n_next < q.cap ? ctl->n_next : q.cap;
        ctl->n_next = 0;
        if (ctl->n_far > q.cap) ctl->n_far = q.cap;
      }
      { gu32_t* t = cur; cur = next; next = t; }
      { KH_AS_LDS uint32_t* t = curL; curL = nextL; nextL = t; }
      __syncthreads();
    }
    // ---- every voxel with d < T is final now
    float tcap = KH_INF;
    if (RAIL) {
      const unsigned long long br = ctl->best_rail;
      if (br != NONE64) {
        const float D = __uint_as_float((uint32_t)(br >> 32));
        if (D < T) break;
        tcap = next_up(D);
      }
    }
    const uint32_t nfar = ctl->n_far;
    if (nfar == 0) break;
    // pass 1: min / mean of the far entries (current distances)
    float mn = KH_INF, sm = 0.0f;
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < nfar; i += nthr) {
      const float d = __uint_as_float(gld_l2(dist + far[i]));
      mn = fminf(mn, d); sm += d; cnt++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, o));
      sm += __shfl_xor(sm, o);
      cnt += __shfl_xor(cnt, o);
    }
    if (lane == 0) { ctl->red_min[wave] = mn; ctl->red_sum[wave] = sm; ctl->red_cnt[wave] = cnt; }
    __syncthreads();
    mn = ctl->red_min[0]; sm = ctl->red_sum[0]; cnt = ctl->red_cnt[0];
    for (int i = 1; i < nwav; i++) { mn = fminf(mn, ctl->red_min[i]); sm += ctl->red_sum[i]; cnt += ctl->red_cnt[i]; }
    if (RAIL && !(mn < tcap)) break;  // nothing left at or below the nearest rail
    const float mean = sm / (float)cnt;
    float step = 0.5f * (mean - mn);
    if (!(step > delta_floor)) step = delta_floor;
    float Tn = mn + step;
    if (!(Tn > mn)) Tn = next_up(mn);
    if (Tn < T) Tn = T;
    if (Tn > tcap) Tn = tcap;
    T = Tn;
    // pass 2: split far -> cur (d < T) + compacted far (into the free `next` buffer); every entry has one thread, which is the
    // only writer of the voxel's membership byte
    for (uint32_t base = 0; base < nfar; base += (uint32_t)nthr) {
      const uint32_t i = base + (uint32_t)tid;
      const bool have = i < nfar;
      const uint32_t v = have ? far[i] : source;
      const float d = __uint_as_float(gld_l2(dist + v));
      const uint32_t q8 = gld8_l2(qs8 + v) & ~2u;
      const bool near = have && d < T;
      const bool t_cur = near && !(q8 & 1u);
      const bool t_far = have && !near;
      if (have) gst8_l2(qs8 + v, q8 | (t_cur ? 1u : 0u) | (t_far ? 2u : 0u));
      const uint32_t pc = wave_slot(t_cur, &ctl->n_cur, lane);
      if (t_cur) { if (pc < NQ) curL[pc] = v; else cur[pc] = v; }    // pc < nfar <= cap
      const uint32_t pf = wave_slot(t_far, &ctl->n_far2, lane);
      if (t_far) next[pf] = v;
    }
    __syncthreads();
    if (tid == 0) { ctl->n_far = ctl->n_far2; ctl->n_far2 = 0; }
    gu32_t* t = far; far = next; next = t;
    __syncthreads();
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// One distance-field search of ONE label by its workgroup (dijkstra3d.euclidean_distance_field, kimimaro/trace.py:139-145, 302-307):
// the field over the label's voxels from `source`, +inf before; task.max_loc / max_val = the farthest voxel (ties -> smallest
// linear index); mode 1 (find_root): task.root = that voxel.  Out of line: the batch kernel and the path kernel share it.
__device__ __attribute__((noinline)) void edf_label(Ctl* ctl_, kh_label_t* task, int mode, const uint32_t* __restrict__ list, uint32_t nf,
                                                    const uint32_t* __restrict__ nbrmask, float* field, uint8_t* qstate, Queues q,
                                                    float delta_floor, uint32_t source, unsigned char* lds, uint32_t lds_bytes) {
  Ctl& ctl = *ctl_;
  const Geometry& g = ctl.g;
  const int tid = threadIdx.x;
  const int nthr = blockDim.x, nwav = nthr >> 6;
  const int lane = tid & 63, wave = tid >> 6;
  for (uint32_t i = tid; i < nf; i += nthr) st_f32_l2(&field[list[i]], KH_INF);
  __syncthreads();
  uint32_t seeded = 0;
  if (mode == 2 && task->fsr > 0.0f) {
    // free_space_radius (trace.py:134,142; dijkstra3d source absent, restated in oracle ko_edf): label
    // voxels closer than the radius in a straight line get that distance and seed the search (far list).
    const float fsr = task->fsr;
    const uint32_t sxu = (uint32_t)g.sx, sxy = (uint32_t)g.sxy;
    const uint32_t z0 = source / sxy, r0 = source - z0 * sxy, y0 = r0 / sxu, x0 = r0 - y0 * sxu;
    if (tid == 0) ctl.u0 = 0;
    __syncthreads();
    for (uint32_t i = tid; i < nf; i += nthr) {
      const uint32_t v = list[i];
      if (v == source) continue;
      const uint32_t z = v / sxy, r = v - z * sxy, y = r / sxu, x = r - y * sxu;
      const float a = g.wx * (float)((int)x - (int)x0), b = g.wy * (float)((int)y - (int)y0), c = g.wz * (float)((int)z - (int)z0);
      float s2 = a * a;
      const float t2 = b * b, u2 = c * c;
      s2 = s2 + t2;
      s2 = s2 + u2;
      const float sd = sqrtf(s2);
      if (sd < fsr) {
        st_f32_l2(&field[v], sd);
        flag_or(qstate, v, 2u);
        const uint32_t p = atomicAdd(&ctl.u0, 1u);
        q.c[p] = v;  // p < nf <= cap
      }
    }
    __syncthreads();
    seeded = ctl.u0;
  }
  sssp<0>(ctl.g, nbrmask, nullptr, field, qstate, source, q, &ctl, delta_floor, lds, lds_bytes, seeded);
  // farthest voxel: max finite distance, ties -> smallest linear index
  unsigned long long best = 0;
  for (uint32_t i = tid; i < nf; i += nthr) {
    const uint32_t v = list[i];
    const uint32_t b = __float_as_uint(ld_f32_l2(&field[v]));
    if (b == INF_BITS) continue;
    const unsigned long long key = ((unsigned long long)b << 32) | (0xFFFFFFFFu - v);
    if (key > best) best = key;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long ob = __shfl_xor(best, o);
    if (ob > best) best = ob;
  }
  if (lane == 0) ctl.red64[wave] = best;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < nwav; i++) if (ctl.red64[i] > best) best = ctl.red64[i];
    const uint32_t loc = 0xFFFFFFFFu - (uint32_t)best;
    task->max_loc = loc;
    task->max_val = __uint_as_float((uint32_t)(best >> 32));
    if (mode == 1) task->root = loc;
    ctl.red64[0] = best;          // (for the callers' other threads: the record itself was written by this thread only)
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void edf_batch_kernel(kh_label_t* tasks, int mode, const uint32_t* __restrict__ lists,
                                                        const uint32_t* __restrict__ nbrmask, Geometry g, float* field,
                                                        uint8_t* qstate, uint32_t* queues, float delta_floor) {
  __shared__ Ctl ctl;
  extern __shared__ __attribute__((aligned(16))) unsigned char search_lds[];     // KH_SSSP_LDS_BYTES per thread
  kh_label_t* task = &tasks[blockIdx.x];
  const int tid = threadIdx.x;
  if (mode == 1 && task->root != 0xFFFFFFFFu) return;
  const uint32_t source = (mode == 2) ? task->root : task->source;
  if (tid == 0) { ctl.status = 0; ctl.g = g; }
  __syncthreads();
  Queues q;
  q.cap = task->q_capacity;
  q.a = queues + (uint64_t)task->q_offset * 4;
  q.b = q.a + q.cap;
  q.c = q.b + q.cap;
  q.touched = q.c + q.cap;
  edf_label(&ctl, task, mode, lists + task->list_offset, task->count, nbrmask, field, qstate, q, delta_floor, source, search_lds,
            KH_SSSP_LDS_BYTES * blockDim.x);
  if (tid == 0) task->status |= ctl.status;
}

// ------------------------------------------------------------------------------------------------
// The invalidation heap: an exact emulation of std::priority_queue<HeapDistanceNode, vector, Compare>
// with Compare = `t1.dist >= t2.dist` (dijkstra_invalidation.hpp:233-237, 262-264) as libstdc++
// implements it (bits/stl_heap.h), because the pop order among equal keys decides which source owns
// a voxel.  The array layout after every operation is identical to libstdc++'s:
//   push  = __push_heap: the new key climbs over every ancestor with key >= it.  The ancestors of the
//           new leaf are known up front, so the wave loads them all at once (lane g = generation g),
//           a ballot gives the climb length and the lanes shift the chain down in one step.
//   pop   = __pop_heap/__adjust_heap: libstdc++ walks the hole to a leaf along the smaller child
//           (ties: left) and then pushes the last element up again.  Because keys never decrease
//           from parent to child this lands exactly where the text-book early-exit sift-down lands
//           (verified against the libstdc++ form in oracle/ tests), so the wave descends 6 levels
//           per memory round trip: 126 speculative child nodes are fetched by the 64 lanes, the path
//           is resolved from registers, and the nodes on it are moved up in one parallel step.
// Nodes are 16-byte records {key bits, voxel, source index, -} in the label's slice of HBM scratch, so
// a node is one dwordx4 load or store.  Keys are non-negative floats: they are compared as their bit
// patterns (unsigned), which lets "ties go left" be written as k < sibling + (1 on left lanes).
// A write-through LDS mirror of heap levels 0-12 was tried twice and measured 10-15 % SLOWER: the pop is
// bound by instruction issue of its single wave more than by memory latency.
// a native LLVM vector (HIP's uint4 is a struct around a union, which ends up in scratch memory here)
typedef uint32_t hnode_t __attribute__((ext_vector_type(4)));
// LDS pointers keep their address space in the type, so LDS and HBM accesses can never be merged into flat_* ones
typedef __attribute__((address_space(3))) hnode_t lds_hnode_t;

// Heap slots 0..TOP-1 live in LDS and nowhere else; slots >= TOP live in the label's slice of HBM scratch.
// TOPL = 1: the root and the first 6-level chunk under it (127 slots, 2 KiB) -- every label affords that.
// TOPL = 2: two chunks (8191 slots, 128 KiB): one such workgroup fits on a CU, so it is reserved for the few
// largest labels, whose sequential chain is the critical path of the whole launch.  Every pop starts in the
// LDS part, so the top read and the first TOPL chunks cost LDS round trips instead of L2 ones.
template <int TOPL_>
struct Heap {
  static constexpr int TOPL = TOPL_;
  static constexpr uint32_t TOP = TOPL_ == 1 ? 127u : 8191u;
  hnode_t* node;   // HBM scratch of this label (L2 resident in practice); slots < TOP unused
  lds_hnode_t* top;  // LDS, TOP + 3 entries (the last LDS chunk's lanes 62/63 read two slots past the end)
  uint32_t cap, n;
  // per-lane constants of the 126-node speculative sub-tree (children of node m: 2m+2, 2m+3; parent of
  // m >= 2: (m-2)>>1).  Lane l holds node m = l ("slot 0", depths 1..6) and m = l+64 ("slot 1", depth 6).
  unsigned long long am0, am1;  // slot-0 ballot bits of the node's ancestors (am0 including itself)
  uint32_t sh0, j0, j1;         // heap index of my slot-0 node = ((hole+1) << sh0) - 1 + j0, slot 1: << 6, + j1
  uint32_t lf;                  // 1 on even lanes (left children), 0 on odd lanes
};

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// value of lane l^1 (the sibling node): DPP quad_perm [1,0,3,2], no LDS crossbar round trip
__device__ __forceinline__ uint32_t sibling_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

template <class H>
__device__ __forceinline__ void heap_init_lane(H& h, int lane) {
  unsigned long long a0 = 0, a1 = 0;
  for (int m = lane; ; m = (m - 2) >> 1) { a0 |= 1ull << m; if (m < 2) break; }
  if (lane + 64 < 126) for (int m = (lane + 62) >> 1; ; m = (m - 2) >> 1) { a1 |= 1ull << m; if (m < 2) break; }
  h.am0 = a0;
  h.am1 = a1;
  const int d0 = 31 - __clz(lane + 2);
  h.sh0 = (uint32_t)d0;
  h.j0 = (uint32_t)(lane + 2 - (1 << d0));
  h.j1 = (uint32_t)(lane + 2);
  h.lf = (lane & 1) ? 0u : 1u;
}

// all 64 lanes of the wave call this with uniform arguments.  One memory round trip: lane g loads
// the whole node of ancestor generation g+1 (from LDS or HBM, whichever holds that slot), a ballot gives
// the climb length m (the ancestors with key >= k form a prefix because keys never decrease from parent
// to child), lanes < m write their ancestor one generation down and lane m drops the new node into
// generation m's slot.
template <class H>
__device__ __forceinline__ bool heap_push_wave(H& h, uint32_t kbits, uint32_t vox, uint32_t src, int lane) {
  if (h.n >= h.cap) return false;
  const uint32_t pos = h.n++;
  const int sh = lane + 1 < 32 ? lane + 1 : 31;
  const uint32_t q = (pos + 1u) >> sh;
  const bool valid = (lane < 31) && q >= 1u;
  const uint32_t ai = q - 1u;
  hnode_t a;
  if (pos < H::TOP) {                     // wave uniform: the whole chain is in LDS
    a = h.top[valid ? ai : 0u];
  } else {
    const bool lo = !valid || ai < H::TOP;
    const hnode_t ag = h.node[lo ? H::TOP : ai];
    const hnode_t al = h.top[lo && valid ? ai : 0u];
    a = lo ? al : ag;
  }
  const unsigned long long climb = ballot64(valid && a.x >= kbits);
  const int m = __ffsll((long long)~climb) - 1;  // length of the leading run of set bits (lane 63 never set)
  const uint32_t dest = ((pos + 1u) >> (lane < 31 ? lane : 31)) - 1u;  // slot of generation `lane` (lane 0: the new leaf)
  const hnode_t fresh = {kbits, vox, src, 0u};
  const hnode_t val = lane < m ? a : fresh;
  if (lane <= m) {
    if (dest < H::TOP) h.top[dest] = val;
    else h.node[dest] = val;
  }
  return true;
}

// removes the top; precondition h.n > 0.  libstdc++'s __adjust_heap walks the hole to a leaf along the
// smaller child (ties: left) and then pushes the former last element up again; the result is: the path
// nodes with key < last.key move up one level and `last` takes the slot of the deepest of them.
// The wave fetches 6 levels (126 whole nodes, 2 per lane) per round trip.  A node is on the path iff it
// and all its ancestors in the sub-tree beat their siblings: one sibling compare per lane, one ballot,
// one mask test against the lane's constant ancestor mask.  The chunks nest (chunk c+1 runs inside
// chunk c, which keeps its two nodes in registers) and every write is issued on the way back up, so the
// load of `last` overlaps the whole descent.  Loads are unconditional (index clamped, key masked
// to +inf): no divergent branches in the descent.  Chunk 0 is exactly the LDS part of the heap.
// 32-bit index math is safe: the hole of chunk c sits at level 6c <= 24, so (hole+1) << 6 < 2^31.
#define KH_POP_CHUNKS 5  /* 5 * 6 = 30 levels */
template <int C, class H>
__device__ __forceinline__ void heap_pop_chunk(const H& h, uint32_t hole, uint32_t len, uint32_t vk, int lane,
                                               uint32_t& deepest, bool& found) {
  const uint32_t i0 = ((hole + 1u) << h.sh0) - 1u + h.j0;
  const uint32_t i1 = ((hole + 1u) << 6) - 1u + h.j1;
  const bool e0 = i0 < len, e1 = (lane < 62) && (i1 < len);
  hnode_t n0, n1;
  if constexpr (C < H::TOPL) { n0 = h.top[i0]; n1 = h.top[i1]; }   // an LDS chunk: i0, i1 < TOP + 3 by construction
  else { n0 = h.node[e0 ? i0 : H::TOP]; n1 = h.node[e1 ? i1 : H::TOP]; }
  const uint32_t k0 = e0 ? n0.x : INF_BITS, k1 = e1 ? n1.x : INF_BITS;
  // a node beats its sibling if it is the left one and left.key <= right.key, or the right one and
  // right.key < left.key (comp(right, left) of dijkstra_invalidation.hpp:233-237: ties go left)
  const bool w0 = e0 & (k0 < sibling_u32(k0) + h.lf);
  const bool w1 = e1 & (k1 < sibling_u32(k1) + h.lf);
  const unsigned long long W0 = ballot64(w0);
  const bool on0 = (W0 & h.am0) == h.am0;
  const bool on1 = w1 && ((W0 & h.am1) == h.am1);
  if constexpr (C + 1 < KH_POP_CHUNKS) {
    // next hole = the depth-6 node of the path, if the path got that deep and that node has children
    const unsigned long long P0 = ballot64(on0), P1 = ballot64(on1);
    uint32_t nh = 0;
    if (P1) nh = rdlane_u32(i1, __ffsll((long long)P1) - 1);
    else if (P0 >> 62) nh = rdlane_u32(i0, (P0 >> 63) ? 63 : 62);
    if (nh != 0u && 2u * nh + 1u < len) heap_pop_chunk<C + 1, H>(h, nh, len, vk, lane, deepest, found);
  }
  // every path node with key < last.key moves to its parent's slot; `last` lands in the slot of the
  // deepest such node (or the root).  Path keys are non-decreasing with depth.
  const bool mv0 = on0 && k0 < vk;
  const bool mv1 = on1 && k1 < vk;
  const uint32_t q0 = (i0 - 1u) >> 1, q1 = (i1 - 1u) >> 1;
  if constexpr (C < H::TOPL) {
    if (mv0) h.top[q0] = n0;
    if (mv1) h.top[q1] = n1;
  } else if constexpr (C == H::TOPL) {
    // the parent of this chunk's two depth-1 nodes (lanes 0, 1) is the hole: a leaf of the LDS part
    if (mv0) { if (lane < 2) h.top[hole] = n0; else h.node[q0] = n0; }
    if (mv1) h.node[q1] = n1;
  } else {
    if (mv0) h.node[q0] = n0;
    if (mv1) h.node[q1] = n1;
  }
  if (!found) {
    const unsigned long long M0 = ballot64(mv0), M1 = ballot64(mv1);
    if (M1) { deepest = rdlane_u32(i1, __ffsll((long long)M1) - 1); found = true; }
    else if (M0) { deepest = rdlane_u32(i0, 63 - __clzll((long long)M0)); found = true; }
  }
}

template <class H>
__device__ __forceinline__ void heap_pop_wave(H& h, int lane) {
  const uint32_t len = h.n - 1u;
  h.n = len;
  if (len == 0) return;
  const hnode_t last = len < H::TOP ? h.top[len] : h.node[len];  // consumed only after the descent
  uint32_t deepest = 0;
  bool found = false;
  if (len > 1u) heap_pop_chunk<0, H>(h, 0u, len, last.x, lane, deepest, found);
  if (lane == 0) {
    if (deepest < H::TOP) h.top[deepest] = last;
    else h.node[deepest] = last;
  }
}

// wave 0 only.  Returns the number of voxels invalidated.  PROF adds the pop / push / neighbour-test
// cycle split (s_memtime waits on the scalar memory counter, so the production kernel leaves it out).
// Out of line, the heap record by value: the emulation is ONE chain of dependent operations per call and every instruction the
// register allocator adds to its loops (a reloaded pointer, a lane written to a spill register) is on that chain; as a function of
// its own it is allocated for itself (inlined, 47 spilled VGPRs of the kernel cost it 8 %).
template <bool PROF, class H>
__device__ __attribute__((noinline)) uint32_t invalidate_ball(const Geometry& g, const kh_label_t* task, const uint32_t* __restrict__ nbrmask,
                                    const float* __restrict__ dbf, uint8_t* alive, const uint32_t* path, uint32_t npath,
                                    float scale, float constant, H h, uint32_t* status, uint32_t* pushes,
                                    unsigned long long* cyc3, const uint8_t* __restrict__ corner_gate = nullptr) {
  const int lane = threadIdx.x & 63;
  unsigned long long c_pop = 0, c_push = 0, c_fire = 0, tt = 0;
  h.n = 0;
  uint32_t npush = 0;
  bool ovf = false;
  for (uint32_t i = 0; i < npath; i++) {
    if (!heap_push_wave(h, 0u, path[i], i, lane)) ovf = true;
    npush++;
  }
  const uint32_t sx = (uint32_t)g.sx, sxy = (uint32_t)g.sxy;
  const uint32_t xmin = task->xmin, xmax = task->xmax;
  int dx, dy, dz;
  dir_delta(lane < 26 ? lane : 0, dx, dy, dz);
  uint32_t count = 0;
  while (h.n > 0) {
    const hnode_t top = h.top[0];
    const uint32_t vox = top.y, si = top.z;
    const uint8_t live = alive[vox];   // issued before the pop so its latency overlaps the sift-down
    if (PROF) tt = clock64();
    heap_pop_wave(h, lane);
    if (PROF) c_pop += clock64() - tt;
    if (!live) continue;
    if (PROF) tt = clock64();
    if (lane == 0) alive[vox] = 0;
    count++;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const uint32_t src = path[si];
    float maxd = scale * dbf[src];   // skeletontricks.pyx:393-395, f32 ops
    maxd = maxd + constant;
    const uint32_t z = vox / sxy, r = vox - z * sxy, y = r / sx, x = r - y * sx;
    const uint32_t oz = src / sxy, orr = src - oz * sxy, oy = orr / sx, ox = orr - oy * sx;
    // neighbour enumeration of dijkstra_invalidation.hpp:60-124 seen from the label's bounding box:
    // a corner entry (k >= 18) whose x step leaves the box degenerates into the yz diagonal.
    bool want = false;
    uint32_t q = 0;
    float nd = 0.0f;
    if (lane < 26) {
      int k = lane;
      int ex = dx;
      const bool xout = (dx < 0 && x == xmin) || (dx > 0 && x == xmax);
      if (xout) {
        if (lane >= 18) { ex = 0; k = 10 + (dy > 0 ? 2 : 0) + (dz > 0 ? 1 : 0); }
        else k = -1;
      }
      // with a voxel_connectivity_graph a degenerate corner entry is gated by the corner's bit, not the diagonal's
      const bool open = k < 0 ? false
                      : (xout && lane >= 18 && corner_gate != nullptr) ? ((corner_gate[vox] >> (lane - 18)) & 1u) != 0u
                                                                       : ((nbrmask[vox] >> k) & 1u) != 0u;
      if (open) {
        q = vox + (uint32_t)(ex + (int)sx * dy + (int)sxy * dz);
        if (alive[q]) {
          const int qx = (int)x + ex, qy = (int)y + dy, qz = (int)z + dz;
          const float a = g.wx * (float)(qx - (int)ox);
          const float b = g.wy * (float)(qy - (int)oy);
          const float c = g.wz * (float)(qz - (int)oz);
          float s = a * a;
          const float t = b * b;
          const float u = c * c;
          s = s + t;
          s = s + u;
          nd = sqrtf(s);
          want = nd < maxd;
        }
      }
    }
    unsigned long long m = ballot64(want);
    if (PROF) { c_fire += clock64() - tt; tt = clock64(); }
    const uint32_t ndb = __float_as_uint(nd);
    // The pushes of one fired voxel go to consecutive leaves, in direction order.  About three quarters of them do
    // not climb (measured: 76 % on the largest label of the bench volume): such a push writes its own leaf and
    // nothing else, and whether it climbs depends on its parent only.  So every pending lane looks at the parent
    // of the leaf it would get, the leading run of non-climbing pushes is appended with one store per lane, the
    // first climbing one goes through the ordinary push, and the rest is looked at again (its parents may have
    // changed).  A new leaf is nobody's parent here because the heap is larger than the batch.
    while (m) {
      const uint32_t base = h.n;
      const uint32_t cnt = (uint32_t)__popcll(m);
      if (base < 64u || base + cnt > h.cap) {   // small heap (new leaves could be parents) or no room: one by one
        const int k = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (!heap_push_wave(h, rdlane_u32(ndb, k), rdlane_u32(q, k), si, lane)) ovf = true;
        npush++;
        continue;
      }
      const bool mine = (m >> lane) & 1ull;
      const uint32_t leaf = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      const uint32_t par = (leaf - 1u) >> 1;
      const bool plo = !mine || par < H::TOP;
      const uint32_t kg = h.node[plo ? H::TOP : par].x;
      const uint32_t kl = h.top[plo && mine ? par : 0u].x;
      const bool stay = mine && (plo ? kl : kg) < ndb;          // __push_heap climbs while parent.key >= key
      const unsigned long long climbers = m & ~ballot64(stay);
      const int c = climbers ? __ffsll((long long)climbers) - 1 : 64;   // first lane whose push climbs
      const unsigned long long run = c < 64 ? (m & ((1ull << c) - 1ull)) : m;
      if ((run >> lane) & 1ull) {
        const hnode_t fresh = {ndb, q, si, 0u};
        if (leaf < H::TOP) h.top[leaf] = fresh;
        else h.node[leaf] = fresh;
      }
      const uint32_t nrun = (uint32_t)__popcll(run);
      h.n = base + nrun;
      npush += nrun;
      m &= ~run;
      if (c < 64) {
        m &= ~(1ull << c);
        if (!heap_push_wave(h, rdlane_u32(ndb, c), rdlane_u32(q, c), si, lane)) ovf = true;
        npush++;
      }
    }
    if (PROF) c_push += clock64() - tt;
  }
  if (lane == 0) {
    if (ovf) atomicOr(status, KH_ST_HEAP_OVERFLOW);
    *pushes += npush;
    if (PROF) { cyc3[0] += c_pop; cyc3[1] += c_push; cyc3[2] += c_fire; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  return count;
}

// wave 0 only: canonical predecessor walk (oracle ko_walk).  Writes the path (start first) to out[0..];
// returns its length (0 on failure).
//   - normally pred(v) = the achieving neighbour (fl(d[u] + f[v]) == d[v]) with d[u] < d[v] minimising (d[u], index);
//   - at the rail end of a railroad (f == 0) every achieving neighbour has d[u] == d[v]: smallest index;
//   - float-absorption plateau (all achieving neighbours have d[u] == d[v]): breadth-first search over the
//     equal-distance achieving neighbours (FIFO, neighbours in direction order) to the first voxel with a
//     strictly smaller achieving predecessor; the BFS route is followed.  bq / bpar: scratch lists (>= Nf
//     entries), visited marks live in bit 4 of the label's own qstate bytes.
// With a voxel graph (kh_apply_voxel_graph) the masks are one-way: bit k of nbrmask[v] says "a step FROM v in direction k is
// allowed".  A predecessor u of v needs the step u -> v, i.e. the bit of the OPPOSITE direction in u's word; without a graph
// the masks are symmetric and v's own word serves (one load less per step).
__device__ __forceinline__ bool pred_edge(const Geometry& g, const uint32_t* __restrict__ nbrmask, uint32_t v, int lane, bool graph) {
  if (lane >= 26) return false;
  if (!graph) return ((nbrmask[v] >> lane) & 1u) != 0u;
  int dx, dy, dz;
  dir_delta(lane, dx, dy, dz);
  const uint32_t sx = (uint32_t)g.sx, sxy = (uint32_t)g.sxy;
  const uint32_t z = v / sxy, r = v - z * sxy, y = r / sx, x = r - y * sx;
  const int nx = (int)x + dx, ny = (int)y + dy, nz = (int)z + dz;
  if (nx < 0 || ny < 0 || nz < 0 || nx >= g.sx || ny >= g.sy || nz >= g.sz) return false;
  // the opposite direction: the table of common.h lists every direction next to its opposite (0/1, 2/3, ... are pairs by
  // construction for the faces; looked up by offset for the rest)
  int opp = 0;
#pragma unroll
  for (int j = 0; j < 26; j++) {
    int ex, ey, ez;
    dir_delta(j, ex, ey, ez);
    if (ex == -dx && ey == -dy && ez == -dz) opp = j;
  }
  const uint32_t u = v + (uint32_t)g.off[lane];
  return ((nbrmask[u] >> opp) & 1u) != 0u;
}

template <bool RAILS>
__device__ __attribute__((noinline)) uint32_t backtrack(const Geometry& g, const uint32_t* __restrict__ nbrmask,
                                                        const float* __restrict__ pdrf, const float* dist,
                                                        uint32_t rail_end, uint32_t target, uint32_t* out, uint32_t cap,
                                                        uint32_t* bq, uint32_t* bpar, uint32_t bcap, uint8_t* qstate,
                                                        uint32_t* status, bool graph = false) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t v = rail_end, n = 0;
  if (lane == 0) out[0] = v;
  n = 1;
  while (v != target) {
    const float dv = ld_f32_l2(&dist[v]);
    const float fv = pdrf[v];
    const bool at_rail_end = RAILS && v == rail_end;
    unsigned long long key = NONE64;
    if (pred_edge(g, nbrmask, v, lane, graph)) {
      const uint32_t u = v + (uint32_t)g.off[lane];
      const float fu = pdrf[u];
      if (!RAILS || fu != 0.0f) {
        const float du = ld_f32_l2(&dist[u]);
        if (du != KH_INF && du + fv == dv && (at_rail_end || du < dv)) key = pack(du, u);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long ok = __shfl_xor(key, o);
      if (ok < key) key = ok;
    }
    if (key != NONE64) {
      if (n >= cap) { if (lane == 0) atomicOr(status, KH_ST_PATH_OVERFLOW); return 0; }
      v = (uint32_t)key;
      if (lane == 0) out[n] = v;
      n++;
      continue;
    }
    if (at_rail_end) { if (lane == 0) atomicOr(status, KH_ST_NO_RAIL); return 0; }
    // ---- plateau search
    uint32_t head = 0, tail = 1, found = 0xFFFFFFFFu;
    if (lane == 0) { bq[0] = v; bpar[0] = 0xFFFFFFFFu; qstate[v] |= 0x10; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    while (head < tail) {
      const uint32_t x = bq[head];
      const float dx = ld_f32_l2(&dist[x]);
      const float fx = pdrf[x];
      bool strict = false, equal = false;
      uint32_t u = 0;
      if (pred_edge(g, nbrmask, x, lane, graph)) {
        u = x + (uint32_t)g.off[lane];
        const float fu = pdrf[u];
        if (!RAILS || fu != 0.0f) {
          const float du = ld_f32_l2(&dist[u]);
          if (du != KH_INF && du + fx == dx) {
            strict = du < dx;
            equal = (du == dx) && !(qstate[u] & 0x10);
          }
        }
      }
      if (head > 0 && __ballot(strict)) { found = head; break; }
      const unsigned long long em = __ballot(equal);
      const uint32_t cnt = (uint32_t)__popcll(em);
      if (tail + cnt > bcap) { found = 0xFFFFFFFFu; head = tail; break; }
      if (equal) {
        const uint32_t p = tail + (uint32_t)__popcll(em & below);
        bq[p] = u;
        bpar[p] = head;
        qstate[u] |= 0x10;
      }
      tail += cnt;
      head++;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    for (uint32_t i = lane; i < tail; i += 64) qstate[bq[i]] &= (uint8_t)~0x10;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (found == 0xFFFFFFFFu) { if (lane == 0) atomicOr(status, KH_ST_PLATEAU); return 0; }
    // route v -> ... -> bq[found] (the BFS tree, backwards), appended forwards
    uint32_t len = 0;
    for (uint32_t i = found; i != 0; i = bpar[i]) len++;
    if (n + len > cap) { if (lane == 0) atomicOr(status, KH_ST_PATH_OVERFLOW); return 0; }
    if (lane == 0) {
      uint32_t pos = n + len - 1;
      for (uint32_t i = found; i != 0; i = bpar[i]) out[pos--] = bq[i];
    }
    n += len;
    v = bq[found];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
  return n;
}

// everything the order-free sweep needs besides the per-label task record
struct SweepGlobal {
  uint32_t on;                  // 0 = sweep disabled (every invalidation runs as the heap emulation)
  uint32_t gq, gx, gy, gz;      // integer mode (sweep.h): key^2 = gq * (gx a^2 + gy b^2 + gz c^2); gq == 0: table mode
  const uint32_t* rank;         // table mode: level table [ra * rb * rc]
  int ra, rb, rc;
  unsigned long long* cstate;   // one word per voxel, all zero on entry and on exit
  uint32_t* sched;              // one word per voxel (sweep.h, pending-deadline filter), SW_SCHED_NONE for live voxels; nullable
  unsigned char* arena;         // event arenas (per label: kh_label_t.ev_offset, in units of 256 bytes)
  uint32_t lds_levels;          // labels with more levels keep their level words in the arena instead of LDS
  uint32_t heap_prio;           // != 0: the wave that runs the heap emulation raises its issue priority (s_setprio 3)
};

// Scratch on demand (round 6).  The heap of the exact emulation (16 B x 1.5 nodes per voxel) and the ghosts' journal (8 B per voxel)
// were reserved for every label of a launch; 141 of c3's 3 402 labels ever run the emulation and 360 ever hold a ghost -- 4.6 GB per
// volume in flight for 0.4 GB of use, and the volumes in flight are bounded by memory.  With KH_TRACE_SCRATCH_POOL `heap_nodes` is
// ONE pool for the launch: node 0 = {nodes handed out so far (starts at 1), nodes in the pool}; a label takes what it needs when it
// first needs it (one atomic add by one thread).  A label the pool cannot serve ends with KH_ST_HEAP_OVERFLOW (heap) -- the host
// traces it again with scratch of its own, like every other overflow -- or goes on without ghosts (journal).
// Whole workgroup; returns nullptr when the pool is exhausted.
__device__ __forceinline__ hnode_t* pool_take(hnode_t* pool, uint32_t nodes, Ctl* ctl) {
  if (threadIdx.x == 0) {
    uint32_t* hdr = reinterpret_cast<uint32_t*>(pool);
    const uint32_t cap = hdr[1];
    uint32_t base = 0u;
    if (nodes != 0u && nodes < cap) {
      base = atomicAdd(&hdr[0], nodes);
      if (base + nodes > cap || base + nodes < base) base = 0u;
    }
    ctl->pool_base = base;
  }
  __syncthreads();
  const uint32_t b = ctl->pool_base;
  __syncthreads();
  return b ? pool + b : nullptr;
}

// One invalidation call by the whole workgroup: the order-free sweep when the label has a level table and the sweep
// certifies the call, the heap emulation (wave 0) otherwise.  Returns the number of voxels invalidated (ghosts that were killed
// not counted: sw->sh->nkg; ghosts made: sw->sh->nghost -- both zero unless allow_ghosts).
//   kill_log      where the sweep logs the voxels it kills (+ the ghosts it makes): the label's journal in ghost mode
//   allow_ghosts  a call that leaves voxels undecided ends as certified, the voxels become ghosts (sweep.h)
//   force_heap    skip the sweep (the redo of a call after a roll-back)
//   heap_ok       false while ghosts exist: the heap emulation needs the exact mask, so a call the sweep abandons cannot be
//                 redone here -- ctl->u0 = 1 tells the caller to roll back (nothing has been changed)
template <bool PROF, class H>
__device__ __forceinline__ uint32_t invalidate(Ctl* ctl, Sweep* sw, kh_label_t* task, const uint32_t* __restrict__ nbrmask,
                                               const float* __restrict__ dbf, uint8_t* alive, const uint32_t* path, uint32_t npath,
                                               float scale, float constant, H& heap, const uint32_t* list, uint32_t nf,
                                               uint32_t* sweep_stats, uint32_t heap_prio, uint32_t* kill_log,
                                               bool allow_ghosts = false, bool force_heap = false, bool heap_ok = true,
                                               const uint8_t* __restrict__ corner_gate = nullptr, hnode_t* pool = nullptr) {
  const int tid = threadIdx.x;
  bool ok = false;
  if (tid == 0) {
    sw->killed = (KH_AS_GLOBAL uint32_t*)kill_log;
    sw->sh->nkg = 0u; sw->sh->nghost = 0u;
    ctl->u0 = 0u;
  }
  __syncthreads();
  if (!force_heap && sw->nlev != 0u && npath > 0 && npath <= 32766u && npath <= nf) {
    const long long got = sweep_ball(*(const KH_AS_LDS Sweep*)sw, path, npath, dbf, scale, constant, task->sweep_rmax, list, nf, allow_ghosts);
    ok = got >= 0;
    const uint32_t cnt = ok ? (uint32_t)got : 0u;
    if (tid == 0) {
      sweep_stats[0]++;
      sweep_stats[2] += sw->sh->levels;
      sweep_stats[3] += sw->sh->events;
#ifdef KH_SWEEP_PROBE
      if (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 1000 || blockIdx.x == 2500)
        printf("SWCYC blk=%u nf=%u ok=%d lev=%u ev=%u commit=%llu next=%llu A=%llu cascA=%llu B=%llu cascB=%llu pairs=%llu "
               "dn=%llu d_own=%llu d_alive=%llu d_rank=%llu d_sched=%llu d_casc=%llu d_push=%llu\n", blockIdx.x, nf, (int)ok,
               sw->sh->levels, sw->sh->events, sw->sh->cyc[0], sw->sh->cyc[1], sw->sh->cyc[2], sw->sh->cyc[3], sw->sh->cyc[6],
               sw->sh->cyc[4], sw->sh->cyc[5], sw->sh->cyd[7], sw->sh->cyd[0], sw->sh->cyd[1], sw->sh->cyd[2], sw->sh->cyd[3],
               sw->sh->cyd[4], sw->sh->cyd[5]);
#endif
      if (!ok) { sweep_stats[1]++; sweep_stats[4] |= sw->sh->bail; sw->sh->nkg = 0u; sw->sh->nghost = 0u; }
      if (sw->sh->maxnev > task->cyc_pop) task->cyc_pop = sw->sh->maxnev;   // diagnostic: busiest level
      if (sw->sh->bump > task->cyc_push) task->cyc_push = sw->sh->bump;      // diagnostic: arena blocks used
      ctl->u1 = cnt;
    }
  } else if (tid == 0 && force_heap) {
    sweep_stats[1]++; sweep_stats[4] |= SW_BAIL_M;       // a call redone by the heap emulation after a roll-back
  }
  if (!ok && !heap_ok) {
    if (tid == 0) { ctl->u0 = 1u; ctl->u1 = 0u; }
    __syncthreads();
    return 0u;
  }
  if (!ok && heap.node == nullptr) {
    // the label's first call of the heap emulation: its heap comes out of the launch's pool now
    heap.node = pool != nullptr ? pool_take(pool, heap.cap, ctl) : nullptr;
    if (heap.node == nullptr) {
      if (tid == 0) { ctl->status |= KH_ST_HEAP_OVERFLOW; ctl->u1 = 0u; }
      __syncthreads();
      return 0u;
    }
  }
  if (!ok) {
    __syncthreads();
    if (tid < 64) {
      // The heap emulation is ONE sequential chain (a third of its time is instruction issue, the rest its own dependent
      // loads) and it sets the wall clock of the label; the waves it shares its SIMD with -- other labels' sweeps and
      // searches -- are throughput work.  With several volumes in flight they compete for issue slots all the time.
      if (heap_prio) __builtin_amdgcn_s_setprio(3);
      const uint32_t c = invalidate_ball<PROF, H>(ctl->g, task, nbrmask, dbf, alive, path, npath, scale, constant, heap,
                                                  &ctl->status, &ctl->u3, ctl->cyc3, corner_gate);
      if (heap_prio) __builtin_amdgcn_s_setprio(0);
      if (tid == 0) ctl->u1 = c;
    }
    // the sweep reads "dead" from the filter words: bring them in line with the mask the heap emulation has edited (a pass over the
    // label's voxels by the whole workgroup, once per such call -- a store per kill inside the emulation would sit on its chain)
    if (sw->sched != nullptr && sw->nlev != 0u) {
      __syncthreads();
      sweep_reset_words(*(const KH_AS_LDS Sweep*)sw, list, nf);
    }
  }
  __syncthreads();
  const uint32_t c = ctl->u1;
  __syncthreads();
  return c;
}

// thread 0: fill the workgroup's Sweep record for `task` (LDS carve-out `lds` = the dynamic shared memory).  The kernel's
// pointers get their address spaces here (sweep.h works on typed pointers only).
__device__ __forceinline__ void sweep_setup(Sweep& sw, SweepShared* swsh, Ctl* ctl, const SweepGlobal& sg, const kh_label_t* task,
                                            const uint32_t* nbrmask, uint8_t* alive, const Queues& q, uint32_t* killed,
                                            uint32_t nf, unsigned char* lds, const uint8_t* corner_gate = nullptr) {
  typedef KH_AS_GLOBAL unsigned char gbyte_t;
  uint32_t nlev = (sg.on && sg.sched != nullptr) ? task->nlev : 0u;
  sw.g = (const KH_AS_LDS Geometry*)&ctl->g;
  sw.nbrmask = (const KH_AS_GLOBAL uint32_t*)nbrmask;
  sw.alive = (KH_AS_GLOBAL uint8_t*)alive;
  sw.cstate = (KH_AS_GLOBAL unsigned long long*)sg.cstate;
  sw.sched = (KH_AS_GLOBAL uint32_t*)sg.sched;
  sw.gq = sg.gq; sw.gx = sg.gx; sw.gy = sg.gy; sw.gz = sg.gz;
  sw.rank = (const KH_AS_GLOBAL uint32_t*)sg.rank;
  sw.ra = sg.ra; sw.rb = sg.rb;
  // The searches' work lists (four of q.cap >= nf + 64 words) are free while an invalidation runs: the first is the kill log, the
  // fourth (`touched`) holds the source records, the two between them the three lists of the level being processed (20 bytes per
  // entry of each; a list that runs over abandons the call, SW_BAIL_LIST; so do more path vertices than records fit).  Round 5 kept
  // all of this in the label's heap slice, which since round 6 only exists once the label needs the heap emulation.
  {
    const uintptr_t t0 = ((uintptr_t)q.touched + 15u) & ~(uintptr_t)15u;
    sw.srcs = (KH_AS_GLOBAL u32x4_t*)(gbyte_t*)t0;
    sw.srcs_cap = (uint32_t)(((uintptr_t)(q.touched + q.cap) - t0) / 16u);
    const uintptr_t b0 = ((uintptr_t)q.b + 7u) & ~(uintptr_t)7u;
    sw.ncap = (uint32_t)(((uintptr_t)(q.b + 2u * (size_t)q.cap) - b0) / 20u);
    sw.wa = (KH_AS_GLOBAL unsigned long long*)(gbyte_t*)b0;
    sw.np = sw.wa + sw.ncap;
    sw.wb = (KH_AS_GLOBAL uint32_t*)(sw.np + sw.ncap);
  }
  // level words + non-empty bitmap live in LDS: a window of them (kh_label_t.lev_window), or one per level when that fits the
  // launch's allotment; a label for which neither does runs without the sweep (heap emulation only)
  const uint32_t win = task->lev_window;
  const bool windowed = win >= 64u && (win & (win - 1u)) == 0u && win <= sg.lds_levels;   // round-robin words
  if (!windowed && nlev > sg.lds_levels) nlev = 0u;
  sw.nslots = windowed ? win : nlev;
  sw.wmask = windowed ? win - 1u : 0xFFFFFFFFu;
  // arena: [spill table: ev_spill keys (u32) + ev_spill candidate words (u64)][(free stack of rounds 4-5: unused)][chunks]
  gbyte_t* fsp = (gbyte_t*)(sg.arena + (size_t)task->ev_offset * 256u);
  const uint32_t spcap = task->ev_spill;          // 0 or a power of two
  sw.spcap = (spcap & (spcap - 1u)) == 0u ? spcap : 0u;
  sw.spc = (KH_AS_GLOBAL unsigned long long*)fsp;
  sw.spk = (KH_AS_GLOBAL uint32_t*)(fsp + (size_t)sw.spcap * 8u);
  fsp += (((size_t)sw.spcap * 12u) + 255u) & ~(size_t)255u;
  sw.chunks = (KH_AS_GLOBAL u32x2_t*)(fsp + ((((size_t)task->ev_chunks * 4u) + 255u) & ~(size_t)255u));
  sw.chcap = task->ev_chunks;
  sw.shift = (int)task->ev_shift;
  sw.killed = (KH_AS_GLOBAL uint32_t*)killed;              // the search work lists are free during an invalidation
  sw.nlev = nlev;                                          // 0: this label's invalidations run as the heap emulation
  sw.chain = (KH_AS_LDS uint32_t*)lds;
  sw.fs = sw.chain + SW_CHAIN;
  sw.words = sw.fs + SW_RING;
  sw.lvbits = sw.words + sw.nslots;
  sw.sh = (KH_AS_LDS SweepShared*)swsh;
  sw.gate = (const KH_AS_GLOBAL uint8_t*)corner_gate;
  sw.xmin = task->xmin; sw.xmax = task->xmax;
}

// Ghosts and roll-back.  A call of the sweep that leaves voxels undecided no longer runs the heap emulation
// at once: the voxels become ghosts (sweep.h) and the loop goes on -- everything the later calls decide holds under either status
// of a ghost, and most ghosts are killed for certain by a later ball.  The loop itself must never act on a ghost: when a ghost
// could be the next target, when no certainly-valid voxel is left while ghosts are, or when a later call would need the heap
// emulation (which needs the exact mask), the label ROLLS BACK to the call that made the first ghost -- the journal holds every
// voxel that changed since (killed or made a ghost: all alive again), the rails of the later paths get their weights back -- and
// that call is redone by the exact heap emulation; then the loop continues from there, without ghosts.
struct GhostState {
  uint32_t nghost;       // ghosts alive now
  uint32_t jpos;         // journal entries (0 whenever no ghost exists)
  uint32_t paths, verts, valid, nb, na;   // the state before the call that made the first ghost
};

template <bool PROF, int TOPL>
#ifndef KH_TRACE_WAVES_PER_EU
#define KH_TRACE_WAVES_PER_EU 3   /* 168 VGPRs; 4 -> 128 VGPRs with 66 of them spilled */
#endif
__global__ __launch_bounds__(256, KH_TRACE_WAVES_PER_EU) void trace_paths_kernel(kh_label_t* tasks, const uint32_t* __restrict__ lists,
                                                          float* list_daf,
                                                          const uint32_t* __restrict__ nbrmask, Geometry g,
                                                          const float* __restrict__ dbf, float* pdrf, float* dist,
                                                          uint8_t* alive, uint8_t* qstate,
                                                          const uint32_t* __restrict__ manual_targets,
                                                          float scale, float constant, uint32_t* queues, hnode_t* heap_nodes,
                                                          uint32_t* path_vertices,
                                                          uint32_t* path_lengths, int fix_branching, SweepGlobal sg,
                                                          uint32_t* journal_buf, float* rail_save, uint32_t ghost_mode,
                                                          const uint8_t* __restrict__ corner_gate, uint32_t lds_bytes) {
  __shared__ Ctl ctl;
  __shared__ Sweep sw;
  __shared__ SweepShared swsh;
  __shared__ uint32_t sweep_stats[5];
  // (Heap<TOPL>::TOP + 3) nodes for the heap emulation; the sweep's level words and lists use the same bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char heap_top[];
  kh_label_t* task = &tasks[blockIdx.x];
  const int tid = threadIdx.x;
  const int nthr = blockDim.x, nwav = nthr >> 6;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t* list = lists + task->list_offset;
  float* ldaf = list_daf + task->list_offset;
  const uint32_t nf = task->count;
  Queues q;
  q.cap = task->q_capacity;
  q.a = queues + (uint64_t)task->q_offset * 4;
  q.b = q.a + q.cap;
  q.c = q.b + q.cap;
  q.touched = q.c + q.cap;
  Heap<TOPL> heap;
  hnode_t* const pool = (ghost_mode & 8u) ? heap_nodes : nullptr;      // KH_TRACE_SCRATCH_POOL: heap and journal on demand
  heap.node = pool ? nullptr : heap_nodes + task->heap_offset;
  heap.top = (lds_hnode_t*)heap_top;
  heap.cap = task->heap_capacity;
  heap.n = 0;
  heap_init_lane(heap, lane);
  uint32_t* pverts = path_vertices + task->path_offset;
  uint32_t* plens = path_lengths + task->path_offset;
  float* psave = rail_save ? rail_save + task->path_offset : nullptr;        // the weight a path vertex had before it became a rail
  uint32_t* journal = (journal_buf && !pool) ? journal_buf + (uint64_t)task->q_offset * 2 : nullptr;   // 2 * q_capacity entries; pool: on demand
  bool journal_off = false;            // the pool could not serve this label's journal: its ghost calls are rolled back at once
  const uint32_t pcap = task->path_capacity;
  uint32_t root = task->root;          // (0xFFFFFFFF with KH_TRACE_FUSED_EDF: found below, trace.py:291-308)
  uint32_t daf_loc = 0xFFFFFFFFu;      // KH_TRACE_FUSED_EDF: the voxel farthest from the root (the implicit first target, trace.py:160-172)
  const uint32_t* before = manual_targets + task->tgt_offset;
  const uint32_t* after = before + task->n_before;
  const bool soma = task->soma_mode != 0;
  const bool implicit = task->n_before == 0 && !soma;   // trace.py:160-172
  uint32_t nb = implicit ? 1u : task->n_before;
  uint32_t na = task->n_after;
  uint32_t valid = nf;
  uint32_t npaths = 0, nverts = 0;
  unsigned long long t_target = 0, t_rail = 0, t_inval = 0, t0 = 0;
  // ghost mode: bit 0 = on, bit 1 = every call that makes a ghost is rolled back at once (tests of the roll-back itself)
  const bool ghosts_on = (ghost_mode & 1u) != 0u && (journal != nullptr || pool != nullptr) && (psave != nullptr || !fix_branching) &&
                         sg.on != 0u && sg.sched != nullptr;
  const bool paranoid = (ghost_mode & 2u) != 0u;
  const bool graph = (ghost_mode & 4u) != 0u;     // the neighbour masks carry a voxel graph: predecessor edges are one-way
  GhostState gs = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
  uint32_t n_ghost_calls = 0, n_rollbacks = 0;
  if (tid == 0) {
    ctl.status = 0; ctl.u2 = 0; ctl.u3 = 0; ctl.cyc3[0] = ctl.cyc3[1] = ctl.cyc3[2] = 0; ctl.g = g;
    for (int i = 0; i < 5; i++) sweep_stats[i] = 0;
    sweep_setup(sw, &swsh, &ctl, sg, task, nbrmask, alive, q, q.a, nf, heap_top, corner_gate);
  }
  __syncthreads();
  // the spill table of the sweep starts all-free (the arena is uninitialised memory)
  for (uint32_t i = tid; i < sw.spcap; i += nthr) { sw.spk[i] = 0u; sw.spc[i] = 0ull; }
  __syncthreads();
  if (ghost_mode & 16u) {
    // KH_TRACE_FUSED_EDF (round 6): the label's two distance-field searches and its PDRF run HERE, by the label's own workgroup,
    // instead of as three launches over all labels in front of this kernel -- with volumes in flight every launch waits for the
    // slots the others' path workgroups hold, and the chain of a volume (its largest label that needs the heap emulation) could
    // only start once the searches of ALL its labels were through: 2.3 s into a 8.6 s round of twenty volumes.
    // find_root (trace.py:291-308), DAF from the root (:139-145) into `dist` (+inf again afterwards), then per voxel of the list:
    // the DAF for the target finder and compute_pdrf (:315-356, the repeated-squaring branch), every operation rounded to f32.
    const float delta_floor = 2.0f * fminf(g.wx, fminf(g.wy, g.wz));
    if (root == 0xFFFFFFFFu) {
      edf_label(&ctl, task, 1, list, nf, nbrmask, dist, qstate, q, delta_floor, task->source, heap_top, lds_bytes);
      root = 0xFFFFFFFFu - (uint32_t)ctl.red64[0];         // (every thread: the record was written by thread 0 only)
      __syncthreads();
    }
    edf_label(&ctl, task, 2, list, nf, nbrmask, dist, qstate, q, delta_floor, root, heap_top, lds_bytes);
    const float max_daf = __uint_as_float((uint32_t)(ctl.red64[0] >> 32));
    daf_loc = 0xFFFFFFFFu - (uint32_t)ctl.red64[0];
    const float M = task->M, pscale = task->pdrf_scale;
    const int nsq = (int)task->pdrf_log2e;
    __syncthreads();
    for (uint32_t i = tid; i < nf; i += nthr) {
      const uint32_t v = list[i];
      float d = ld_f32_l2(&dist[v]);
      ldaf[i] = d;
      float p = dbf[v] * M;        // np.multiply(DBF, M)            trace.py:341
      p = 1.0f - p;                // np.subtract(f(1), PDRF)        trace.py:342
      for (int sq = 0; sq < nsq; sq++) p = p * p;   //              trace.py:343-345
      p = p * pscale;              // PDRF *= f(pdrf_scale)          trace.py:349
      if (d == KH_INF) d = 0.0f;   // inf2zero                       trace.py:146
      if (max_daf != 0.0f) {
        const float inv = 1.0f / max_daf;   // (1 / max_daf) in float32 (numpy 2 scalar)  trace.py:353
        d = d * inv;
        p = p + d;                 //                                trace.py:354
      }
      pdrf[v] = p;
      st_f32_l2(&dist[v], KH_INF);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
  }
  if (soma) {
    // trace.py:160-168: one-off invalidation around the soma centre, before valid_labels is counted (:211)
    if (tid == 0) pverts[0] = root;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    valid -= invalidate<PROF, Heap<TOPL>>(&ctl, &sw, task, nbrmask, dbf, alive, pverts, 1, task->soma_scale, task->soma_const,
                                          heap, list, nf, sweep_stats, sg.heap_prio, q.a, false, false, true, corner_gate, pool);   // trace.py:211 counts what is left
  }
  const uint32_t max_paths = task->max_paths ? task->max_paths : valid;  // trace.py:214-215
  if (nb + na >= max_paths) {                           // trace.py:217-218
    if (tid == 0) { task->n_paths = 0; task->n_vertices = 0; task->status |= ctl.status; }
    return;
  }
  if (fix_branching) {
    if (tid == 0) pdrf[root] = 0.0f;                    // trace.py:220 (initial rail)
  } else {
    // trace.py:155: one weighted Dijkstra from the root; every path is then a predecessor walk
    sssp<2>(ctl.g, nbrmask, pdrf, dist, qstate, root, q, &ctl, 0.0f, heap_top, lds_bytes);
  }
  __syncthreads();
  bool redo = false;      // this iteration redoes the invalidation of path `npaths` by the heap emulation (after a roll-back)
  for (;;) {
    if (npaths >= max_paths) break;
    bool rollback = false;
    uint32_t plen = 0;
    uint32_t* out = pverts + nverts;
    if (gs.nghost > 0 && (valid == 0 || paranoid || journal_off)) rollback = true;   // nothing certainly valid is left, but ghosts are
    if (!rollback && !redo) {
      if (!(valid > 0 || nb > 0 || na > 0)) break;
      // ---- target selection, trace.py:225-230
      t0 = clock64();
      uint32_t target;
      if (nb > 0) { nb--; target = implicit ? (daf_loc != 0xFFFFFFFFu ? daf_loc : task->max_loc) : before[nb]; }
      else if (valid == 0) { na--; target = after[na]; }
      else {
        // CachedTargetFinder.find_target: the valid voxel with the largest DAF (ties: largest index)
        unsigned long long best = 0;
        for (uint32_t i = tid; i < nf; i += nthr) {
          const uint32_t v = list[i];
          if (!alive[v]) continue;
          const unsigned long long key = ((unsigned long long)__float_as_uint(ldaf[i]) << 32) | v;
          if (key >= best) best = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long ob = __shfl_xor(best, o);
          if (ob > best) best = ob;
        }
        if (lane == 0) ctl.red64[wave] = best;
        __syncthreads();
        best = ctl.red64[0];
        for (int i = 1; i < nwav; i++) if (ctl.red64[i] > best) best = ctl.red64[i];
        target = (uint32_t)best;
        __syncthreads();
        if (gs.nghost > 0 && alive[target] == SW_GHOST) rollback = true;   // a ghost could be the target: its status decides
      }
      if (!rollback) {
      // ---- railroad, trace.py:240-242
      t_target += clock64() - t0; t0 = clock64();
      if (nverts >= pcap || npaths >= pcap) {
        if (tid == 0) atomicOr(&ctl.status, KH_ST_PATH_OVERFLOW);
        __syncthreads();
        break;
      }
      if (!fix_branching) {
        // dijkstra3d.path_from_parents (trace.py:244): walk target -> root, return root -> target
        if (tid == 0) ctl.u0 = 0;
        __syncthreads();
        if (wave == 0) {
          const uint32_t n = backtrack<false>(ctl.g, nbrmask, pdrf, dist, target, root, out, pcap - nverts, q.a, q.b, q.cap,
                                              qstate, &ctl.status, graph);
          for (uint32_t i = lane; i < n / 2; i += 64) { const uint32_t a = out[i]; out[i] = out[n - 1 - i]; out[n - 1 - i] = a; }
          if (lane == 0) ctl.u0 = n;
        }
        __syncthreads();
        plen = ctl.u0;
        if (plen == 0) break;
      } else if (pdrf[target] == 0.0f) {
        if (tid == 0) out[0] = target;
        plen = 1;
      } else {
        sssp<1>(ctl.g, nbrmask, pdrf, dist, qstate, target, q, &ctl, 0.0f, heap_top, lds_bytes);
        const unsigned long long br = ctl.best_rail;
        if (tid == 0) { ctl.u0 = 0; ctl.u2 += ctl.n_touched; }
        __syncthreads();
        if (br == NONE64) {
          if (tid == 0) atomicOr(&ctl.status, KH_ST_NO_RAIL);
        } else if (wave == 0) {
          const uint32_t n = backtrack<true>(ctl.g, nbrmask, pdrf, dist, (uint32_t)br, target, out, pcap - nverts, q.a, q.b,
                                             q.cap, qstate, &ctl.status, graph);
          if (lane == 0) ctl.u0 = n;
        }
        __syncthreads();
        plen = ctl.u0;
        // restore dist = +inf and the queue flags on everything the search touched
        const uint32_t nt = ctl.n_touched < q.cap ? ctl.n_touched : q.cap;
        for (uint32_t i = tid; i < nt; i += nthr) {
          const uint32_t v = q.touched[i];
          st_f32_l2(&dist[v], KH_INF);
          qstate[v] = 0;
        }
        __syncthreads();
        if (plen == 0) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      if (soma) {
        // trace.py:246-251: path = concat(path[:1], path[dist_to_soma_root > soma_radius]) -- path[0] is kept
        // unconditionally AND again if it passes the test itself (the reference duplicates it then).  The
        // distance is float64 like numpy's (float32 anisotropy * int64 offsets -> float64, norm in float64).
        uint32_t* tmp = q.touched;  // free between searches; capacity >= Nf + 64 > plen + 1
        if (wave == 0) {
          const uint32_t sxu = (uint32_t)ctl.g.sx, sxy = (uint32_t)ctl.g.sxy;
          const uint32_t rz = root / sxy, rr = root - rz * sxy, ry = rr / sxu, rx = rr - ry * sxu;
          const double sr = (double)task->soma_radius;
          uint32_t kept = 1;
          if (lane == 0) tmp[0] = out[0];
          for (uint32_t b0 = 0; b0 < plen; b0 += 64) {
            const uint32_t i = b0 + lane;
            bool keep = false;
            uint32_t v = 0;
            if (i < plen) {
              v = out[i];
              const uint32_t z = v / sxy, r = v - z * sxy, y = r / sxu, x = r - y * sxu;
              const double ax = (double)ctl.g.wx * (double)((long long)x - (long long)rx);
              const double ay = (double)ctl.g.wy * (double)((long long)y - (long long)ry);
              const double az = (double)ctl.g.wz * (double)((long long)z - (long long)rz);
              const double d = sqrt(ax * ax + ay * ay + az * az);
              keep = d > sr;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) tmp[kept + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = v;
            kept += (uint32_t)__popcll(m);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (kept > pcap - nverts) { if (lane == 0) atomicOr(&ctl.status, KH_ST_PATH_OVERFLOW); kept = 0; }
          for (uint32_t i = lane; i < kept; i += 64) out[i] = tmp[i];
          if (lane == 0) ctl.u0 = kept;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        plen = ctl.u0;
        __syncthreads();
        if (plen == 0) break;
      }
      t_rail += clock64() - t0;
      }
    } else if (!rollback) {
      plen = __hip_atomic_load(&plens[npaths], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the path whose invalidation is redone
    }
    // ---- invalidation, trace.py:253-259
    t0 = clock64();
    if (!rollback && valid > 0) {
      if (gs.nghost == 0) {           // (the state a roll-back returns to, should this call make the first ghost)
        gs.jpos = 0; gs.paths = npaths; gs.verts = nverts; gs.valid = valid; gs.nb = nb; gs.na = na;
      }
      const bool go_ghost = ghosts_on && !redo;
      // the call's kill log: the journal once it exists and a ghost is alive; else the first work list (a call that makes the
      // FIRST ghost has its log copied to the journal below -- most labels never need one)
      const bool to_journal = go_ghost && journal != nullptr;
      const uint32_t killed = invalidate<PROF, Heap<TOPL>>(&ctl, &sw, task, nbrmask, dbf, alive, out, plen, scale, constant, heap,
                                                          list, nf, sweep_stats, sg.heap_prio, to_journal ? journal + gs.jpos : q.a,
                                                          go_ghost, redo, gs.nghost == 0, corner_gate, pool);
      if (ctl.u0 != 0u) {
        rollback = true;                                   // the sweep abandoned the call while ghosts exist
      } else {
        const uint32_t made = swsh.nghost, gkilled = swsh.nkg;
        valid -= killed + made;                            // the voxels that are certainly valid
        if (go_ghost) {
          if (made != 0u && journal == nullptr) {
            // the label's first ghost: its journal comes out of the pool now and takes this call's log over
            const uint32_t nlog = killed + gkilled + made;
            __syncthreads();
            journal = journal_off ? nullptr : reinterpret_cast<uint32_t*>(pool_take(pool, (2u * q.cap + 3u) / 4u, &ctl));
            if (journal != nullptr) {
              for (uint32_t i = tid; i < nlog; i += nthr) journal[i] = q.a[i];
            } else {
              journal = q.a;                               // no room: this call is rolled back from its own log, at the top of the
              journal_off = true;                          // next iteration (its path has to be on record first: plens[npaths])
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
          }
          gs.nghost += made;
          gs.nghost -= gkilled;
          gs.jpos = gs.nghost ? gs.jpos + killed + gkilled + made : 0u;
          if (made) n_ghost_calls++;
        }
      }
      __syncthreads();                                     // (swsh / ctl are rewritten by the next call)
    }
    t_inval += clock64() - t0;
    if (rollback) {
      // every voxel the journal holds is alive again, the rails of the later paths get their weights back
      for (uint32_t i = tid; i < gs.jpos; i += nthr) {
        const uint32_t v = journal[i];
        alive[v] = 1;
        if (sg.sched != nullptr) sg.sched[v] = SW_SCHED_NONE;
      }
      const uint32_t keep = gs.verts + __hip_atomic_load(&plens[gs.paths], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the path of the call being redone stays a rail
      if (fix_branching) for (uint32_t i = keep + tid; i < nverts; i += nthr) { const float w = psave[i]; if (w != 0.0f) pdrf[pverts[i]] = w; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      npaths = gs.paths; nverts = gs.verts; valid = gs.valid; nb = gs.nb; na = gs.na;
      gs.nghost = 0; gs.jpos = 0;
      if (journal_off) journal = nullptr;
      n_rollbacks++;
      redo = true;
      continue;
    }
    redo = false;
    // ---- rails, trace.py:261-263
    if (fix_branching) for (uint32_t i = tid; i < plen; i += nthr) {
      const uint32_t v = out[i];
      if (psave) psave[nverts + i] = pdrf[v];
      pdrf[v] = 0.0f;
    }
    if (tid == 0) plens[npaths] = plen;
    npaths++;
    nverts += plen;
    __syncthreads();
    if (ctl.status) break;
  }
  __syncthreads();
  if (!fix_branching) {  // leave dist = +inf behind
    for (uint32_t i = tid; i < nf; i += nthr) st_f32_l2(&dist[list[i]], KH_INF);
  }
  if (tid == 0) {
    task->n_paths = npaths;
    task->n_vertices = nverts;
    task->status |= ctl.status;
    task->stat_settled = ctl.u2;
    task->stat_heap_pushes = ctl.u3;
    task->cyc_target = (uint32_t)(t_target >> 10);
    task->cyc_rail = (uint32_t)(t_rail >> 10);
    task->cyc_inval = (uint32_t)(t_inval >> 10);
    if (PROF) task->cyc_pop = (uint32_t)(ctl.cyc3[0] >> 10);
    if (PROF) task->cyc_push = (uint32_t)(ctl.cyc3[1] >> 10);
    task->cyc_fire = (uint32_t)(ctl.cyc3[2] >> 10);
    task->stat_sweep_calls = sweep_stats[0];
    task->stat_sweep_bails = sweep_stats[1];
    task->stat_sweep_levels = sweep_stats[2];
    task->stat_sweep_events = sweep_stats[3];
    task->stat_sweep_why = sweep_stats[4];
    task->stat_ghost_calls = n_ghost_calls;
    task->stat_rollbacks = n_rollbacks;
  }
}

// ------------------------------------------------------------------------------------------------
// a9 on its own: roll_invalidation_ball_inside_component (skeletontricks.pyx:373-418) for ONE object, the same device
// routine the path loop uses (order-free sweep, heap emulation as the fall-back).
// (the same register bound as the path kernel: the sweep's out-of-line functions are shared, and they inherit the bound of their
// callers only when every kernel that reaches them has it)
__global__ __launch_bounds__(256, KH_TRACE_WAVES_PER_EU) void invalidate_ball_kernel(kh_label_t* task, const uint32_t* __restrict__ lists,
                                                              const uint32_t* __restrict__ nbrmask, Geometry g,
                                                              const float* __restrict__ dbf, uint8_t* alive, uint32_t* queues,
                                                              hnode_t* heap_nodes, const uint32_t* __restrict__ path, uint32_t npath,
                                                              float scale, float constant, SweepGlobal sg, long long* invalidated,
                                                              const uint8_t* __restrict__ corner_gate) {
  __shared__ Ctl ctl;
  __shared__ Sweep sw;
  __shared__ SweepShared swsh;
  __shared__ uint32_t sweep_stats[5];
  extern __shared__ __attribute__((aligned(16))) unsigned char heap_top[];
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t nf = task->count;
  Heap<1> heap;
  heap.node = heap_nodes + task->heap_offset;
  heap.top = (lds_hnode_t*)heap_top;
  heap.cap = task->heap_capacity;
  heap.n = 0;
  heap_init_lane(heap, lane);
  if (tid == 0) {
    ctl.status = 0; ctl.u1 = 0; ctl.u3 = 0; ctl.cyc3[0] = ctl.cyc3[1] = ctl.cyc3[2] = 0; ctl.g = g;
    for (int i = 0; i < 5; i++) sweep_stats[i] = 0;
    Queues q;
    q.cap = task->q_capacity;
    q.a = queues + (uint64_t)task->q_offset * 4;
    q.b = q.a + q.cap;
    q.c = q.b + q.cap;
    q.touched = q.c + q.cap;
    sweep_setup(sw, &swsh, &ctl, sg, task, nbrmask, alive, q, q.a, nf, heap_top, corner_gate);
  }
  __syncthreads();
  for (uint32_t i = tid; i < sw.spcap; i += blockDim.x) { sw.spk[i] = 0u; sw.spc[i] = 0ull; }   // (the arena is uninitialised memory)
  // the filter words say which voxels of the caller's mask are dead (kh_trace_paths keeps them up to date itself)
  if (sw.sched != nullptr) sweep_reset_words(*(const KH_AS_LDS Sweep*)&sw, lists + task->list_offset, nf);
  __syncthreads();
  const uint32_t c = invalidate<false, Heap<1>>(&ctl, &sw, task, nbrmask, dbf, alive, path, npath, scale, constant, heap,
                                               lists + task->list_offset, nf, sweep_stats, sg.heap_prio,
                                               queues + (uint64_t)task->q_offset * 4, false, false, true, corner_gate);
  if (tid == 0) {
    *invalidated = (long long)c;
    task->status |= ctl.status;
    task->stat_heap_pushes = ctl.u3;
    task->stat_sweep_calls = sweep_stats[0];
    task->stat_sweep_bails = sweep_stats[1];
    task->stat_sweep_levels = sweep_stats[2];
    task->stat_sweep_events = sweep_stats[3];
    task->stat_sweep_why = sweep_stats[4];
  }
}

// ------------------------------------------------------------------------------------------------
// a7 / a8 on their own (the path loop has them inside trace_paths_kernel): one object, one search.
//   mode 0  dijkstra3d.railroad(field, source): search from `src` over the field to the nearest zero-weight voxel,
//           path written rail end first (trace.py:240-242); dist is +inf again on exit.
//   mode 1  the search of dijkstra3d.parental_field(field, source) (trace.py:155): leaves the distance field in `dist`.
//   mode 2  dijkstra3d.path_from_parents (trace.py:244) on that distance field: path src(root) -> dst(target).
__global__ __launch_bounds__(256, KH_TRACE_WAVES_PER_EU) void path_search_kernel(kh_label_t* task, int mode, const uint32_t* __restrict__ lists,
                                                          const uint32_t* __restrict__ nbrmask, Geometry g, const float* pdrf,
                                                          float* dist, uint8_t* qstate, uint32_t* queues, uint32_t src, uint32_t dst,
                                                          uint32_t* out, uint32_t cap, uint32_t* out_n, int graph) {
  __shared__ Ctl ctl;
  __shared__ __attribute__((aligned(16))) unsigned char search_lds[KH_SSSP_LDS_BYTES * 256];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  Queues q;
  q.cap = task->q_capacity;
  q.a = queues + (uint64_t)task->q_offset * 4;
  q.b = q.a + q.cap;
  q.c = q.b + q.cap;
  q.touched = q.c + q.cap;
  if (tid == 0) { ctl.status = 0; ctl.u0 = 0; ctl.g = g; }
  __syncthreads();
  if (mode == 1) {
    sssp<2>(ctl.g, nbrmask, pdrf, dist, qstate, src, q, &ctl, 0.0f, search_lds, (uint32_t)sizeof(search_lds));
  } else if (mode == 2) {
    if (wave == 0) {
      const uint32_t n = backtrack<false>(ctl.g, nbrmask, pdrf, dist, dst, src, out, cap, q.a, q.b, q.cap, qstate, &ctl.status, graph != 0);
      for (uint32_t i = lane; i < n / 2; i += 64) { const uint32_t a = out[i]; out[i] = out[n - 1 - i]; out[n - 1 - i] = a; }
      if (lane == 0) ctl.u0 = n;
    }
  } else if (pdrf[src] == 0.0f) {
    if (tid == 0) { out[0] = src; ctl.u0 = 1; }
  } else {
    sssp<1>(ctl.g, nbrmask, pdrf, dist, qstate, src, q, &ctl, 0.0f, search_lds, (uint32_t)sizeof(search_lds));
    const unsigned long long br = ctl.best_rail;
    if (br == NONE64) {
      if (tid == 0) atomicOr(&ctl.status, KH_ST_NO_RAIL);
    } else if (wave == 0) {
      const uint32_t n = backtrack<true>(ctl.g, nbrmask, pdrf, dist, (uint32_t)br, src, out, cap, q.a, q.b, q.cap, qstate, &ctl.status, graph != 0);
      if (lane == 0) ctl.u0 = n;
    }
    __syncthreads();
    const uint32_t nt = ctl.n_touched < q.cap ? ctl.n_touched : q.cap;
    for (uint32_t i = tid; i < nt; i += nthr) { const uint32_t v = q.touched[i]; st_f32_l2(&dist[v], KH_INF); qstate[v] = 0; }
  }
  __syncthreads();
  if (tid == 0) { *out_n = ctl.u0; task->status |= ctl.status; }
}

// ------------------------------------------------------------------------------------------------
// a7 as arrays: dijkstra3d.parental_field(field, source) returns a parents ARRAY (linear index of the predecessor + 1, 0 = none)
// which trace.py edits (`parents[tuple(root)] = 0`, kimimaro/trace.py:220) and hands to path_from_parents (:244).
// parents_kernel: after the search of path_search_kernel mode 1 has left the distances in `dist`, one thread per voxel of the
// object applies the canonical predecessor rule (oracle ko_pred: the neighbour u with fl(d[u] + f[v]) == d[v] minimising
// (d[u], index of u)).  A voxel all of whose achieving neighbours lie at ITS OWN distance (a float-absorption plateau) has no
// parent that a pointer chase could follow without cycles: KH_ST_PLATEAU, like the oracle's KO_EPLATEAU (the fused path loop and
// kh_path_search mode 2 cross such plateaus by a breadth-first search; a parents array cannot say that).
__global__ __launch_bounds__(256) void parents_kernel(kh_label_t* task, const uint32_t* __restrict__ lists,
                                                      const uint32_t* __restrict__ nbrmask, Geometry g,
                                                      const float* __restrict__ field, const float* __restrict__ dist,
                                                      uint32_t source, uint32_t* parents, int graph) {
  const uint32_t nf = task->count;
  const uint32_t* list = lists + task->list_offset;
  const uint32_t sx = (uint32_t)g.sx, sxy = (uint32_t)g.sxy;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
    const uint32_t v = list[i];
    const float dv = dist[v];
    if (v == source || dv == KH_INF) { parents[v] = 0u; continue; }
    const float fv = field[v];
    const uint32_t nm = nbrmask[v];
    const uint32_t z = v / sxy, r = v - z * sxy, y = r / sx, x = r - y * sx;
    unsigned long long best = NONE64;
    for (int k = 0; k < 26; k++) {
      int dx, dy, dz;
      sweep_dir(k, dx, dy, dz);
      const int nx = (int)x + dx, ny = (int)y + dy, nz = (int)z + dz;
      if (nx < 0 || ny < 0 || nz < 0 || nx >= g.sx || ny >= g.sy || nz >= g.sz) continue;
      const uint32_t u = v + (uint32_t)(dx + (int)sx * dy + (int)sxy * dz);
      bool edge;
      if (!graph) {
        edge = ((nm >> k) & 1u) != 0u;           // symmetric masks: v's own word serves
      } else {
        int opp = 0;                              // the step u -> v is gated by u's word (one-way edges)
        for (int j = 0; j < 26; j++) {
          int ex, ey, ez;
          sweep_dir(j, ex, ey, ez);
          if (ex == -dx && ey == -dy && ez == -dz) opp = j;
        }
        edge = ((nbrmask[u] >> opp) & 1u) != 0u;
      }
      if (!edge) continue;
      const float du = dist[u];
      if (du == KH_INF || du + fv != dv) continue;
      const unsigned long long key = pack(du, u);
      if (key < best) best = key;
    }
    if (best == NONE64) { parents[v] = 0u; continue; }
    if (!(__uint_as_float((uint32_t)(best >> 32)) < dv)) { atomicOr(&task->status, KH_ST_PLATEAU); parents[v] = 0u; continue; }
    parents[v] = (uint32_t)best + 1u;
  }
}

// dijkstra3d.path_from_parents(parents, target) (kimimaro/trace.py:244): a pointer chase target -> source, returned source first.
// One wave: lane 0 chases (a chain of dependent loads by nature), the wave reverses.
__global__ __launch_bounds__(64) void path_from_parents_kernel(const uint32_t* __restrict__ parents, uint32_t nvox, uint32_t target,
                                                               uint32_t* out, uint32_t cap, uint32_t* out_n) {
  __shared__ uint32_t n_sh;
  const int lane = threadIdx.x;
  if (lane == 0) {
    uint32_t n = 0, v = target;
    for (;;) {
      if (n >= cap) { n = 0; break; }             // no room (or a cycle in a caller-edited array): empty path
      out[n++] = v;
      const uint32_t p = parents[v];
      if (p == 0u || p > nvox) break;
      v = p - 1u;
    }
    n_sh = n;
  }
  __syncthreads();
  const uint32_t n = n_sh;
  for (uint32_t i = lane; i < n / 2; i += 64) { const uint32_t a = out[i]; out[i] = out[n - 1 - i]; out[n - 1 - i] = a; }
  if (lane == 0) *out_n = n;
}

// ------------------------------------------------------------------------------------------------
// a10: roll_invalidation_cube.  One workgroup per path vertex; bytes are cleared with a 32-bit
// atomicAnd so each voxel is counted exactly once however many boxes overlap.
__global__ __launch_bounds__(256) void invalidate_cube_kernel(uint8_t* mask, const float* __restrict__ dbf, int sx, int sy,
                                                              int sz, float wx, float wy, float wz,
                                                              const uint64_t* __restrict__ path, float scale,
                                                              float constant, unsigned long long* invalidated) {
  const uint64_t loc = path[blockIdx.x];
  const int64_t sxy = (int64_t)sx * sy;
  float radius = scale * dbf[loc];
  radius = radius + constant;
  const int64_t z = (int64_t)(loc / (uint64_t)sxy), r = (int64_t)(loc % (uint64_t)sxy), y = r / sx, x = r % sx;
  const float rr[3] = {radius / wx, radius / wy, radius / wz};
  const int64_t c[3] = {x, y, z};
  const int64_t s[3] = {sx, sy, sz};
  int64_t lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float fl = (float)c[a] - rr[a];              // skeletontricks.hpp:97-102
    int64_t l = (int64_t)fl;
    if (l < 0) l = 0;
    const float fh = (float)c[a] + rr[a];
    const double dh = 0.5 + (double)fh;
    int64_t h = (int64_t)dh;
    if (h > s[a] - 1) h = s[a] - 1;
    lo[a] = l; hi[a] = h;
  }
  const int64_t nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
  unsigned long long cnt = 0;
  if (nx > 0 && ny > 0 && nz > 0) {
    const int64_t total = nx * ny * nz;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
      const int64_t xx = lo[0] + i % nx, yy = lo[1] + (i / nx) % ny, zz = lo[2] + i / (nx * ny);
      const int64_t qi = xx + sx * yy + sxy * zz;
      uint32_t* word = reinterpret_cast<uint32_t*>(mask + (qi & ~3ll));
      const int sh = (int)(qi & 3) * 8;
      if ((*reinterpret_cast<volatile uint8_t*>(mask + qi)) == 0) continue;
      const uint32_t old = atomicAnd(word, ~(0xFFu << sh));
      if ((old >> sh) & 0xFFu) cnt++;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(invalidated, cnt);
}

}  // namespace kh

using namespace kh;

extern "C" int kh_edf_batch(kh_label_t* tasks, int ntasks, int mode, const uint32_t* lists, const uint32_t* nbrmask,
                            int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz, float* field,
                            uint8_t* qstate, uint32_t* queues, void* stream) {
  if (int rc = require_device()) return rc;
  if (ntasks <= 0) return KH_OK;
  if (sx * sy * sz >= (1ll << 32)) { set_error("kh_edf_batch: volume must have < 2^32 voxels"); return KH_EINVAL; }
  if (((uintptr_t)qstate & 3) != 0) { set_error("kh_edf_batch: qstate must be 4-byte aligned"); return KH_EINVAL; }
  const int nthreads = (mode >> 8) ? (mode >> 8) : 512;
  mode &= 0xFF;
  if (mode < 0 || mode > 2 || nthreads < 64 || nthreads > 1024 || (nthreads & 63)) {
    set_error("kh_edf_batch: mode must be 0, 1 or 2 (+ threads per label << 8: a multiple of 64 up to 1024)");
    return KH_EINVAL;
  }
  Geometry g;
  make_geometry(g, sx, sy, sz, wx, wy, wz);
  float mn = wx < wy ? wx : wy;
  if (wz < mn) mn = wz;
  const float delta_floor = 2.0f * mn;
  // 512 threads per label: measured 0.176 / 0.135 / 0.131 s for the two runs at c3 with 256 / 512 / 1024 threads
  // (one volume alone; with volumes in flight the lanes ask for fewer: a workgroup needs all its waves' slots on one CU at
  // once, and next to thousands of one-wave path workgroups eight free slots rarely come together)
  hipLaunchKernelGGL(edf_batch_kernel, dim3(ntasks), dim3(nthreads), KH_SSSP_LDS_BYTES * nthreads, (hipStream_t)stream, tasks, mode, lists, nbrmask, g, field,
                     qstate, queues, delta_floor);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

namespace kh {
// dynamic LDS of a workgroup that runs the sweep with `nlev` level words: chunk chain, free-chunk stack, level words, bitmap
static size_t sweep_lds_bytes(uint32_t nlev) {
  return ((size_t)SW_CHAIN + SW_RING + nlev + (nlev >> 5) + 2) * 4;
}
// The sweep's launch-wide arguments from the entry points' (level_rank, ra, rb, rc): a table of ranks [ra, rb, rc], or --
// level_rank == NULL and ra > 0 -- integer mode with (gx, gy, gz) = (ra, rb, rc): the squared voxel pitches divided by their
// greatest common divisor gq (checked here against wx, wy, wz).  Returns false on inconsistent arguments.
static bool sweep_global(SweepGlobal& sg, const uint32_t* level_rank, int64_t ra, int64_t rb, int64_t rc, float wx, float wy, float wz,
                         uint64_t* cstate, uint32_t* sched, void* event_arena, int64_t max_nlev) {
  sg.on = 0u; sg.gq = sg.gx = sg.gy = sg.gz = 0u;
  sg.rank = level_rank;
  sg.ra = (int)ra; sg.rb = (int)rb; sg.rc = (int)rc;
  sg.cstate = reinterpret_cast<unsigned long long*>(cstate);
  sg.sched = sched;
  sg.arena = reinterpret_cast<unsigned char*>(event_arena);
  sg.lds_levels = (uint32_t)max_nlev;
  sg.heap_prio = 0u;
  if (level_rank == nullptr && ra <= 0) return true;                    // sweep off
  if (!cstate || !sched || !event_arena || ra <= 0 || rb <= 0 || rc <= 0 || max_nlev < 0 || max_nlev > KH_SWEEP_LDS_LEVELS ||
      ((uintptr_t)event_arena & 255) != 0)
    return false;
  if (level_rank == nullptr) {
    const double qx = (double)wx * wx, qy = (double)wy * wy, qz = (double)wz * wz;
    const double gq = qx / (double)ra;
    if (!(gq >= 1.0) || gq != (double)(uint32_t)gq || gq * (double)rb != qy || gq * (double)rc != qz || qx > 4.0e9 || qy > 4.0e9 ||
        qz > 4.0e9)
      return false;
    sg.gq = (uint32_t)gq; sg.gx = (uint32_t)ra; sg.gy = (uint32_t)rb; sg.gz = (uint32_t)rc;
  }
  sg.on = 1u;
  return true;
}
template <bool PROF, int TOPL = 1>
static int launch_trace(int count, hipStream_t st, kh_label_t* tasks, const uint32_t* lists, float* list_daf,
                        const uint32_t* nbrmask, const Geometry& g, const float* dbf, float* pdrf, float* dist,
                        uint8_t* alive, uint8_t* qstate, const uint32_t* manual_targets, float scale, float constant,
                        uint32_t* queues, hnode_t* heap_nodes, uint32_t* path_vertices, uint32_t* path_lengths,
                        int fix_branching, const SweepGlobal& sg, uint32_t max_nlev, unsigned nthreads, uint32_t* journal,
                        float* rail_save, uint32_t ghost_mode, const uint8_t* corner_gate) {
  if (count <= 0) return KH_OK;
  size_t lds = (size_t)(Heap<TOPL>::TOP + 3) * sizeof(hnode_t);
  const size_t swl = sweep_lds_bytes(max_nlev);
  if (sg.on && swl > lds) lds = swl;
  if ((size_t)KH_SSSP_LDS_BYTES * nthreads > lds) lds = (size_t)KH_SSSP_LDS_BYTES * nthreads;     // the searches' scratch (same bytes)
  {
    // More than 48 KiB of dynamic LDS has to be allowed per kernel.  The attribute belongs to the function, not to the
    // launch, and several host threads launch at once (kimimaro_amd/lanes.py): always the same value -- the largest a
    // launch can ask for -- so that a concurrent caller never lowers it under somebody else's launch.
    const size_t lds_max = sweep_lds_bytes(KH_SWEEP_LDS_LEVELS);
    KH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_paths_kernel<PROF, TOPL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_max > lds ? lds_max : lds)));
  }
  hipLaunchKernelGGL((trace_paths_kernel<PROF, TOPL>), dim3(count), dim3(nthreads), lds, st, tasks, lists, list_daf, nbrmask,
                     g, dbf, pdrf, dist, alive, qstate, manual_targets, scale, constant, queues, heap_nodes, path_vertices,
                     path_lengths, fix_branching, sg, journal, rail_save, ghost_mode, corner_gate, (uint32_t)lds);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

// key of the offset (a, b, c) exactly as the flood computes it (dijkstra_invalidation.hpp:310-316)
__global__ __launch_bounds__(256) void level_keys_kernel(int ra, int rb, int rc, float wx, float wy, float wz, float* keys) {
  const int64_t n = (int64_t)ra * rb * rc;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int a = (int)(i % ra), b = (int)((i / ra) % rb), c = (int)(i / ((int64_t)ra * rb));
  const float fa = wx * (float)a, fb = wy * (float)b, fc = wz * (float)c;
  float s = fa * fa;
  const float t = fb * fb;
  const float u = fc * fc;
  s = s + t;
  s = s + u;
  keys[i] = sqrtf(s);
}
}  // namespace kh

extern "C" int kh_level_keys(int64_t ra, int64_t rb, int64_t rc, float wx, float wy, float wz, float* keys, void* stream) {
  if (int rc2 = require_device()) return rc2;
  if (ra <= 0 || rb <= 0 || rc <= 0 || ra * rb * rc >= (1ll << 31)) { set_error("kh_level_keys: bad table size"); return KH_EINVAL; }
  const int64_t n = ra * rb * rc;
  hipLaunchKernelGGL(level_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)ra, (int)rb,
                     (int)rc, wx, wy, wz, keys);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_trace_paths(kh_label_t* tasks, int ntasks, const uint32_t* lists, float* list_daf,
                              const uint32_t* nbrmask, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                              const float* dbf, float* pdrf, float* dist, uint8_t* alive, uint8_t* qstate,
                              const uint32_t* manual_targets, float scale, float constant, uint32_t* queues,
                              void* heap_nodes, uint32_t* path_vertices, uint32_t* path_lengths,
                              const uint32_t* level_rank, int64_t ra, int64_t rb, int64_t rc, int64_t max_nlev,
                              uint64_t* cstate, uint32_t* sched, void* event_arena, uint32_t* journal, float* rail_save,
                              const uint8_t* corner_gate, int flags, int fix_branching, void* stream) {
  if (int rc2 = require_device()) return rc2;
  if (ntasks <= 0) return KH_OK;
  if (sx * sy * sz >= (1ll << 32)) { set_error("kh_trace_paths: volume must have < 2^32 voxels"); return KH_EINVAL; }
  if (((uintptr_t)qstate & 3) != 0) { set_error("kh_trace_paths: qstate must be 4-byte aligned"); return KH_EINVAL; }
  if (((uintptr_t)heap_nodes & 15) != 0) { set_error("kh_trace_paths: heap_nodes must be 16-byte aligned"); return KH_EINVAL; }
  if ((flags & ~(KH_TRACE_PROFILE | KH_TRACE_HEAP_PRIO | KH_TRACE_THREADS_64 | KH_TRACE_THREADS_128 | KH_TRACE_NO_GHOSTS |
                 KH_TRACE_GHOST_PARANOID | KH_TRACE_BIG_LDS_HEAP | KH_TRACE_VOXEL_GRAPH | KH_TRACE_SCRATCH_POOL | KH_TRACE_FUSED_EDF)) ||
      ((flags & KH_TRACE_THREADS_64) && (flags & KH_TRACE_THREADS_128))) {
    set_error("kh_trace_paths: unknown flags");
    return KH_EINVAL;
  }
  const unsigned nthreads = (flags & KH_TRACE_THREADS_64) ? 64u : (flags & KH_TRACE_THREADS_128) ? 128u : 256u;
  Geometry g;
  make_geometry(g, sx, sy, sz, wx, wy, wz);
  SweepGlobal sg;
  if (!sweep_global(sg, level_rank, ra, rb, rc, wx, wy, wz, cstate, sched, event_arena, max_nlev)) {
    set_error("kh_trace_paths: sweep arguments (level table or integer-mode factors, cstate, sched, a 256-byte aligned event arena, max_nlev)");
    return KH_EINVAL;
  }
  sg.heap_prio = (flags & KH_TRACE_HEAP_PRIO) ? 1u : 0u;
  hipStream_t st = (hipStream_t)stream;
  const bool prof = (flags & KH_TRACE_PROFILE) != 0;
  // ghosts need the journal (and, with rails, the saved weights); bit 1: roll every ghost call back at once
  const bool use_pool = (flags & KH_TRACE_SCRATCH_POOL) != 0;
  const uint32_t ghost_mode = ((journal || use_pool) && !(flags & KH_TRACE_NO_GHOSTS) ? 1u : 0u) | ((flags & KH_TRACE_GHOST_PARANOID) ? 2u : 0u) |
                              ((flags & KH_TRACE_VOXEL_GRAPH) ? 4u : 0u) | (use_pool ? 8u : 0u) | ((flags & KH_TRACE_FUSED_EDF) ? 16u : 0u);
  if (flags & KH_TRACE_BIG_LDS_HEAP)
    return launch_trace<false, 2>(ntasks, st, tasks, lists, list_daf, nbrmask, g, dbf, pdrf, dist, alive, qstate, manual_targets,
                                  scale, constant, queues, (hnode_t*)heap_nodes, path_vertices, path_lengths, fix_branching, sg,
                                  (uint32_t)max_nlev, nthreads, journal, rail_save, ghost_mode, corner_gate);
  return prof ? launch_trace<true>(ntasks, st, tasks, lists, list_daf, nbrmask, g, dbf, pdrf, dist, alive, qstate, manual_targets,
                                   scale, constant, queues, (hnode_t*)heap_nodes, path_vertices, path_lengths, fix_branching, sg,
                                   (uint32_t)max_nlev, nthreads, journal, rail_save, ghost_mode, corner_gate)
              : launch_trace<false>(ntasks, st, tasks, lists, list_daf, nbrmask, g, dbf, pdrf, dist, alive, qstate, manual_targets,
                                    scale, constant, queues, (hnode_t*)heap_nodes, path_vertices, path_lengths, fix_branching, sg,
                                    (uint32_t)max_nlev, nthreads, journal, rail_save, ghost_mode, corner_gate);
}

extern "C" int kh_invalidate_ball(kh_label_t* task, const uint32_t* lists, const uint32_t* nbrmask, int64_t sx, int64_t sy,
                                  int64_t sz, float wx, float wy, float wz, const float* dbf, uint8_t* alive, uint32_t* queues,
                                  void* heap_nodes, const uint32_t* path, int64_t npath, float scale, float constant,
                                  const uint32_t* level_rank, int64_t ra, int64_t rb, int64_t rc, int64_t max_nlev,
                                  uint64_t* cstate, uint32_t* sched, void* event_arena, const uint8_t* corner_gate,
                                  int64_t* invalidated, void* stream) {
  if (int rc2 = require_device()) return rc2;
  if (!task || !lists || !nbrmask || !dbf || !alive || !queues || !heap_nodes || !path || !invalidated || npath < 0 ||
      npath >= (1ll << 32) || sx * sy * sz >= (1ll << 32) || ((uintptr_t)heap_nodes & 15) != 0) {
    set_error("kh_invalidate_ball: bad arguments");
    return KH_EINVAL;
  }
  Geometry g;
  make_geometry(g, sx, sy, sz, wx, wy, wz);
  SweepGlobal sg;
  if (!sweep_global(sg, level_rank, ra, rb, rc, wx, wy, wz, cstate, sched, event_arena, max_nlev)) {
    set_error("kh_invalidate_ball: sweep arguments (level table or integer-mode factors, cstate, sched, a 256-byte aligned event arena, max_nlev)");
    return KH_EINVAL;
  }
  size_t lds = (size_t)(Heap<1>::TOP + 3) * sizeof(hnode_t);
  const size_t swl = sweep_lds_bytes((uint32_t)max_nlev);
  if (sg.on && swl > lds) lds = swl;
  {
    // (constant value: see launch_trace)
    const size_t lds_max = sweep_lds_bytes(KH_SWEEP_LDS_LEVELS);
    KH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&invalidate_ball_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_max > lds ? lds_max : lds)));
  }
  hipLaunchKernelGGL(invalidate_ball_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, task, lists, nbrmask, g, dbf, alive, queues,
                     (hnode_t*)heap_nodes, path, (uint32_t)npath, scale, constant, sg, (long long*)invalidated, corner_gate);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_path_search(kh_label_t* task, int mode, const uint32_t* lists, const uint32_t* nbrmask, int64_t sx, int64_t sy,
                              int64_t sz, float wx, float wy, float wz, const float* field, float* dist, uint8_t* qstate,
                              uint32_t* queues, uint64_t source, uint64_t target, uint32_t* path, int64_t path_capacity,
                              uint32_t* path_length, int voxel_graph, void* stream) {
  if (int rc2 = require_device()) return rc2;
  if (!task || !lists || !nbrmask || !field || !dist || !qstate || !queues || !path_length || mode < 0 || mode > 2 ||
      (mode != 1 && !path) || sx * sy * sz >= (1ll << 32) || path_capacity < 0 || path_capacity >= (1ll << 32) ||
      ((uintptr_t)qstate & 3) != 0) {
    set_error("kh_path_search: bad arguments");
    return KH_EINVAL;
  }
  Geometry g;
  make_geometry(g, sx, sy, sz, wx, wy, wz);
  hipLaunchKernelGGL(path_search_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, task, mode, lists, nbrmask, g, field, dist, qstate,
                     queues, (uint32_t)source, (uint32_t)target, path, (uint32_t)path_capacity, path_length, voxel_graph);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_invalidate_cube(uint8_t* mask, const float* dbf, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                                  float wz, const uint64_t* path, int64_t npath, float scale, float constant,
                                  int64_t* invalidated, void* stream) {
  if (int rc = require_device()) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (((uintptr_t)mask & 3) != 0) { set_error("kh_invalidate_cube: mask must be 4-byte aligned"); return KH_EINVAL; }
  KH_HIP_CHECK(hipMemsetAsync(invalidated, 0, sizeof(int64_t), st));
  if (npath <= 0) return KH_OK;
  hipLaunchKernelGGL(invalidate_cube_kernel, dim3((unsigned)npath), dim3(256), 0, st, mask, dbf, (int)sx, (int)sy, (int)sz, wx,
                     wy, wz, path, scale, constant, (unsigned long long*)invalidated);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_parental_field(kh_label_t* task, const uint32_t* lists, const uint32_t* nbrmask, int64_t sx, int64_t sy, int64_t sz,
                                 const float* field, float* dist, uint8_t* qstate, uint32_t* queues, uint64_t source,
                                 uint32_t* parents, int voxel_graph, void* stream) {
  if (int rc2 = require_device()) return rc2;
  if (!task || !lists || !nbrmask || !field || !dist || !qstate || !queues || !parents || sx * sy * sz >= (1ll << 32) ||
      source >= (uint64_t)(sx * sy * sz) || ((uintptr_t)qstate & 3) != 0) {
    set_error("kh_parental_field: bad arguments");
    return KH_EINVAL;
  }
  Geometry g;
  make_geometry(g, sx, sy, sz, 1.0f, 1.0f, 1.0f);     // (field costs: the edge lengths are not used)
  hipStream_t st = (hipStream_t)stream;
  KH_HIP_CHECK(hipMemsetAsync(parents, 0, (size_t)(sx * sy * sz) * sizeof(uint32_t), st));
  hipLaunchKernelGGL(path_search_kernel, dim3(1), dim3(256), 0, st, task, 1, lists, nbrmask, g, field, dist, qstate, queues,
                     (uint32_t)source, 0u, (uint32_t*)nullptr, 0u, parents /* out_n: overwritten below */, voxel_graph);
  KH_LAUNCH_CHECK();
  hipLaunchKernelGGL(parents_kernel, dim3(256), dim3(256), 0, st, task, lists, nbrmask, g, field, dist, (uint32_t)source, parents,
                     voxel_graph);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_path_from_parents(const uint32_t* parents, int64_t nvox, uint64_t target, uint32_t* path, int64_t path_capacity,
                                    uint32_t* path_length, void* stream) {
  if (int rc2 = require_device()) return rc2;
  if (!parents || !path || !path_length || nvox <= 0 || nvox >= (1ll << 32) || target >= (uint64_t)nvox || path_capacity <= 0 ||
      path_capacity >= (1ll << 32)) {
    set_error("kh_path_from_parents: bad arguments");
    return KH_EINVAL;
  }
  hipLaunchKernelGGL(path_from_parents_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, parents, (uint32_t)nvox, (uint32_t)target,
                     path, (uint32_t)path_capacity, path_length);
  KH_LAUNCH_CHECK();
  return KH_OK;
}
