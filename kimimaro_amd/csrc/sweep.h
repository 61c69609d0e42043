// sweep.h -- order-free evaluation of roll_invalidation_ball_inside_component for gfx950
//            (skeletontricks.pyx:373-418 -> dijkstra_invalidation.hpp:239-332), included by trace.hip.
//
// The reference's flood is a priority-first search whose keys are NOT monotone (key = straight-line distance of a
// voxel from the path vertex that owns the node), and among equal keys std::priority_queue's array layout decides
// which node pops first.  The set of voxels that end up invalidated can depend on that order, which is
// why the reference can only be reproduced in general by emulating the libstdc++ heap (trace.hip, invalidate_ball).
//
// Model P: a multiset of nodes (key, source c, voxel v); pop ANY node of minimal key; if v is alive, kill it
// (owner = c) and push (|q - p_c|, c, q) for every alive same-component neighbour q with |q - p_c| < r_c.
// libstdc++'s heap is one resolution of "ANY".  This file evaluates P for ALL resolutions at once and proves, call
// by call, that the dead set does not depend on the resolution; only when the proof fails does the kernel fall
// back to the heap emulation.  (Measured on the bench volumes: > 95 % of the calls, > 95 % of the voxels certify.)
//
// Levels.  Keys take few distinct values: sqrt(fl(fl((wx*a)^2 + (wy*b)^2) + (wz*c)^2)) for integer offsets (a, b, c).
// The host tabulates them once per call of skeletonize (kh_level_keys + sort/unique): rank[|a|, |b|, |c|] = index of
// the key in the sorted list of distinct keys.  Rank is monotone in the key, so "level" = rank and a level is
// exactly one key.  P never pops a key of level > L while a node of level <= L is pending, so an execution is a
// sequence of levels (the running maximum of the popped keys), inside which the order is arbitrary.
//
// Abstract sweep (sound for every resolution; argument in DESIGN.md 3.4):
//   voxel states   A alive under every resolution, M may be dead, D dead under every resolution.
//   P event (L, c, v)   "from level L on a node (c, v) may pop": if v is not D yet, c joins Cand(v) (A -> M) and
//                       (v, c) emits P events to the neighbours c covers, at level max(L, rank of their key)
//                       (the same level = a cascade, resolved inside the level).
//   D event (L, v)      "v is dead once level L is complete": v becomes D; for every neighbour q that ALL of Cand(v)
//                       cover, a D event at max(L, largest rank of q's keys over Cand(v)).
//   A P and a D event of the same (voxel, level) from a voxel with one candidate travel as one PD event (the
//   common case: unique owner).  Path vertices start with a PD event at level 0.
// Certified iff no voxel is left in state M and no capacity was exceeded; then the dead set is D.
//
// Machine mapping: one workgroup per label (the caller's: one to four waves), one thread per event of the current level;
// per-voxel state = one 64-bit word in HBM (4 x 15-bit candidates + the "dying in this level" bit) + one 32-bit word (the
// earliest pending deadline: the filter below), events = 8-byte records in per-level chunk lists (HBM, label-private arena,
// chunks recycled level by level), the list heads of a WINDOW of levels and a non-empty-level bitmap in LDS, the cascade
// lists in HBM.  Every pointer carries its address space in its type (common.h).  No float atomics, no MFMA: irregular
// integer/f32 work bound by dependent round trips to L2 / HBM (about ten per level).
#pragma once
#include "common.h"

namespace kh {

static constexpr uint32_t SW_NONE = 0xFFFFFFFFu;
static constexpr unsigned long long SW_DYING = 1ull << 63;
static constexpr unsigned long long SW_SPILL = 1ull << 15;      // the voxel's fifth to eighth possible owners are in the spill table
static constexpr uint8_t SW_GHOST = 3;                          // alive byte of a voxel that is dead under SOME resolutions of an earlier call
static constexpr uint32_t SW_P = 1u << 16, SW_D = 1u << 17;     // event meta = source index (15 bits) | type
#ifndef KH_SW_CHAIN
#define KH_SW_CHAIN 512
#endif
#ifndef KH_SW_RING
#define KH_SW_RING 256
#endif
static constexpr uint32_t SW_CHAIN = KH_SW_CHAIN;                        // chunks per level the reader can take (LDS list)
static constexpr uint32_t SW_RING = KH_SW_RING;                         // free chunk ids kept in LDS (the chunks of finished levels)
static constexpr int SW_FILL_BITS = 12;                          // level word = (newest chunk << SW_FILL_BITS) | next free slot.  The fill
static constexpr uint32_t SW_FILL_MASK = (1u << SW_FILL_BITS) - 1u;   // field keeps counting while a chunk is full: up to 13 adds per thread (sweep_push26)
static constexpr uint32_t SW_NOCHUNK = 0xFFFFFu;                 // of a 256-thread workgroup + the chunk size must fit it; 20-bit chunk ids
static constexpr uint32_t SW_SCHED_NONE = 0xFFFFFFFFu;           // filter word of a live voxel without a pending deadline
static constexpr uint32_t SW_SCHED_DEAD = 0u;                    // filter word of a dead voxel (no deadline value is 0: only a source has level 0)
// bail reasons (kh_label_t.stat_sweep_bail is the OR over the label's calls)
static constexpr uint32_t SW_BAIL_M = 1, SW_BAIL_CAND = 2, SW_BAIL_ARENA = 4, SW_BAIL_LEVEL = 8, SW_BAIL_LIST = 16,
                          SW_BAIL_UNTOUCHED = 32;

struct SweepShared {
  uint32_t lvl, nev, ord, openid;
  uint32_t na, nb, nnp, bump, nkill, snap, bail;
  uint32_t nkg;        // killed voxels that were ghosts (they do not count as invalidated: the caller's count of valid voxels never held them)
  uint32_t nspill;     // voxels with an entry in the spill table (the table is cleared at the end of a call that used it)
  uint32_t nghost;     // out: voxels this call left undecided (ghosts made)
  int32_t nM;
  int32_t nfree;       // chunks on the free stack (transiently negative while lanes that found it empty put their claim back)
  uint32_t nprev;      // chunks of the level being processed: they go to the free stack when the next level is looked for
  uint32_t levels, events, maxnev;
#ifdef KH_SWEEP_PROBE
  unsigned long long cyc[8];   // developer probe: cycles per phase of the level loop (thread 0's clock)
  unsigned long long cyd[8];   // ... and of thread 0's own deadline event: loads / keys / claims / cascade / pushes; [6] scratch, [7] = events
#endif
};

#ifdef KH_SWEEP_PROBE
#define SW_D0() do { if (threadIdx.x == 0) { s.sh->cyd[6] = (unsigned long long)clock64(); s.sh->cyd[7]++; } } while (0)
#define SW_DT(i, val) do { asm volatile("" :: "v"(val)); if (threadIdx.x == 0) { const unsigned long long n_ = (unsigned long long)clock64(); s.sh->cyd[i] += n_ - s.sh->cyd[6]; s.sh->cyd[6] = n_; } } while (0)
#else
#define SW_D0()
#define SW_DT(i, val)
#endif

struct Sweep {
  // uniform over the workgroup.  The record itself lives in LDS; its pointers carry their address space (common.h)
  const KH_AS_LDS Geometry* g;     // LDS copy
  const KH_AS_GLOBAL uint32_t* nbrmask;
  KH_AS_GLOBAL uint8_t* alive;
  KH_AS_GLOBAL unsigned long long* cstate;
  KH_AS_GLOBAL uint32_t* sched;    // per voxel: SW_SCHED_NONE = alive, no deadline pending; SW_SCHED_DEAD = dead; else the pending
                                   // deadline (level << cb | source + 1).  The word BEFORE the volume's first and the one behind its last
                                   // must be readable (the neighbours are read as rows of three).  nullptr: no sweep
  // levels.  Integer mode (gq != 0): the anisotropy is integral and the squared distances of all offsets a ball can reach are exact
  // in float (the host checks), so key^2 = gq * S with S = gx a^2 + gy b^2 + gz c^2 an integer, sqrtf is strictly monotone on those
  // values, and S itself serves as the level: no table, no float operation per neighbour.  Table mode (gq == 0): rank[] as before.
  uint32_t gq, gx, gy, gz;
  const KH_AS_GLOBAL uint32_t* rank;   // [ra * rb * rc] (table mode)
  int ra, rb;
  KH_AS_GLOBAL u32x4_t* srcs;      // per path vertex {x, y, z, radius bits (table mode) / number of levels its ball covers (integer mode)}
  uint32_t srcs_cap;               // records that fit (a call with more path vertices runs as the heap emulation)
  KH_AS_GLOBAL u32x2_t* chunks;    // arena: chunk c = slots [c << shift, (c + 1) << shift); slot 0 = {previous chunk of the level, -}
  KH_AS_LDS uint32_t* fs;          // [SW_RING] free stack: ids of chunks whose level has been processed (LDS; what does not fit is not reused)
  uint32_t chcap;                  // chunks available
  int shift;                       // log2(slots per chunk), <= 7
  KH_AS_GLOBAL uint32_t* killed;   // HBM log of the voxels killed by this call
  uint32_t nlev;
  // LDS (a label whose level words do not fit the launch's allotment runs without the sweep)
  KH_AS_LDS uint32_t* words;       // [nslots] (newest chunk << SW_FILL_BITS) | next free slot of level lv at words[lv & wmask]
  KH_AS_LDS uint32_t* lvbits;      // [nslots / 32 + 1] non-empty levels (bit lv & wmask)
  uint32_t nslots, wmask;          // level window: nslots = a power of two and wmask = nslots - 1 when every pending event
                                   // lies less than nslots levels ahead of the level being processed (the slots are then
                                   // used round robin); nslots = nlev, wmask = all ones otherwise
  KH_AS_LDS uint32_t* chain;       // [SW_CHAIN] chunks of the level being processed, newest first
  KH_AS_LDS SweepShared* sh;
  // HBM lists of the level being processed (label-private scratch)
  KH_AS_GLOBAL unsigned long long* wa;   // [ncap] candidate cascade (voxel << 32 | source)
  KH_AS_GLOBAL unsigned long long* np;   // [ncap] pairs added by pure P events (their voxel may survive the level)
  KH_AS_GLOBAL uint32_t* wb;             // [ncap] deadline cascade
  uint32_t ncap;
  // voxel_connectivity_graph only (nullptr otherwise): bit j = corner entry 18 + j of the voxel is allowed AND the yz diagonal it
  // degenerates into at an x face of the label's array exists (kh_apply_voxel_graph); xmin / xmax = those faces
  const KH_AS_GLOBAL uint8_t* gate;
  uint32_t xmin, xmax;
  // candidate spill: open-addressing table keyed by voxel, for the voxels with more than four possible owners (front of the arena)
  KH_AS_GLOBAL uint32_t* spk;              // [spcap] voxel + 1, 0 = free
  KH_AS_GLOBAL unsigned long long* spc;    // [spcap] four more 15-bit candidate slots, laid out like the voxel's own word
  uint32_t spcap;                          // a power of two, 0 = no table (a fifth candidate abandons the call)
};
typedef const KH_AS_LDS Sweep& SweepRef;   // the workgroup's record (LDS)

// atomics on address-space-qualified pointers (HIP's atomicAdd & co take generic ones); relaxed, like those
#define SW_L_ADD(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SW_L_OR(p, v) __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SW_G_ADD(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SW_G_OR(p, v) __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SW_G_AND(p, v) __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
template <class T>
__device__ __forceinline__ T sw_g_cas(KH_AS_GLOBAL T* p, T expect, T want) {      // returns the old value like atomicCAS
  __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expect;
}

__device__ __forceinline__ void sweep_bail(SweepRef s, uint32_t why) { SW_L_OR(&s.sh->bail, why); }
// The level words and the bitmap are updated with atomics (LDS): read them the same way.  Global words that other waves of the
// workgroup store to are read past the CU's vector cache (sc1: served by the L2).
__device__ __forceinline__ uint32_t sweep_ld(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t sweep_ld(const KH_AS_LDS uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t sweep_ld(const KH_AS_GLOBAL uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- keys and levels.  The key of voxel q from source c is the straight-line distance with the float operation order of
// dijkstra_invalidation.hpp:310-316; q is inside c's ball iff key < r_c (strict).  Levels only have to order the keys and tell
// equal ones: table mode looks the rank of the offset up, integer mode uses the exact integer S = key^2 / gq (see Sweep).
struct SweepKeys {     // the per-call constants of the key evaluation, in registers
  uint32_t gq, gx, gy, gz;
  float wx, wy, wz;
  const KH_AS_GLOBAL uint32_t* rank;
  int ra, rb;
};
__device__ __forceinline__ SweepKeys sweep_keys(SweepRef s) {
  SweepKeys k;
  k.gq = s.gq; k.gx = s.gx; k.gy = s.gy; k.gz = s.gz;
  k.wx = s.g->wx; k.wy = s.g->wy; k.wz = s.g->wz;
  k.rank = s.rank; k.ra = s.ra; k.rb = s.rb;
  return k;
}
// the float key itself (what the reference computes)
__device__ __forceinline__ float sweep_fkey(const SweepKeys& K, int ex, int ey, int ez) {
  const float a = K.wx * (float)ex;
  const float b = K.wy * (float)ey;
  const float c = K.wz * (float)ez;
  float t = a * a;
  const float u = b * b;
  const float v = c * c;
  t = t + u;
  t = t + v;
  return sqrtf(t);
}
// is the voxel at offset (ex, ey, ez) from the source inside its ball, and at which level does its key lie
__device__ __forceinline__ bool sweep_eval(const SweepKeys& K, const u32x4_t src, int ex, int ey, int ez, uint32_t& rk) {
  if (K.gq) {
    const uint32_t S = K.gx * (uint32_t)(ex * ex) + K.gy * (uint32_t)(ey * ey) + K.gz * (uint32_t)(ez * ez);
    rk = S;
    return S < src.w;
  }
  const float d = sweep_fkey(K, ex, ey, ez);
  if (!(d < __uint_as_float(src.w))) return false;
  const int ax = ex < 0 ? -ex : ex, ay = ey < 0 ? -ey : ey, az = ez < 0 ? -ez : ez;
  rk = K.rank[ax + K.ra * (ay + K.rb * az)];
  return true;
}
// integer mode: the number of levels a ball of radius r covers = the smallest S whose key is not below r
__device__ __forceinline__ uint32_t sweep_slim(uint32_t gq, float r) {
  if (!(r > 0.0f)) return 0u;
  const float est = r * r / (float)gq;
  uint32_t S = est < 4.0e9f ? (uint32_t)est : 0xFFFFFFFEu;
  while (S < 0xFFFFFFFEu && sqrtf((float)((unsigned long long)gq * S)) < r) S++;
  while (S > 0u && !(sqrtf((float)((unsigned long long)gq * (S - 1u))) < r)) S--;
  return S;
}

// direction k by a run-time index: 2 bits per component and direction in three 64-bit constants (from dir_delta's table)
constexpr unsigned long long sweep_dir_word(int comp) {
  unsigned long long w = 0;
  for (int k = 0; k < 26; k++) {
    int d[3] = {0, 0, 0};
    dir_delta(k, d[0], d[1], d[2]);
    w |= (unsigned long long)(d[comp] + 1) << (2 * k);
  }
  return w;
}
__device__ __forceinline__ void sweep_dir(int k, int& dx, int& dy, int& dz) {
  constexpr unsigned long long WX = sweep_dir_word(0), WY = sweep_dir_word(1), WZ = sweep_dir_word(2);
  dx = (int)((WX >> (2 * k)) & 3ull) - 1;
  dy = (int)((WY >> (2 * k)) & 3ull) - 1;
  dz = (int)((WZ >> (2 * k)) & 3ull) - 1;
}

// Event storage.  A level's events live in a chain of fixed-size chunks (slot 0 of a chunk = id of the previous
// chunk of the level); the level's LDS word holds the newest chunk and its next free slot.
// sweep_push appends an event to level lv.  Free of wait states (a lane never waits for another lane's future action,
// which lock-step execution could not deliver).  Fast path: one atomic add on the level word takes the next slot -- no
// retry however many lanes push to the same level at once (a CAS here costs a retry per concurrent pusher, and a busy
// level has hundreds).  A lane whose add finds the chunk full (or the level empty: the empty word reads as full)
// installs a fresh chunk with a CAS on the overfull word and takes its slot 1; whoever loses that race just starts over,
// and the chunk it had reserved stays with the thread (`spare`) for its next opening.  The fill field of a full chunk
// keeps counting (at most 13 adds per thread before the install, sweep_push26: < 1 << SW_FILL_BITS), readers clamp it to the chunk size.
// Level window.  An event goes to a level ahead of the one being processed (`cur`), and never far ahead: its key is the
// distance of a NEIGHBOUR of the processed voxel from a source whose key of that voxel is not above the current level, so
// it exceeds the current key by one step at most -- a few hundred to a few thousand levels, which the host bounds
// (kh_label_t.lev_window).  The level words are therefore kept for a window of nslots levels only, level lv in
// slot lv & wmask: 4-8 KiB of LDS instead of 4 bytes for every level of the label.  The bound is checked, not trusted: an
// event that would leave the window abandons the call (SW_BAIL_LEVEL -> heap emulation).
// Round 6: the ids of the chunks of finished levels wait for reuse in LDS (they were a stack in HBM: every opening of a chunk -- a
// few per level -- was a dependent round trip to it); the stack holds SW_RING ids, what does not fit is simply not reused.

// the rest of a push whose atomic add on the level word returned `w` (the slot is taken when the chunk had room)
__device__ __forceinline__ void sweep_push_from(SweepRef s, uint32_t& spare, KH_AS_LDS uint32_t* word, uint32_t slot, uint32_t vox,
                                                uint32_t meta, uint32_t w) {
  const uint32_t CH = 1u << s.shift;
  for (;; w = SW_L_ADD(word, 1u)) {
    const uint32_t fill = w & SW_FILL_MASK;
    if (fill < CH) {
      s.chunks[((size_t)(w >> SW_FILL_BITS) << s.shift) + fill] = u32x2_t{vox, meta};
      return;
    }
    if (spare == SW_NONE) {
      // a chunk of a level that is done, if there is one (the stack is only filled between the levels, by wave 0, when
      // nobody takes from it), a fresh one otherwise
      const int have = SW_L_ADD(&s.sh->nfree, -1);
      if (have > 0) spare = s.fs[have - 1];
      else { SW_L_ADD(&s.sh->nfree, 1); spare = SW_L_ADD(&s.sh->bump, 1u); }
    }
    const uint32_t id = spare;
    if (id >= s.chcap || id >= SW_NOCHUNK) { sweep_bail(s, SW_BAIL_ARENA); return; }   // the call is abandoned
    bool mine = false;
    uint32_t seen = w + 1u;
    for (;;) {
      if ((seen & SW_FILL_MASK) < CH) break;                      // somebody installed a chunk: take a slot of it
      uint32_t old = seen;
      __hip_atomic_compare_exchange_strong(word, &old, (id << SW_FILL_BITS) | 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (old == seen) { mine = true; break; }
      seen = old;
    }
    if (!mine) continue;
    spare = SW_NONE;
    const uint32_t prev = seen >> SW_FILL_BITS;
    if (prev == SW_NOCHUNK) SW_L_OR(&s.lvbits[slot >> 5], 1u << (slot & 31u));
    KH_AS_GLOBAL u32x2_t* c = s.chunks + ((size_t)id << s.shift);
    c[0] = u32x2_t{prev, 0u};
    c[1] = u32x2_t{vox, meta};
    return;
  }
}
__device__ __forceinline__ void sweep_push(SweepRef s, uint32_t& spare, uint32_t cur, uint32_t lv, uint32_t vox, uint32_t meta) {
  if (lv - cur > s.wmask) { sweep_bail(s, SW_BAIL_LEVEL); return; }      // (wmask = all ones: never)
  const uint32_t slot = lv & s.wmask;
  KH_AS_LDS uint32_t* word = &s.words[slot];
  sweep_push_from(s, spare, word, slot, vox, meta, SW_L_ADD(word, 1u));
}
// The same event (meta) to the neighbours of v named by `push`, neighbour k at level rk[k].  A dying voxel hands deadlines to
// several neighbours; one after the other that was, per push, a reload of the rank (a run-time index into the rank registers would
// put them in scratch memory), an LDS atomic and a store -- a chain per lane, and the wave runs to its widest lane.  Here the slots
// of 13 pushes are taken first (13 LDS atomics in flight together, the levels picked by constant indices), then the stores go out.
// A push that finds its chunk full (a few per cent: the first event of a level opens one) is left for sweep_push_rest.
template <int K0>
__device__ __forceinline__ uint32_t sweep_push13(SweepRef s, uint32_t cur, uint32_t v, uint32_t push, const uint32_t (&rk)[26],
                                                 uint32_t meta) {
  const uint32_t wmask = s.wmask, CH = 1u << s.shift;
  const int shift = s.shift;
  KH_AS_LDS uint32_t* words = s.words;
  KH_AS_GLOBAL u32x2_t* chunks = s.chunks;
  const int sx = s.g->sx, sxy = s.g->sxy;
  uint32_t w[13];
  uint32_t slow = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    w[j] = 0u;
    if ((push >> (K0 + j)) & 1u) {
      if (rk[K0 + j] - cur > wmask) slow |= 1u << (K0 + j);                 // beyond the window: sweep_push abandons the call
      else w[j] = SW_L_ADD(&words[rk[K0 + j] & wmask], 1u);
    }
  }
  push &= ~slow;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    if (!((push >> (K0 + j)) & 1u)) continue;
    int dx, dy, dz;
    dir_delta(K0 + j, dx, dy, dz);
    const uint32_t fill = w[j] & SW_FILL_MASK;
    if (fill < CH) chunks[((size_t)(w[j] >> SW_FILL_BITS) << shift) + fill] = u32x2_t{v + (uint32_t)(dx + sx * dy + sxy * dz), meta};
    else slow |= 1u << (K0 + j);
  }
  return slow;
}
// the pushes sweep_push13 left: one at a time through the general path (their level is evaluated again: `rk` must not be indexed
// by a run-time value)
__device__ __forceinline__ void sweep_push_rest(SweepRef s, const SweepKeys& K, const u32x4_t src, int x, int y, int z, uint32_t& spare,
                                                uint32_t cur, uint32_t v, uint32_t slow, uint32_t meta) {
  for (uint32_t m = slow; m; m &= m - 1u) {
    const int k = __ffs((int)m) - 1;
    int dx, dy, dz;
    sweep_dir(k, dx, dy, dz);
    uint32_t lv = 0;
    (void)sweep_eval(K, src, x + dx - (int)src.x, y + dy - (int)src.y, z + dz - (int)src.z, lv);
    sweep_push(s, spare, cur, lv, v + (uint32_t)s.g->off[k], meta);
  }
}
__device__ __forceinline__ void sweep_push26(SweepRef s, const SweepKeys& K, const u32x4_t src, int x, int y, int z, uint32_t& spare,
                                             uint32_t cur, uint32_t v, uint32_t push, const uint32_t (&rk)[26], uint32_t meta) {
  uint32_t slow = 0;
  if (push & 0x1FFFu) slow |= sweep_push13<0>(s, cur, v, push, rk, meta);
  if (push >> 13) slow |= sweep_push13<13>(s, cur, v, push, rk, meta);
  if (slow) sweep_push_rest(s, K, src, x, y, z, spare, cur, v, slow, meta);
}

__device__ __forceinline__ void sweep_coords(SweepRef s, uint32_t v, int& x, int& y, int& z) {
  const uint32_t sx = (uint32_t)s.g->sx, sxy = (uint32_t)s.g->sxy;
  const uint32_t zz = v / sxy, r = v - zz * sxy, yy = r / sx;
  z = (int)zz; y = (int)yy; x = (int)(r - yy * sx);
}

// The neighbours the flood can reach from v.  With a voxel_connectivity_graph the reference gates a corner entry that degenerated
// into a yz diagonal at an x face by the CORNER's bit (dijkstra_invalidation.hpp:116-123, 182-190), so that diagonal can be
// entered although its own bit is clear: the sweep follows it too (the heap emulation reads the gate bytes for the same reason).
__device__ __forceinline__ uint32_t sweep_mask(SweepRef s, uint32_t v, uint32_t nm) {
  if (s.gate == nullptr) return nm;
  const uint32_t g = s.gate[v];
  const uint32_t x = v % (uint32_t)s.g->sx;
  uint32_t extra = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    int dx, dy, dz;
    dir_delta(18 + j, dx, dy, dz);
    const int kd = 10 + (dy > 0 ? 2 : 0) + (dz > 0 ? 1 : 0);
    const bool face = dx < 0 ? x == s.xmin : x == s.xmax;
    extra |= (uint32_t)(((g >> j) & 1u) && face) << kd;
  }
  return nm | extra;
}

// ---- the neighbourhood of an event in ONE round trip (round 6).  An event looks at its 26 neighbours: are they alive, which
// deadline is pending for them.  Those were 26 byte loads (alive) + 26 word loads (filter words) + 26 rank loads per event and
// phase, and a gather costs the CU's address unit a cycle or two per LANE: measured (KH_SWEEP_PROBE) 3.3-5 k cycles for the alive
// bytes, 7-9 k for keys and ranks, 5-6 k for the filter words of ONE deadline event.  Now the filter word says "dead" itself
// (SW_SCHED_DEAD, written when a kill is committed), the 27 words around v are read as nine rows of three consecutive words
// (global_load_dwordx3; plain loads: the words are only ever written by plain stores of this workgroup, i.e. through this CU's own
// vector cache -- a first version read them with the nt hint, which keeps them out of the L2 as well: 512 GB per volume against
// 447 in round 5, hit rate 0.56), their addresses depend on nothing but v, so they travel with the event's own words; and in
// integer mode the levels are arithmetic.
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef u32x3_t u32x3_a4_t __attribute__((aligned(4)));
// w[(dx + 1) + 3 * ((dy + 1) + 3 * (dz + 1))] = filter word of the voxel at (x + dx, y + dy, z + dz); rows outside the volume
// read v's own row (their neighbours are not in anybody's mask)
__device__ __forceinline__ void sweep_rows(const KH_AS_GLOBAL uint32_t* sched, int sx, int sxy, int sy, int sz, uint32_t v, int y, int z,
                                           uint32_t (&w)[27]) {
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const int dy = r % 3 - 1, dz = r / 3 - 1;
    const bool in = (unsigned)(y + dy) < (unsigned)sy && (unsigned)(z + dz) < (unsigned)sz;
    const long long base = (long long)v + (in ? dy * sx + dz * sxy : 0) - 1;      // (-1 for voxel 0: the word in front of the volume)
    const u32x3_t t = *(const KH_AS_GLOBAL u32x3_a4_t*)(sched + base);
    w[3 * r + 0] = t.x; w[3 * r + 1] = t.y; w[3 * r + 2] = t.z;
  }
}
constexpr int sweep_widx(int k) {
  int dx = 0, dy = 0, dz = 0;
  dir_delta(k, dx, dy, dz);
  return (dx + 1) + 3 * ((dy + 1) + 3 * (dz + 1));
}
// the neighbours in `nm` that are alive
__device__ __forceinline__ uint32_t sweep_alive26(const uint32_t (&w)[27], uint32_t nm) {
  uint32_t am = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) am |= (uint32_t)(w[sweep_widx(k)] != SW_SCHED_DEAD) << k;
  return am & nm;
}
// table mode (any anisotropy): the float keys and a gather of 26 ranks; out of line, the levels come back through memory
__device__ __attribute__((noinline)) uint32_t sweep_keys26_table(const SweepKeys& K, const u32x4_t src, int ex0, int ey0, int ez0, uint32_t am,
                                                                uint32_t (&rk)[26]) {
  const float r = __uint_as_float(src.w);
  uint32_t cov = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) {
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    const int ex = ex0 + dx, ey = ey0 + dy, ez = ez0 + dz;
    const float d = sweep_fkey(K, ex, ey, ez);
    const bool in = ((am >> k) & 1u) && d < r;
    const int ax = ex < 0 ? -ex : ex, ay = ey < 0 ? -ey : ey, az = ez < 0 ? -ez : ez;
    rk[k] = K.rank[in ? ax + K.ra * (ay + K.rb * az) : 0];
    cov |= (uint32_t)in << k;
  }
  return cov;
}
// neighbours of (x, y, z) that are in `am` and inside the ball of src: coverage mask (bit k) + the levels of their keys
__device__ __forceinline__ uint32_t sweep_keys26(const SweepKeys& K, const u32x4_t src, int x, int y, int z, uint32_t am,
                                                 uint32_t (&rk)[26]) {
  const int ex0 = x - (int)src.x, ey0 = y - (int)src.y, ez0 = z - (int)src.z;
  uint32_t cov = 0;
  if (K.gq) {
    uint32_t X[3], Y[3], Z[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      X[d] = K.gx * (uint32_t)((ex0 + d - 1) * (ex0 + d - 1));
      Y[d] = K.gy * (uint32_t)((ey0 + d - 1) * (ey0 + d - 1));
      Z[d] = K.gz * (uint32_t)((ez0 + d - 1) * (ez0 + d - 1));
    }
#pragma unroll
    for (int k = 0; k < 26; k++) {
      int dx, dy, dz;
      dir_delta(k, dx, dy, dz);
      const uint32_t S = X[dx + 1] + Y[dy + 1] + Z[dz + 1];
      rk[k] = S;
      cov |= (uint32_t)(S < src.w) << k;
    }
    return cov & am;
  }
  uint32_t tmp[26];            // (an array handed to an out-of-line function lives in memory: keep that away from rk)
  cov = sweep_keys26_table(K, src, ex0, ey0, ez0, am, tmp);
#pragma unroll
  for (int k = 0; k < 26; k++) rk[k] = tmp[k];
  return cov;
}
// covered neighbours whose level lies above `lvl` (the levels are looked at with constant indices only: a level picked by a
// run-time index turns the array into a scratch-memory table -- 26 stores per event and a dependent load per pick)
__device__ __forceinline__ uint32_t sweep_above(const uint32_t (&rk)[26], uint32_t cov, uint32_t lvl) {
  uint32_t up = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) up |= (uint32_t)(rk[k] > lvl) << k;
  return up & cov;
}
// ---- pending-deadline filter.  An event that carries a deadline (D or PD) for voxel q at level t makes every event of q
// at a later level a no-op (q is dead once level t is complete: sweep_possible and sweep_deadline return at their alive
// test), and a second identical event is a no-op as well (the first adds the candidate / sets the dying bit, the second
// returns at its first test).  A voxel is handed such events by every neighbour that dies before it does -- about a dozen
// per voxel, nearly all of them the voxel's own key from the same source -- so unfiltered the sweep stores, reloads and
// dismisses ~13 events per voxel.  sched[q] holds the value (level << cb | code) of a deadline-carrying event that WAS pushed to q
// (code = source + 1 for PD, 0 for a pure D; cb = the bits the codes of THIS call need, so a call with few sources -- a soma's
// single root, the plates of the reference's tests -- can have millions of levels).  A new event is pushed iff its value lies below
// the word (an earlier deadline) or ties its level under another code (two sources on the same key: the tie the certificate is
// about).  Skipped are only exact duplicates and events at a level above a pending deadline: the machine's states, bails and
// result are those of the unfiltered sweep.
// Round 6: no read-modify-write on these words.  Every global atomic of gfx950 is executed on the memory side of the fabric and
// drops its line from the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"; the ISA has no scope bit on an atomic: agent
// and workgroup scope compile to the same instruction).  The filter does not need the minimum: it needs the word to hold the value
// of SOME deadline-carrying event that was pushed for the voxel.  A skipped event is then still an exact duplicate or lies behind
// a pending deadline; a word that is larger than the true minimum (two lanes stored at once and the larger value landed) only lets
// a few more no-op events through.  So: look (the word is in the rows the event has read anyway) and store when lower.
// A live voxel reads SW_SCHED_NONE at the start of every call: a word that was lowered belongs to a voxel that dies in the call
// (its deadline is processed) or the call is abandoned and the label's words are reset.
struct SweepFilter {
  KH_AS_GLOBAL uint32_t* sched;
  int cb;            // bits of the source code
};
// the filter of a call with npath sources on a label with nlev levels; sched == nullptr: (level << cb | code) does not fit 32 bits
__device__ __forceinline__ SweepFilter sweep_filter(SweepRef s, uint32_t npath) {
  SweepFilter f;
  f.cb = 32 - __clz((int)npath);                 // codes 1 .. npath
  f.sched = (s.sched != nullptr && s.nlev < (0xFFFFFFFFu >> f.cb)) ? s.sched : nullptr;
  return f;
}
__device__ __forceinline__ bool sweep_claim(const SweepFilter f, uint32_t q, uint32_t tr, uint32_t code) {
  const uint32_t val = (tr << f.cb) | code;
  const uint32_t old = sweep_ld(&f.sched[q]);
  if (val < old) f.sched[q] = val;
  return val < old || ((old >> f.cb) == tr && old != val);
}
// a pure P event of q at level tr is a no-op when a deadline of q is pending at an earlier level
__device__ __forceinline__ bool sweep_moot(const SweepFilter f, uint32_t q, uint32_t tr) {
  return (sweep_ld(&f.sched[q]) >> f.cb) < tr;
}
// the same for all neighbours of v at once, from the words the event has read (`want`: bit k = neighbour k gets a PD event of source
// code - 1 at level rk[k]).  Returns the events to push.
__device__ __forceinline__ uint32_t sweep_claim26(KH_AS_GLOBAL uint32_t* sched, int cb, int sx, int sxy, uint32_t v, uint32_t want,
                                                  const uint32_t (&rk)[26], const uint32_t (&w)[27], uint32_t code) {
  uint32_t keep = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) {
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    const uint32_t old = w[sweep_widx(k)];
    const uint32_t val = (rk[k] << cb) | code;
    const bool lower = ((want >> k) & 1u) && val < old;
    if (lower) sched[v + (uint32_t)(dx + sx * dy + sxy * dz)] = val;
    keep |= (uint32_t)(lower || ((old >> cb) == rk[k] && old != val)) << k;
  }
  return keep & want;
}

// ---- candidate spill.  Four possible owners per voxel are the rule; a voxel near the bisector planes of several path vertices
// (or deep inside overlapping balls of a thick object) can have more, and without their identities no deadline can be derived
// from its death.  Owners five to eight go to a small label-private hash table (linear probing on voxel + 1); the voxel's own word
// gets the SW_SPILL flag once the entry is complete.  Candidates are only added in the candidate phase of a level and only read
// in its deadline phase, with workgroup barriers between the two, so a reader always sees a finished entry.
// Returns 0 = c was a candidate already, 1 = added, 2 = no room (table full or a ninth candidate).
__device__ __attribute__((noinline)) int sweep_spill_add(SweepRef s, uint32_t v, uint32_t c) {
  if (s.spcap == 0u) return 2;
  const uint32_t mask = s.spcap - 1u;
  uint32_t h = (v * 0x9E3779B1u) >> 7 & mask;
  for (uint32_t probes = 0;; probes++, h = (h + 1u) & mask) {
    if (probes == s.spcap) return 2;
    const uint32_t k = sw_g_cas(&s.spk[h], 0u, v + 1u);
    if (k == 0u) { SW_L_ADD(&s.sh->nspill, 1u); break; }
    if (k == v + 1u) break;
  }
  unsigned long long cs = __hip_atomic_load(&s.spc[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    int freeslot = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--) {
      const uint32_t sl = (uint32_t)(cs >> (16 * i)) & 0x7fffu;
      if (sl == c + 1u) return 0;
      if (sl == 0u) freeslot = i;
    }
    if (freeslot < 0) return 2;
    const unsigned long long old = sw_g_cas(&s.spc[h], cs, cs | ((unsigned long long)(c + 1u) << (16 * freeslot)));
    if (old == cs) return 1;
    cs = old;
  }
}
// the spilled candidates of v (its own word carries SW_SPILL)
__device__ __attribute__((noinline)) unsigned long long sweep_spill_get(SweepRef s, uint32_t v) {
  const uint32_t mask = s.spcap - 1u;
  uint32_t h = (v * 0x9E3779B1u) >> 7 & mask;
  for (uint32_t probes = 0; probes < s.spcap; probes++, h = (h + 1u) & mask) {
    const uint32_t k = sweep_ld(&s.spk[h]);
    if (k == v + 1u) return __hip_atomic_load(&s.spc[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == 0u) break;
  }
  return 0ull;
}

// a P event (c may own v from this level on)
// General form (table mode; out of line): the levels of the 26 neighbours are kept in an array.
__device__ __attribute__((noinline)) void sweep_possible_gen(SweepRef s, const SweepFilter flt, uint32_t lvl, uint32_t v, uint32_t c, bool has_deadline) {
  // one round trip: the voxel's own words, the source record, the 27 filter words around it
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  unsigned long long cs = s.cstate[v];
  const uint32_t nm0 = s.nbrmask[v];
  const u32x4_t src = s.srcs[c];
  uint32_t w[27];
  sweep_rows(flt.sched, s.g->sx, s.g->sxy, s.g->sy, s.g->sz, v, y, z, w);
  if (w[13] == SW_SCHED_DEAD) return;
  const uint32_t am = sweep_alive26(w, sweep_mask(s, v, nm0));
  unsigned long long want;
  for (;;) {
    int freeslot = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--) {
      const uint32_t sl = (uint32_t)(cs >> (16 * i)) & 0x7fffu;
      if (sl == c + 1u) return;                  // already a candidate
      if (sl == 0u) freeslot = i;
    }
    if (freeslot < 0) {
      // a fifth-plus possible owner: the spill table (rare)
      const int r = sweep_spill_add(s, v, c);
      if (r == 2) { sweep_bail(s, SW_BAIL_CAND); return; }
      if (r == 0) return;
      if (!(cs & SW_SPILL)) SW_G_OR(&s.cstate[v], SW_SPILL);
      break;
    }
    want = cs | ((unsigned long long)(c + 1u) << (16 * freeslot));
    const unsigned long long old = sw_g_cas(&s.cstate[v], cs, want);
    if (old == cs) break;
    cs = old;
  }
  if ((cs & ~SW_DYING) == 0ull) SW_L_ADD(&s.sh->nM, 1);
  if (!has_deadline) {
    const uint32_t p = SW_L_ADD(&s.sh->nnp, 1u);
    if (p < s.ncap) s.np[p] = ((unsigned long long)v << 32) | c; else sweep_bail(s, SW_BAIL_LIST);
  }
  // cascade: neighbours whose key from c is not above this level may be owned by c inside this level
  if (!am) return;
  uint32_t lm;
  {
    const SweepKeys K = sweep_keys(s);
    uint32_t rk[26];
    const uint32_t cov = sweep_keys26(K, src, x, y, z, am, rk);
    lm = cov & ~sweep_above(rk, cov, lvl);
  }
  for (uint32_t m = lm; m; m &= m - 1u) {
    const int k = __ffs((int)m) - 1;
    const uint32_t q = v + (uint32_t)s.g->off[k];
    // (a cheap look keeps most duplicates out of the list; the real test is the CAS of the entry's own turn)
    const unsigned long long qs = s.cstate[q];
    bool have = false;
#pragma unroll
    for (int i = 0; i < 4; i++) have = have || ((uint32_t)(qs >> (16 * i)) & 0x7fffu) == c + 1u;
    if (have) continue;
    const uint32_t p = SW_L_ADD(&s.sh->na, 1u);
    if (p < s.ncap) s.wa[p] = ((unsigned long long)q << 32) | c; else sweep_bail(s, SW_BAIL_LIST);
  }
}

// P events of the pair (v, c) to the levels above lvl (the pair was added by a pure P event and v survives the level, or v is a
// dying ghost with the single possible owner c)
__device__ __attribute__((noinline)) void sweep_emit_possible(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                                              uint32_t c) {
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  const uint32_t nm0 = s.nbrmask[v];
  const u32x4_t src = s.srcs[c];
  uint32_t w[27];
  sweep_rows(flt.sched, s.g->sx, s.g->sxy, s.g->sy, s.g->sz, v, y, z, w);
  const uint32_t am = sweep_alive26(w, sweep_mask(s, v, nm0));
  if (!am) return;
  const SweepKeys K = sweep_keys(s);
  uint32_t rk[26];
  const uint32_t cov = sweep_keys26(K, src, x, y, z, am, rk);
  // a pure P event is moot when a deadline is pending for the neighbour at an earlier level (the word is in the rows)
  uint32_t push = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) push |= (uint32_t)(rk[k] > lvl && !((w[sweep_widx(k)] >> flt.cb) < rk[k])) << k;
  sweep_push26(s, K, src, x, y, z, spare, lvl, v, push & cov, rk, c | SW_P);
}

// the neighbours of a dying voxel with a single candidate source: covered ones die with it (deadline at their own key)
__device__ __forceinline__ void sweep_deadline_one(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                                   uint32_t cid, const u32x4_t src, int x, int y, int z, uint32_t am,
                                                   const uint32_t (&w)[27]) {
  const SweepKeys K = sweep_keys(s);
  uint32_t rk[26];
  const uint32_t cov = sweep_keys26(K, src, x, y, z, am, rk);
  const uint32_t up = sweep_above(rk, cov, lvl);   // covered neighbours whose own key lies above this level: a PD event there
  SW_DT(1, up);
  const uint32_t push = sweep_claim26(flt.sched, flt.cb, s.g->sx, s.g->sxy, v, up, rk, w, cid + 1u);
  SW_DT(2, push);
  for (uint32_t m = cov & ~up; m; m &= m - 1u) {      // same level: the cascade of this level
    const uint32_t q = v + (uint32_t)s.g->off[__ffs((int)m) - 1];
    if (s.cstate[q] & SW_DYING) continue;
    const uint32_t p = SW_L_ADD(&s.sh->nb, 1u);
    if (p < s.ncap) s.wb[p] = q; else sweep_bail(s, SW_BAIL_LIST);
  }
  SW_DT(3, push);
  if (push) sweep_push26(s, K, src, x, y, z, spare, lvl, v, push, rk, cid | SW_P | SW_D);
  SW_DT(4, spare);
}

// the neighbours of a dying voxel with up to eight candidate sources (w0: its own word, w1: the spilled ones) or of a dying
// ghost -- the rare cases, written as plain loops.  A neighbour dies with v when ALL candidates cover it (deadline at the largest
// of their keys); every candidate's possible node of a later level goes out on its own.  A ghost may have been dead all along:
// it emits its possible nodes but no deadlines.
__device__ __attribute__((noinline)) void sweep_deadline_many(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl,
                                                             uint32_t v, unsigned long long w0, unsigned long long w1, uint32_t am,
                                                             bool ghost) {
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  const SweepKeys K = sweep_keys(s);
  for (uint32_t m = am; m; m &= m - 1u) {
    const int k = __ffs((int)m) - 1;
    const uint32_t q = v + (uint32_t)s.g->off[k];
    int dx, dy, dz;
    sweep_dir(k, dx, dy, dz);
    bool all = true;
    uint32_t tr = 0;
    for (int i = 0; i < 8; i++) {
      const uint32_t sl = (uint32_t)((i < 4 ? w0 : w1) >> (16 * (i & 3))) & 0x7fffu;
      if (sl == 0u) continue;
      uint32_t rk = 0;
      const u32x4_t src = s.srcs[sl - 1u];
      const bool cov = sweep_eval(K, src, x + dx - (int)src.x, y + dy - (int)src.y, z + dz - (int)src.z, rk);
      all = all && cov;
      if (cov && rk > tr) tr = rk;
      if (cov && rk > lvl && !sweep_moot(flt, q, rk)) sweep_push(s, spare, lvl, rk, q, (sl - 1u) | SW_P);
    }
    if (!all || ghost) continue;
    if (tr <= lvl) {
      if (s.cstate[q] & SW_DYING) continue;
      const uint32_t p = SW_L_ADD(&s.sh->nb, 1u);
      if (p < s.ncap) s.wb[p] = q; else sweep_bail(s, SW_BAIL_LIST);
    } else if (sweep_claim(flt, q, tr, 0u)) {
      sweep_push(s, spare, lvl, tr, q, SW_D);
    }
  }
}

// the rare kinds of a dying voxel (out of line): more than four possible owners (spill table), a ghost, two to four possible owners.
// `old` = the voxel's candidate word before the dying bit, `am` = its alive neighbours.
__device__ __attribute__((noinline)) void sweep_deadline_rest(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                                              unsigned long long old, uint32_t am, bool ghost) {
  uint32_t cid[4];
  bool has[4];
  int nc = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t sl = (uint32_t)(old >> (16 * i)) & 0x7fffu;
    has[i] = sl != 0u;
    cid[i] = sl - 1u;
    nc += has[i] ? 1 : 0;
  }
  if ((old & SW_SPILL) || (ghost && nc > 1)) {
    sweep_deadline_many(s, flt, spare, lvl, v, old, (old & SW_SPILL) ? sweep_spill_get(s, v) : 0ull, am, ghost);
    return;
  }
  if (nc == 1) {                      // a ghost with one possible owner: its possible nodes, no deadlines
    uint32_t c1 = cid[0];
#pragma unroll
    for (int i = 1; i < 4; i++) if (has[i]) c1 = cid[i];
    sweep_emit_possible(s, flt, spare, lvl, v, c1);
    return;
  }
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  const SweepKeys K = sweep_keys(s);
  u32x4_t src[4];
#pragma unroll
  for (int i = 0; i < 4; i++) src[i] = s.srcs[has[i] ? cid[i] : 0u];
  for (uint32_t m = am; m; m &= m - 1u) {
    const int k = __ffs((int)m) - 1;
    const uint32_t q = v + (uint32_t)s.g->off[k];
    int dx, dy, dz;
    sweep_dir(k, dx, dy, dz);
    bool all = true;
    uint32_t tr = 0, rk[4];
    bool cov[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      cov[i] = false;
      rk[i] = 0u;
      if (has[i]) {
        cov[i] = sweep_eval(K, src[i], x + dx - (int)src[i].x, y + dy - (int)src[i].y, z + dz - (int)src[i].z, rk[i]);
        all = all && cov[i];
        if (cov[i] && rk[i] > tr) tr = rk[i];
      }
    }
    if (all) {
      if (tr <= lvl) {
        if (s.cstate[q] & SW_DYING) continue;
        const uint32_t p = SW_L_ADD(&s.sh->nb, 1u);
        if (p < s.ncap) s.wb[p] = q; else sweep_bail(s, SW_BAIL_LIST);
      } else if (sweep_claim(flt, q, tr, 0u)) {
        sweep_push(s, spare, lvl, tr, q, SW_D);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (cov[i] && rk[i] > lvl && !sweep_moot(flt, q, rk[i])) sweep_push(s, spare, lvl, rk[i], q, cid[i] | SW_P);
  }
}

// a D event: v is dead once this level is complete
// `hint`: the source of the event when it is a PD event (the usual single candidate of its voxel), SW_NONE otherwise.
// The one round trip carries everything that does not depend on anything else: the voxel's alive byte and neighbour mask, the
// hinted source record, the 27 filter words around it, and the dying bit itself -- set before the alive byte is known and taken
// back when the voxel turns out to be dead (a dead voxel's word is zero and nobody looks at it: dead voxels are not in anybody's
// neighbour set).
__device__ __attribute__((noinline)) void sweep_deadline_gen(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                                             uint32_t hint) {
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  SW_D0();
  const uint8_t live = s.alive[v];
  const uint32_t nm0 = s.nbrmask[v];
  const u32x4_t hsrc = s.srcs[hint != SW_NONE ? hint : 0u];
  const unsigned long long old = SW_G_OR(&s.cstate[v], SW_DYING);
  uint32_t w[27];
  sweep_rows(flt.sched, s.g->sx, s.g->sxy, s.g->sy, s.g->sz, v, y, z, w);
  SW_DT(0, (uint32_t)old + live + nm0 + hsrc.x + w[0] + w[26]);
  const uint32_t am = sweep_alive26(w, sweep_mask(s, v, nm0));
  if (!live) {
    if (!(old & SW_DYING)) SW_G_AND(&s.cstate[v], ~SW_DYING);
    return;
  }
  if (old & SW_DYING) return;
  if (old == 0ull) { sweep_bail(s, SW_BAIL_UNTOUCHED); return; }
  s.killed[SW_L_ADD(&s.sh->nkill, 1u)] = v;
  const bool ghost = live == SW_GHOST;
  if (ghost) SW_L_ADD(&s.sh->nkg, 1u);
  // the candidate slots stay where they are in the word (no compaction: every array below is indexed by constants only,
  // so nothing of this lives in scratch memory)
  uint32_t cid[4];
  bool has[4];
  int nc = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t sl = (uint32_t)(old >> (16 * i)) & 0x7fffu;
    has[i] = sl != 0u;
    cid[i] = sl - 1u;
    nc += has[i] ? 1 : 0;
  }
  if (!am) return;
  if ((old & SW_SPILL) || ghost || nc != 1) { sweep_deadline_rest(s, flt, spare, lvl, v, old, am, ghost); return; }
  uint32_t c1 = cid[0];
#pragma unroll
  for (int i = 1; i < 4; i++) if (has[i]) c1 = cid[i];
  const u32x4_t s1 = c1 == hint ? hsrc : s.srcs[c1];       // (the hinted record is here already)
  sweep_deadline_one(s, flt, spare, lvl, v, c1, s1, x, y, z, am, w);
}

// ---- integer mode, the form the path loop runs (each handler expanded in ONE place: the loops of sweep_ball take a level's own events
// and its cascade through the same call; everything rare is out of line -- with all of it inlined everywhere the kernel's code had
// grown to 570 KB, nine times the instruction cache, and the heap emulation's chain, which shares that cache, ran 8 % slower).
// The level of neighbour (dx, dy, dz) is X[dx + 1] + Y[dy + 1] + Z[dz + 1]: nine registers instead of 26.
struct SweepLv { uint32_t X[3], Y[3], Z[3]; };
__device__ __forceinline__ SweepLv sweep_lv(uint32_t gx, uint32_t gy, uint32_t gz, const u32x4_t src, int x, int y, int z) {
  const int ex0 = x - (int)src.x, ey0 = y - (int)src.y, ez0 = z - (int)src.z;
  SweepLv L;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    L.X[d] = gx * (uint32_t)((ex0 + d - 1) * (ex0 + d - 1));
    L.Y[d] = gy * (uint32_t)((ey0 + d - 1) * (ey0 + d - 1));
    L.Z[d] = gz * (uint32_t)((ez0 + d - 1) * (ez0 + d - 1));
  }
  return L;
}
#define SW_LV(L, k) ([&]() { int dx_ = 0, dy_ = 0, dz_ = 0; dir_delta(k, dx_, dy_, dz_); return (L).X[dx_ + 1] + (L).Y[dy_ + 1] + (L).Z[dz_ + 1]; }())

__device__ __forceinline__ void sweep_possible_int(SweepRef s, const SweepFilter flt, uint32_t lvl, uint32_t v, uint32_t c, bool has_deadline) {
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  unsigned long long cs = s.cstate[v];
  const uint32_t nm0 = s.nbrmask[v];
  const u32x4_t src = s.srcs[c];
  uint32_t w[27];
  sweep_rows(flt.sched, s.g->sx, s.g->sxy, s.g->sy, s.g->sz, v, y, z, w);
  if (w[13] == SW_SCHED_DEAD) return;
  const uint32_t am = sweep_alive26(w, sweep_mask(s, v, nm0));
  unsigned long long want;
  for (;;) {
    int freeslot = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--) {
      const uint32_t sl = (uint32_t)(cs >> (16 * i)) & 0x7fffu;
      if (sl == c + 1u) return;                  // already a candidate
      if (sl == 0u) freeslot = i;
    }
    if (freeslot < 0) {
      const int r = sweep_spill_add(s, v, c);    // a fifth-plus possible owner: the spill table (rare)
      if (r == 2) { sweep_bail(s, SW_BAIL_CAND); return; }
      if (r == 0) return;
      if (!(cs & SW_SPILL)) SW_G_OR(&s.cstate[v], SW_SPILL);
      break;
    }
    want = cs | ((unsigned long long)(c + 1u) << (16 * freeslot));
    const unsigned long long old = sw_g_cas(&s.cstate[v], cs, want);
    if (old == cs) break;
    cs = old;
  }
  if ((cs & ~SW_DYING) == 0ull) SW_L_ADD(&s.sh->nM, 1);
  if (!has_deadline) {
    const uint32_t p = SW_L_ADD(&s.sh->nnp, 1u);
    if (p < s.ncap) s.np[p] = ((unsigned long long)v << 32) | c; else sweep_bail(s, SW_BAIL_LIST);
  }
  if (!am) return;
  // cascade: neighbours whose key from c is not above this level may be owned by c inside this level
  uint32_t lm = 0;
  {
    const SweepLv L = sweep_lv(s.gx, s.gy, s.gz, src, x, y, z);
#pragma unroll
    for (int k = 0; k < 26; k++) { const uint32_t S = SW_LV(L, k); lm |= (uint32_t)(S < src.w && S <= lvl) << k; }
  }
  for (uint32_t m = lm & am; m; m &= m - 1u) {
    const int k = __ffs((int)m) - 1;
    const uint32_t q = v + (uint32_t)s.g->off[k];
    const unsigned long long qs = s.cstate[q];
    bool have = false;
#pragma unroll
    for (int i = 0; i < 4; i++) have = have || ((uint32_t)(qs >> (16 * i)) & 0x7fffu) == c + 1u;
    if (have) continue;
    const uint32_t p = SW_L_ADD(&s.sh->na, 1u);
    if (p < s.ncap) s.wa[p] = ((unsigned long long)q << 32) | c; else sweep_bail(s, SW_BAIL_LIST);
  }
}

// thirteen of a dying voxel's pushes: the slots first, then the stores; returns the pushes that found their chunk full
template <int K0>
__device__ __forceinline__ uint32_t sweep_push13_int(SweepRef s, const SweepLv& L, uint32_t cur, uint32_t v, uint32_t push, uint32_t meta) {
  const uint32_t wmask = s.wmask, CH = 1u << s.shift;
  const int shift = s.shift;
  KH_AS_LDS uint32_t* words = s.words;
  KH_AS_GLOBAL u32x2_t* chunks = s.chunks;
  const int sx = s.g->sx, sxy = s.g->sxy;
  uint32_t w[13];
  uint32_t slow = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    w[j] = 0u;
    if ((push >> (K0 + j)) & 1u) {
      const uint32_t S = SW_LV(L, K0 + j);
      if (S - cur > wmask) slow |= 1u << (K0 + j);                          // beyond the window: sweep_push abandons the call
      else w[j] = SW_L_ADD(&words[S & wmask], 1u);
    }
  }
  push &= ~slow;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    if (!((push >> (K0 + j)) & 1u)) continue;
    int dx, dy, dz;
    dir_delta(K0 + j, dx, dy, dz);
    const uint32_t fill = w[j] & SW_FILL_MASK;
    if (fill < CH) chunks[((size_t)(w[j] >> SW_FILL_BITS) << shift) + fill] = u32x2_t{v + (uint32_t)(dx + sx * dy + sxy * dz), meta};
    else slow |= 1u << (K0 + j);
  }
  return slow;
}

__device__ __forceinline__ void sweep_deadline_int(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                                   uint32_t hint) {
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  SW_D0();
  const uint8_t live = s.alive[v];
  const uint32_t nm0 = s.nbrmask[v];
  const u32x4_t hsrc = s.srcs[hint != SW_NONE ? hint : 0u];
  const unsigned long long old = SW_G_OR(&s.cstate[v], SW_DYING);
  uint32_t w[27];
  sweep_rows(flt.sched, s.g->sx, s.g->sxy, s.g->sy, s.g->sz, v, y, z, w);
  SW_DT(0, (uint32_t)old + live + nm0 + hsrc.x + w[0] + w[26]);
  const uint32_t am = sweep_alive26(w, sweep_mask(s, v, nm0));
  if (!live) {
    if (!(old & SW_DYING)) SW_G_AND(&s.cstate[v], ~SW_DYING);
    return;
  }
  if (old & SW_DYING) return;
  if (old == 0ull) { sweep_bail(s, SW_BAIL_UNTOUCHED); return; }
  s.killed[SW_L_ADD(&s.sh->nkill, 1u)] = v;
  const bool ghost = live == SW_GHOST;
  if (ghost) SW_L_ADD(&s.sh->nkg, 1u);
  if (!am) return;
  // one possible owner, no ghost: the common case; everything else out of line
  uint32_t c1 = 0;
  int nc = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t sl = (uint32_t)(old >> (16 * i)) & 0x7fffu;
    if (sl != 0u) { c1 = sl - 1u; nc++; }
  }
  if ((old & SW_SPILL) || ghost || nc != 1) { sweep_deadline_rest(s, flt, spare, lvl, v, old, am, ghost); return; }
  const u32x4_t src = c1 == hint ? hsrc : s.srcs[c1];       // (the hinted record is here already)
  const SweepLv L = sweep_lv(s.gx, s.gy, s.gz, src, x, y, z);
  // covered neighbours die with v: those whose own key lies above this level get a PD event there -- unless the filter word the
  // event has read says an identical or an earlier deadline is pending; the others belong to this level's cascade
  const int cb = flt.cb, sx = s.g->sx, sxy = s.g->sxy;
  const uint32_t code = c1 + 1u;
  uint32_t push = 0, casc = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) {
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    const uint32_t S = SW_LV(L, k);
    const bool in = ((am >> k) & 1u) && S < src.w;
    const bool up = in && S > lvl;
    const uint32_t oldw = w[sweep_widx(k)];
    const uint32_t val = (S << cb) | code;
    const bool lower = up && val < oldw;
    if (lower) flt.sched[v + (uint32_t)(dx + sx * dy + sxy * dz)] = val;
    push |= (uint32_t)(up && (lower || ((oldw >> cb) == S && oldw != val))) << k;
    casc |= (uint32_t)(in && !up) << k;
  }
  SW_DT(2, push);
  for (uint32_t m = casc; m; m &= m - 1u) {      // same level: the cascade of this level
    const uint32_t q = v + (uint32_t)s.g->off[__ffs((int)m) - 1];
    if (s.cstate[q] & SW_DYING) continue;
    const uint32_t p = SW_L_ADD(&s.sh->nb, 1u);
    if (p < s.ncap) s.wb[p] = q; else sweep_bail(s, SW_BAIL_LIST);
  }
  SW_DT(3, push);
  if (push) {
    const uint32_t meta = c1 | SW_P | SW_D;
    uint32_t slow = 0;
    if (push & 0x1FFFu) slow |= sweep_push13_int<0>(s, L, lvl, v, push, meta);
    if (push >> 13) slow |= sweep_push13_int<13>(s, L, lvl, v, push, meta);
    for (uint32_t m = slow; m; m &= m - 1u) {      // a few per cent: the chunk was full (or the level not open yet)
      const int k = __ffs((int)m) - 1;
      int dx, dy, dz;
      sweep_dir(k, dx, dy, dz);
      const int ex = x + dx - (int)src.x, ey = y + dy - (int)src.y, ez = z + dz - (int)src.z;
      const uint32_t S = s.gx * (uint32_t)(ex * ex) + s.gy * (uint32_t)(ey * ey) + s.gz * (uint32_t)(ez * ez);
      sweep_push(s, spare, lvl, S, v + (uint32_t)s.g->off[k], meta);
    }
  }
  SW_DT(4, spare);
}

// event e of the level being processed: the newest chunk (chain[0]) holds `newest` events, the others are full
__device__ __forceinline__ u32x2_t sweep_event(SweepRef s, uint32_t e, uint32_t newest) {
  const uint32_t per = (1u << s.shift) - 1u;
  uint32_t c = 0, slot = e + 1u;
  if (e >= newest) { const uint32_t r = e - newest; c = 1u + r / per; slot = r - (c - 1u) * per + 1u; }
  return s.chunks[((size_t)s.chain[c] << s.shift) + slot];
}

// the filter words of the label's voxels as a call expects them: "none" for the live ones, "dead" for the others
__device__ __forceinline__ void sweep_reset_words(SweepRef s, const uint32_t* list, uint32_t nf) {
  for (uint32_t i = threadIdx.x; i < nf; i += blockDim.x) {
    const uint32_t v = list[i];
    s.sched[v] = s.alive[v] ? SW_SCHED_NONE : SW_SCHED_DEAD;
  }
}

// Whole workgroup.  path / npath: the vertices of the new path; srcs has room for npath records.
// Returns true when certified (alive updated, *count = voxels invalidated); false when the call has to be redone by
// the heap emulation (alive and cstate are as they were on entry).
// allow_ghosts: a call that ends with undecided voxels (state M: alive under some resolutions of the heap's tie order only) is
// not abandoned; those voxels become GHOSTS (alive byte SW_GHOST) and the call counts as certified.  In the calls that follow a
// ghost takes candidates and passes possible nodes on like an alive voxel (it may be one), a deadline kills it for certain, and
// it emits no deadlines itself (it may have been dead all along) -- so everything those calls decide holds under either
// status.  The caller must not let a ghost influence its control flow (it rolls back to this call and redoes
// it by the heap emulation when one would).  The ghosts made are appended to s.killed behind the killed voxels
// (sh->nkill of them; sh->nghost ghosts), *count = killed voxels that were no ghosts, sh->nkg = killed ghosts.
// Out of line ON PURPOSE: as a function of its own its registers are allocated for the sweep alone -- inlined into the path kernel
// it shared them with everything that kernel keeps alive across an invalidation, and whatever did not fit was spilled in the
// sweep's loops and in the heap emulation's (which ran 8 % slower for it).  Returns -1 (redo by the heap) or the count.
__device__ __attribute__((noinline)) long long sweep_ball(SweepRef s, const uint32_t* path, uint32_t npath, const float* __restrict__ dbf,
                                                          float scale, float constant, float rmax, const uint32_t* list, uint32_t nf,
                                                          bool allow_ghosts) {
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  KH_AS_LDS SweepShared* sh = s.sh;
  const SweepFilter flt = sweep_filter(s, npath);
  if (flt.sched == nullptr || npath > s.srcs_cap) {   // no filter words, (level << cb | code) does not fit them, or no room for the sources
    __syncthreads();
    if (tid == 0) sh->bail = SW_BAIL_LEVEL;
    __syncthreads();
    return -1;
  }
  uint32_t spare = SW_NONE;
  const bool ikeys = s.gq != 0u;
  const uint32_t EMPTY = (SW_NOCHUNK << SW_FILL_BITS) | (1u << s.shift);
  const uint32_t nwords = (s.nslots >> 5) + 1u;
  for (uint32_t i = tid; i < s.nslots; i += nthr) s.words[i] = EMPTY;
  for (uint32_t i = tid; i < nwords; i += nthr) s.lvbits[i] = 0u;
  if (tid == 0) {
    sh->na = sh->nb = sh->nnp = sh->bump = sh->nkill = sh->snap = sh->bail = 0u;
    sh->nM = 0;
    sh->nkg = sh->nspill = sh->nghost = 0u;
    sh->nfree = 0;
    sh->nprev = 0u;
    sh->lvl = 0u;
    sh->levels = sh->events = sh->maxnev = 0u;
#ifdef KH_SWEEP_PROBE
    for (int i = 0; i < 8; i++) { sh->cyc[i] = 0; sh->cyd[i] = 0; }
#endif
  }
  __syncthreads();
  KH_AS_GLOBAL u32x4_t* srcs = s.srcs;
  for (uint32_t i = tid; i < npath; i += nthr) {
    const uint32_t v = path[i];
    float r = scale * dbf[v];        // skeletontricks.pyx:393-395: numpy float32 scalar arithmetic
    r = r + constant;
    if (!(r <= rmax)) sweep_bail(s, SW_BAIL_LEVEL);
    int x, y, z;
    sweep_coords(s, v, x, y, z);
    srcs[i] = u32x4_t{(uint32_t)x, (uint32_t)y, (uint32_t)z, s.gq ? sweep_slim(s.gq, r) : __float_as_uint(r)};
  }
  __syncthreads();
  if (sh->bail) return -1;
  for (uint32_t i = tid; i < npath; i += nthr) if (s.alive[path[i]]) sweep_push(s, spare, 0u, 0u, path[i], i | SW_P | SW_D);
#ifdef KH_SWEEP_PROBE
  long long tq = clock64();
#define SW_T(i) if (tid == 0) { const long long n_ = clock64(); sh->cyc[i] += n_ - tq; tq = n_; }
#else
#define SW_T(i)
#endif
  uint32_t next_from = 0, committed = 0;
  for (;;) {
    __syncthreads();
    SW_T(5)   // pairs of pure P events (end of the previous level)
    // ---- commit the previous level (every push of it is done) and find the next non-empty level
    const uint32_t k1 = sh->nkill;   // stable: no deadline runs in this phase
    for (uint32_t i = committed + tid; i < k1; i += nthr) {
      const uint32_t v = s.killed[i];
      s.alive[v] = 0;
      s.cstate[v] = 0ull;
      s.sched[v] = SW_SCHED_DEAD;
    }
    SW_T(0)   // commit
    if (wave == 0) {
      // the chunks of the level that has just been processed are free again (their ids are still in the chain list)
      const uint32_t nprev = sh->nprev;
      const int nfree0 = sh->nfree;          // >= 0: every lane that found the stack empty has put its claim back
      const uint32_t room = SW_RING - min((uint32_t)nfree0, SW_RING);
      const uint32_t nput = min(nprev, room);              // (what does not fit the LDS stack is not reused in this call)
      for (uint32_t i = (uint32_t)lane; i < nput; i += 64u) s.fs[(uint32_t)nfree0 + i] = s.chain[i];
      uint32_t found = SW_NONE;
      if (s.wmask == 0xFFFFFFFFu) {
        // one bit per level of the label: the first set bit at or above next_from
        const uint32_t w0 = next_from >> 5;
        for (uint32_t base = w0; base < nwords; base += 64u) {
          const uint32_t idx = base + (uint32_t)lane;
          uint32_t v = idx < nwords ? sweep_ld(&s.lvbits[idx]) : 0u;
          if (idx == w0) v &= ~((1u << (next_from & 31u)) - 1u);
          const unsigned long long ball = __builtin_amdgcn_ballot_w64(v != 0u);
          if (ball) {
            const int l = __ffsll((long long)ball) - 1;
            const uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
            found = ((base + (uint32_t)l) << 5) + (uint32_t)(__ffs((int)vv) - 1);
            break;
          }
        }
      } else {
        // level window: the bits are used round robin; walk them from slot next_from & wmask once around (step j looks at
        // bitmap word (w0 + j) mod nw; the first word is split: its bits from p0 up open the walk, those below p0 close it)
        const uint32_t nw = s.nslots >> 5, p0 = next_from & s.wmask, w0 = p0 >> 5, lowbits = (1u << (p0 & 31u)) - 1u;
        for (uint32_t base = 0; base <= nw; base += 64u) {
          const uint32_t j = base + (uint32_t)lane;
          const uint32_t wi = (w0 + j) & (nw - 1u);
          uint32_t v = j <= nw ? sweep_ld(&s.lvbits[wi]) : 0u;
          if (j == 0u) v &= ~lowbits;
          if (j == nw) v &= lowbits;
          const unsigned long long ball = __builtin_amdgcn_ballot_w64(v != 0u);
          if (ball) {
            const int l = __ffsll((long long)ball) - 1;
            const uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
            const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)wi, l);
            const uint32_t slot = (wl << 5) + (uint32_t)(__ffs((int)vv) - 1);
            found = next_from + ((slot - p0) & s.wmask);
            break;
          }
        }
      }
      if (lane == 0) {
        if (sh->bail) found = SW_NONE;   // the only place the loop's exit is decided: every thread reads sh->lvl
        sh->lvl = found;
        if (found != SW_NONE) {
          const uint32_t fslot = found & s.wmask;
          const uint32_t w = __hip_atomic_exchange(&s.words[fslot], EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_and(&s.lvbits[fslot >> 5], ~(1u << (fslot & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // the level's chunks, newest first (a chain of dependent loads: one per chunk)
          uint32_t n = 0;
          for (uint32_t id = w >> SW_FILL_BITS; id != SW_NOCHUNK; id = s.chunks[(size_t)id << s.shift].x) {
            if (n == SW_CHAIN) { sweep_bail(s, SW_BAIL_LEVEL); found = SW_NONE; sh->lvl = SW_NONE; break; }
            s.chain[n++] = id;
          }
          const uint32_t newest = min(w & SW_FILL_MASK, 1u << s.shift) - 1u;   // (the fill of a full chunk keeps counting)
          sh->ord = newest;
          sh->nev = newest + (n - 1u) * ((1u << s.shift) - 1u);
          sh->levels++;
          sh->events += sh->nev;
          if (sh->nev > sh->maxnev) sh->maxnev = sh->nev;
          sh->nfree = nfree0 + (int32_t)nput;
          sh->nprev = n;
        }
        sh->nM -= (int32_t)(k1 - committed);
        sh->na = sh->nb = sh->nnp = 0u;
      }
    }
    committed = k1;
    __syncthreads();
    SW_T(1)   // next level + its chunk chain
    const uint32_t lvl = sh->lvl;
    if (lvl == SW_NONE) break;
    next_from = lvl + 1u;
    const uint32_t nev = sh->nev, newest = sh->ord;
    // ---- A: candidates -- the level's own P events, then the cascades (rare) off the list `wa`, through the same code.
    // sh->na is stable whenever it is read here: appends only happen inside the work loop, between two barriers
    {
      uint32_t lo = 0, hi = nev;
      bool lst = false;
      for (;;) {
        for (uint32_t i = lo + tid; i < hi; i += nthr) {
          uint32_t v, c;
          bool dl = false, go = true;
          if (!lst) {
            const u32x2_t ev = sweep_event(s, i, newest);
            go = (ev.y & SW_P) != 0u; v = ev.x; c = ev.y & 0x7fffu; dl = (ev.y & SW_D) != 0u;
          } else {
            const unsigned long long it = s.wa[i];
            v = (uint32_t)(it >> 32); c = (uint32_t)it;
          }
          if (go) { if (ikeys) sweep_possible_int(s, flt, lvl, v, c, dl); else sweep_possible_gen(s, flt, lvl, v, c, dl); }
        }
        __syncthreads();
        const uint32_t avail = sh->na < s.ncap ? sh->na : s.ncap;
        const uint32_t from = lst ? hi : 0u;
        __syncthreads();
        if (avail == from) break;
        lo = from; hi = avail; lst = true;
      }
    }
    SW_T(2)   // A and its cascade
    // ---- B: deadlines (a voxel's candidates are complete now) -- the level's own D events, then the cascades off `wb`
    {
      uint32_t lo = 0, hi = nev;
      bool lst = false;
      for (;;) {
        for (uint32_t i = lo + tid; i < hi; i += nthr) {
          uint32_t v, hint = SW_NONE;
          bool go = true;
          if (!lst) {
            const u32x2_t ev = sweep_event(s, i, newest);
            go = (ev.y & SW_D) != 0u; v = ev.x;
            if (ev.y & SW_P) hint = ev.y & 0x7fffu;
          } else {
            v = s.wb[i];
          }
          if (go) { if (ikeys) sweep_deadline_int(s, flt, spare, lvl, v, hint); else sweep_deadline_gen(s, flt, spare, lvl, v, hint); }
        }
        __syncthreads();
        const uint32_t avail = sh->nb < s.ncap ? sh->nb : s.ncap;
        const uint32_t from = lst ? hi : 0u;
        __syncthreads();
        if (avail == from) break;
        lo = from; hi = avail; lst = true;
      }
    }
    SW_T(6)   // B and its cascade
    // ---- pairs added by pure P events whose voxel survives the level: their possible nodes go out now
    {
      const uint32_t nnp = sh->nnp < s.ncap ? sh->nnp : s.ncap;
      for (uint32_t i = tid; i < nnp; i += nthr) {
        const unsigned long long it = s.np[i];
        const uint32_t v = (uint32_t)(it >> 32);
        if (!(s.cstate[v] & SW_DYING)) sweep_emit_possible(s, flt, spare, lvl, v, (uint32_t)it);
      }
    }
  }
  __syncthreads();
  uint32_t bail = sh->bail;
  const int nM = sh->nM;
  if (!bail && nM != 0 && !allow_ghosts) bail = SW_BAIL_M;
  const uint32_t nk = sh->nkill, nsp = sh->nspill;
  __syncthreads();
  if (nsp) {
    // the spill table goes back to all-free (only a call that used it pays for this)
    for (uint32_t i = tid; i < s.spcap; i += nthr) { s.spk[i] = 0u; s.spc[i] = 0ull; }
  }
  if (bail) {
    // undo: the voxels this call killed are alive again (a ghost whose kill had been committed comes back as a plain live voxel, not
    // as a ghost: the caller never continues from that state -- with ghosts alive it answers a bail by rolling the whole journal back,
    // trace.hip "heap_ok"), every per-voxel word of the label is cleared
    if (tid == 0) sh->bail = bail;
    for (uint32_t i = tid; i < nk; i += nthr) { const uint32_t v = s.killed[i]; s.alive[v] = (uint8_t)(s.alive[v] == 0 ? 1 : s.alive[v]); }
    for (uint32_t i = tid; i < nf; i += nthr) s.cstate[list[i]] = 0ull;
    __syncthreads();
    sweep_reset_words(s, list, nf);        // (voxels that were dead before the call stay "dead")
    __syncthreads();
    return -1;
  }
  if (nM != 0) {
    // undecided voxels = the ones that still carry candidates: they become ghosts (an old ghost that was touched again stays one)
    for (uint32_t i = tid; i < nf; i += nthr) {
      const uint32_t v = list[i];
      if (s.cstate[v] == 0ull) continue;
      s.cstate[v] = 0ull;
      if (s.alive[v] == 1) {
        s.alive[v] = SW_GHOST;
        s.killed[nk + SW_L_ADD(&sh->nghost, 1u)] = v;
      }
    }
    __syncthreads();
  }
  return (long long)(nk - sh->nkg);
}

}  // namespace kh
