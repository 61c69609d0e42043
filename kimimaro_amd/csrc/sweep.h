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
static constexpr uint32_t SW_CHAIN = 512;                        // chunks per level the reader can take (LDS list)
static constexpr int SW_FILL_BITS = 12;                          // level word = (newest chunk << SW_FILL_BITS) | next free slot.  The fill
static constexpr uint32_t SW_FILL_MASK = (1u << SW_FILL_BITS) - 1u;   // field keeps counting while a chunk is full: up to 13 adds per thread (sweep_push13)
static constexpr uint32_t SW_NOCHUNK = 0xFFFFFu;                 // of a 256-thread workgroup + the chunk size must fit it; 20-bit chunk ids
static constexpr uint32_t SW_SCHED_NONE = 0xFFFFFFFFu;           // sched word of a voxel without a pending deadline
static constexpr uint32_t SW_SCHED_LEVELS = (1u << 17) - 1u;     // with up to 32766 sources (15 bits) the filter holds this many levels
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
  unsigned long long cyd[8];   // ... and of thread 0's own deadline event: own words / neighbours' alive bytes / ranks / filter words / pushes; [7] = events
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
  KH_AS_GLOBAL uint32_t* sched;    // per voxel: earliest pending deadline (level << cb | source + 1), SW_SCHED_NONE = none;
                                   // nullptr = no filter (every event is pushed)
  const KH_AS_GLOBAL uint32_t* rank;   // [ra * rb * rc]
  int ra, rb;
  KH_AS_GLOBAL u32x4_t* srcs;      // per path vertex {x, y, z, radius bits} (HBM)
  KH_AS_GLOBAL u32x2_t* chunks;    // arena: chunk c = slots [c << shift, (c + 1) << shift); slot 0 = {previous chunk of the level, -}
  KH_AS_GLOBAL uint32_t* fs;       // [chcap] free stack: ids of chunks whose level has been processed (HBM, front of the arena)
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
#define SW_G_MIN(p, v) __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#ifndef KH_SWEEP_LOOK
#define KH_SWEEP_LOOK 1     /* sweep_claim13 reads sched[q] before its atomicMin (0: the atomic straight away, rounds 4-5) */
#endif
// Round 6: no read-modify-write on the filter's words.  Every global atomic of gfx950 is executed on the memory side of the
// fabric and DROPS its line from the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"; the ISA has no scope bit on an
// atomic: agent and workgroup scope compile to the same instruction), so each atomicMin on sched[] cost a fabric round trip
// and made the next look at that line -- by the neighbours that die next -- miss the L2 as well.  The filter does not need the
// minimum: it needs the word to hold the value of SOME deadline-carrying event that was pushed for the voxel (or "none").
// A skipped event is then still an exact duplicate or lies behind a pending deadline; a word that is larger than the true
// minimum (two lanes stored at once and the larger value landed) only lets a few more no-op events through.  So: look, and
// store when lower -- plain stores keep the line in the L2 (KH_SWEEP_PLAIN_SCHED=0: the atomic of rounds 4-5 for A/B runs).
#ifndef KH_SWEEP_PLAIN_SCHED
#define KH_SWEEP_PLAIN_SCHED 1
#endif
#if KH_SWEEP_PLAIN_SCHED
#define SW_SCHED_LOWER(p, v) (*(p) = (v))
#else
#define SW_SCHED_LOWER(p, v) ((void)SW_G_MIN(p, v))
#endif
template <class T>
__device__ __forceinline__ T sw_g_cas(KH_AS_GLOBAL T* p, T expect, T want) {      // returns the old value like atomicCAS
  __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expect;
}

__device__ __forceinline__ void sweep_bail(SweepRef s, uint32_t why) { SW_L_OR(&s.sh->bail, why); }
// The level words and the bitmap are updated with atomics (LDS, or L2 when they live in HBM): read them the same way,
// a plain load could be served from a stale line of the CU's vector cache.
__device__ __forceinline__ uint32_t sweep_ld(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t sweep_ld(const KH_AS_LDS uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t sweep_ld(const KH_AS_GLOBAL uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// distance of voxel (qx, qy, qz) from the source, with the float operation order of dijkstra_invalidation.hpp:310-316
__device__ __forceinline__ bool sweep_eval(SweepRef s, const u32x4_t src, int qx, int qy, int qz, uint32_t& rk) {
  const int ex = qx - (int)src.x, ey = qy - (int)src.y, ez = qz - (int)src.z;
  const float a = s.g->wx * (float)ex;
  const float b = s.g->wy * (float)ey;
  const float c = s.g->wz * (float)ez;
  float t = a * a;
  const float u = b * b;
  const float v = c * c;
  t = t + u;
  t = t + v;
  const float d = sqrtf(t);
  if (!(d < __uint_as_float(src.w))) return false;
  const int ax = ex < 0 ? -ex : ex, ay = ey < 0 ? -ey : ey, az = ez < 0 ? -ez : ez;
  rk = s.rank[ax + s.ra * (ay + s.rb * az)];
  return true;
}

// Event storage.  A level's events live in a chain of fixed-size chunks (slot 0 of a chunk = id of the previous
// chunk of the level); the level's LDS word holds the newest chunk and its next free slot.
// sweep_push appends an event to level lv.  Free of wait states (a lane never waits for another lane's future action,
// which lock-step execution could not deliver).  Fast path: one atomic add on the level word takes the next slot -- no
// retry however many lanes push to the same level at once (a CAS here costs a retry per concurrent pusher, and a busy
// level has hundreds).  A lane whose add finds the chunk full (or the level empty: the empty word reads as full)
// installs a fresh chunk with a CAS on the overfull word and takes its slot 1; whoever loses that race just starts over,
// and the chunk it had reserved stays with the thread (`spare`) for its next opening.  The fill field of a full chunk
// keeps counting (at most 13 adds per thread before the install, sweep_push13: < 1 << SW_FILL_BITS), readers clamp it to the chunk size.
// Level window.  An event goes to a level ahead of the one being processed (`cur`), and never far ahead: its key is the
// distance of a NEIGHBOUR of the processed voxel from a source whose key of that voxel is not above the current level, so
// it exceeds the current key by one step at most -- a few hundred to a few thousand levels, which the host bounds from the
// key table (kh_label_t.lev_window).  The level words are therefore kept for a window of nslots levels only, level lv in
// slot lv & wmask: 4-8 KiB of LDS instead of 4 bytes for every level of the label.  The bound is checked, not trusted: an
// event that would leave the window abandons the call (SW_BAIL_LEVEL -> heap emulation).
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
// The same event (meta) to the neighbours K0 .. K0+12 of v named by `push`, neighbour k at level rk[k - K0].  A dying voxel hands
// deadlines to several neighbours; one after the other that was, per push, a reload of the rank (a run-time index into the rank
// registers would put them in scratch memory), an LDS atomic and a store -- a chain per lane, and the wave runs to its widest lane.
// Here the slot of every push is taken first (13 LDS atomics in flight together, the ranks picked by constant indices), then the
// stores go out; only a push that finds its chunk full (a few per cent: the first event of a level opens one) takes the loop.
template <int K0>
__device__ __forceinline__ void sweep_push13(SweepRef s, uint32_t& spare, uint32_t cur, uint32_t v, uint32_t push,
                                             const uint32_t (&rk)[13], uint32_t meta) {
  const uint32_t wmask = s.wmask, CH = 1u << s.shift;
  const int shift = s.shift;
  KH_AS_LDS uint32_t* words = s.words;
  KH_AS_GLOBAL u32x2_t* chunks = s.chunks;
  const int sx = s.g->sx, sxy = s.g->sxy;
  uint32_t w[13];
  uint32_t far = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    w[j] = 0u;
    if ((push >> (K0 + j)) & 1u) {
      if (rk[j] - cur > wmask) far |= 1u << j;
      else w[j] = SW_L_ADD(&words[rk[j] & wmask], 1u);
    }
  }
  if (far) { sweep_bail(s, SW_BAIL_LEVEL); push &= ~(far << K0); }
  uint32_t slow = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    if (!((push >> (K0 + j)) & 1u)) continue;
    int dx, dy, dz;
    dir_delta(K0 + j, dx, dy, dz);
    const uint32_t fill = w[j] & SW_FILL_MASK;
    if (fill < CH) chunks[((size_t)(w[j] >> SW_FILL_BITS) << shift) + fill] = u32x2_t{v + (uint32_t)(dx + sx * dy + sxy * dz), meta};
    else slow |= 1u << j;
  }
  if (slow) {
#pragma unroll
    for (int j = 0; j < 13; j++) {
      if (!((slow >> j) & 1u)) continue;
      int dx, dy, dz;
      dir_delta(K0 + j, dx, dy, dz);
      const uint32_t slot = rk[j] & wmask;
      sweep_push_from(s, spare, &words[slot], slot, v + (uint32_t)(dx + sx * dy + sxy * dz), meta, w[j]);
    }
  }
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

// ---- neighbour batches.  An event looks at its 26 neighbours; done one after the other that is a chain of ~50
// dependent L2 round trips per event (alive byte, then rank word, per neighbour) and the sweep is nothing but such
// chains.  The helpers below issue the loads of all neighbours before any is consumed (full unrolling, addresses made
// valid by predication instead of branches), so an event costs a handful of round trips.
__device__ __forceinline__ uint32_t sweep_alive_nbrs(SweepRef s, uint32_t v, uint32_t nm) {
  const int sx = s.g->sx, sxy = s.g->sxy;
  uint32_t am = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) {
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    const uint32_t q = ((nm >> k) & 1u) ? v + (uint32_t)(dx + sx * dy + sxy * dz) : v;
    am |= (uint32_t)(s.alive[q] != 0) << k;
  }
  return am & nm;
}
// neighbours K0 .. K0+12 of (x, y, z) that are in `am` and inside the ball of src: coverage mask (bit k) + their ranks
template <int K0>
__device__ __forceinline__ uint32_t sweep_eval13(SweepRef s, const u32x4_t src, int x, int y, int z, uint32_t am,
                                                 uint32_t (&rk)[13]) {
  const float r = __uint_as_float(src.w);
  const float wx = s.g->wx, wy = s.g->wy, wz = s.g->wz;
  uint32_t cov = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    const int k = K0 + j;
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    const int ex = x + dx - (int)src.x, ey = y + dy - (int)src.y, ez = z + dz - (int)src.z;
    const float a = wx * (float)ex;
    const float b = wy * (float)ey;
    const float c = wz * (float)ez;
    float t = a * a;
    const float u = b * b;
    const float w = c * c;
    t = t + u;
    t = t + w;
    const float d = sqrtf(t);
    const bool in = ((am >> k) & 1u) && d < r;
    const int ax = ex < 0 ? -ex : ex, ay = ey < 0 ? -ey : ey, az = ez < 0 ? -ez : ez;
    rk[j] = s.rank[in ? ax + s.ra * (ay + s.rb * az) : 0];
    cov |= (uint32_t)in << k;
  }
  return cov;
}
// covered neighbours whose rank lies above `lvl` (the ranks are looked at with constant indices only: a rank picked by a
// run-time index turns the two arrays into a scratch-memory table -- 26 stores per event and a dependent load per pick)
__device__ __forceinline__ uint32_t sweep_above(const uint32_t (&rk0)[13], const uint32_t (&rk1)[13], uint32_t cov, uint32_t lvl) {
  uint32_t up = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    up |= (uint32_t)(rk0[j] > lvl) << j;
    up |= (uint32_t)(rk1[j] > lvl) << (13 + j);
  }
  return up & cov;
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
// neighbour k (run-time index) of v = (x, y, z), known to be covered by src: its voxel index and the rank of its key
__device__ __forceinline__ uint32_t sweep_nbr_rank(SweepRef s, const u32x4_t src, uint32_t v, int x, int y, int z, int k,
                                                   uint32_t& q) {
  int dx, dy, dz;
  sweep_dir(k, dx, dy, dz);
  q = v + (uint32_t)(dx + s.g->sx * dy + s.g->sxy * dz);
  const int ex = x + dx - (int)src.x, ey = y + dy - (int)src.y, ez = z + dz - (int)src.z;
  const int ax = ex < 0 ? -ex : ex, ay = ey < 0 ? -ey : ey, az = ez < 0 ? -ez : ez;
  return s.rank[ax + s.ra * (ay + s.rb * az)];
}

// ---- pending-deadline filter.  An event that carries a deadline (D or PD) for voxel q at level t makes every event of q
// at a later level a no-op (q is dead once level t is complete: sweep_possible and sweep_deadline return at their alive
// test), and a second identical event is a no-op as well (the first adds the candidate / sets the dying bit, the second
// returns at its first test).  A voxel is handed such events by every neighbour that dies before it does -- about a dozen
// per voxel, nearly all of them the voxel's own key from the same source -- so unfiltered the sweep stores, reloads and
// dismisses ~13 events per voxel.  sched[q] holds the smallest (level << cb | code) over the deadline-carrying events
// pushed to q so far (code = source + 1 for PD, 0 for a pure D; cb = the bits the codes of THIS call need, so a call with few
// sources -- a soma's single root, the plates of the reference's tests -- can have millions of levels).  A new event is pushed iff it lowers the word (an earlier
// deadline) or ties its level under another code (two sources on the same key: the tie the certificate is about).
// Skipped are only exact duplicates and events at a level above a pending deadline: the machine's states, bails and
// result are those of the unfiltered sweep.  Every word that is ever set belongs to a voxel that is dead at the end of a
// certified call (its deadline was processed) and dead voxels are never offered events, so the words of live voxels
// read SW_SCHED_NONE at the start of every call; a bail resets the label's words together with cstate.
struct SweepFilter {
  KH_AS_GLOBAL uint32_t* sched;   // nullptr: this call runs unfiltered
  int cb;            // bits of the source code
};
// the filter of a call with npath sources on a label with nlev levels: every (level << cb | code) stays below SW_SCHED_NONE
__device__ __forceinline__ SweepFilter sweep_filter(SweepRef s, uint32_t npath) {
  SweepFilter f;
  f.cb = 32 - __clz((int)npath);                 // codes 1 .. npath
  f.sched = (s.sched != nullptr && s.nlev <= (0xFFFFFFFFu >> f.cb)) ? s.sched : nullptr;
  return f;
}
__device__ __forceinline__ bool sweep_claim(const SweepFilter f, uint32_t q, uint32_t tr, uint32_t code) {
  if (f.sched == nullptr) return true;
  const uint32_t val = (tr << f.cb) | code;
#if KH_SWEEP_PLAIN_SCHED
  const uint32_t old = sweep_ld(&f.sched[q]);
  if (val < old) f.sched[q] = val;
#else
  const uint32_t old = SW_G_MIN(&f.sched[q], val);
#endif
  return val < old || ((old >> f.cb) == tr && old != val);
}
// a pure P event of q at level tr is a no-op when a deadline of q is pending at an earlier level
__device__ __forceinline__ bool sweep_moot(const SweepFilter f, uint32_t q, uint32_t tr) {
  return f.sched != nullptr && (sweep_ld(&f.sched[q]) >> f.cb) < tr;
}
// the same for neighbours K0 .. K0+12 of v at once (`want`: bit k = neighbour k gets a PD event of source code - 1 at
// level rk[k - K0]): all atomics are in flight before the first result is looked at.  Returns the events to push.
// (sched, sx, sxy are handed over in registers: the Sweep record lives in LDS, and a flat atomic counts on the LDS
// counter as well, so an LDS read between two atomics would wait for the first one to return)
template <int K0>
__device__ __forceinline__ uint32_t sweep_claim13(KH_AS_GLOBAL uint32_t* sched, int cb, int sx, int sxy, uint32_t v, uint32_t want,
                                                  const uint32_t (&rk)[13], uint32_t code) {
  uint32_t old[13];
#if KH_SWEEP_LOOK
  // A look before the atomic.  A voxel is handed the same deadline by about a dozen dying neighbours and only the first atomicMin
  // lowers its word; the others are read-modify-writes that leave the line dirty for nothing (the path kernel wrote 138 GB per c3
  // volume, a third of its traffic).  The words only go down within a call, so a coherent (L2) read that already shows a value <=
  // ours decides like the atomic's return value would ("identical" and "a deadline at an earlier level" stay true whatever happens
  // to the word later).  A read that shows a larger value is followed by the atomic WITHOUT return: two neighbours that die in the
  // same level both push then -- an identical event is a no-op of the machine, only its slot is spent.
#pragma unroll
  for (int j = 0; j < 13; j++) {
    const int k = K0 + j;
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    old[j] = 0u;
    if ((want >> k) & 1u) old[j] = sweep_ld(&sched[v + (uint32_t)(dx + sx * dy + sxy * dz)]);
  }
  uint32_t keep = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    const int k = K0 + j;
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    const uint32_t val = (rk[j] << cb) | code;
    const bool lower = ((want >> k) & 1u) && val < old[j];
    if (lower) SW_SCHED_LOWER(&sched[v + (uint32_t)(dx + sx * dy + sxy * dz)], val);
    keep |= (uint32_t)(lower || ((old[j] >> cb) == rk[j] && old[j] != val)) << k;
  }
  return keep & want;
#else
#pragma unroll
  for (int j = 0; j < 13; j++) {
    const int k = K0 + j;
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    old[j] = 0u;
    if ((want >> k) & 1u) old[j] = SW_G_MIN(&sched[v + (uint32_t)(dx + sx * dy + sxy * dz)], (rk[j] << cb) | code);
  }
  uint32_t keep = 0;
#pragma unroll
  for (int j = 0; j < 13; j++) {
    const uint32_t val = (rk[j] << cb) | code;
    keep |= (uint32_t)(val < old[j] || ((old[j] >> cb) == rk[j] && old[j] != val)) << (K0 + j);
  }
  return keep & want;
#endif
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
__device__ __forceinline__ void sweep_possible(SweepRef s, uint32_t lvl, uint32_t v, uint32_t c, bool has_deadline) {
  const uint8_t live = s.alive[v];
  unsigned long long cs = s.cstate[v];
  const uint32_t nm = sweep_mask(s, v, s.nbrmask[v]);
  const u32x4_t src = s.srcs[c];
  if (!live) return;
  // the neighbours' alive bytes are asked for before the CAS below, so that the two round trips overlap (an event that
  // turns out to be a duplicate has loaded them in vain: rare since the pending-deadline filter)
  const uint32_t am = sweep_alive_nbrs(s, v, nm);
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
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  uint32_t lm = 0;
  {
    uint32_t rk0[13], rk1[13];
    const uint32_t cov = sweep_eval13<0>(s, src, x, y, z, am, rk0) | sweep_eval13<13>(s, src, x, y, z, am, rk1);
#pragma unroll
    for (int j = 0; j < 13; j++) {
      lm |= (uint32_t)(((cov >> j) & 1u) && rk0[j] <= lvl) << j;
      lm |= (uint32_t)(((cov >> (13 + j)) & 1u) && rk1[j] <= lvl) << (13 + j);
    }
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

// P events of the pair (v, c) to the levels above lvl (the pair was added by a pure P event and v survives the level)
__device__ __forceinline__ void sweep_emit_possible(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                                    uint32_t c) {
  const uint32_t nm = sweep_mask(s, v, s.nbrmask[v]);
  const u32x4_t src = s.srcs[c];
  const uint32_t am = sweep_alive_nbrs(s, v, nm);
  if (!am) return;
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  uint32_t rk0[13], rk1[13];
  const uint32_t cov = sweep_eval13<0>(s, src, x, y, z, am, rk0) | sweep_eval13<13>(s, src, x, y, z, am, rk1);
  for (uint32_t m = sweep_above(rk0, rk1, cov, lvl); m; m &= m - 1u) {
    uint32_t q;
    const uint32_t r = sweep_nbr_rank(s, src, v, x, y, z, __ffs((int)m) - 1, q);
    if (!sweep_moot(flt, q, r)) sweep_push(s, spare, lvl, r, q, c | SW_P);
  }
}

// the neighbours of a dying voxel with a single candidate source: covered ones die with it (deadline at their own key)
__device__ __forceinline__ void sweep_deadline_one(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                                   uint32_t cid, const u32x4_t src, int x, int y, int z, uint32_t am) {
  uint32_t rk0[13], rk1[13];
  const uint32_t cov = sweep_eval13<0>(s, src, x, y, z, am, rk0) | sweep_eval13<13>(s, src, x, y, z, am, rk1);
  const uint32_t up = sweep_above(rk0, rk1, cov, lvl);   // covered neighbours whose own key lies above this level: a PD event there
  SW_DT(2, up);
  uint32_t push = up;
  if (flt.sched != nullptr) {
    const int sx = s.g->sx, sxy = s.g->sxy;
    push = sweep_claim13<0>(flt.sched, flt.cb, sx, sxy, v, up, rk0, cid + 1u) |
           sweep_claim13<13>(flt.sched, flt.cb, sx, sxy, v, up, rk1, cid + 1u);
  }
  SW_DT(3, push);
  for (uint32_t m = cov & ~up; m; m &= m - 1u) {      // same level: the cascade of this level
    const uint32_t q = v + (uint32_t)s.g->off[__ffs((int)m) - 1];
    if (s.cstate[q] & SW_DYING) continue;
    const uint32_t p = SW_L_ADD(&s.sh->nb, 1u);
    if (p < s.ncap) s.wb[p] = q; else sweep_bail(s, SW_BAIL_LIST);
  }
  SW_DT(4, push);
  if (push) {
    sweep_push13<0>(s, spare, lvl, v, push, rk0, cid | SW_P | SW_D);
    sweep_push13<13>(s, spare, lvl, v, push, rk1, cid | SW_P | SW_D);
  }
  SW_DT(5, spare);
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
      const bool cov = sweep_eval(s, s.srcs[sl - 1u], x + dx, y + dy, z + dz, rk);
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

// a D event: v is dead once this level is complete
// `hint`: the source of the event when it is a PD event (the usual single candidate of its voxel), SW_NONE otherwise.
// The first round trip carries everything that does not depend on anything else: the voxel's alive byte and neighbour mask,
// the hinted source record, and the dying bit itself -- set before the alive byte is known and taken back when the voxel
// turns out to be dead (a dead voxel's word is zero and nobody looks at it: dead voxels are not in anybody's neighbour set);
// the neighbours' alive bytes follow as soon as the mask is there, while the atomic is still on its way.
__device__ __forceinline__ void sweep_deadline(SweepRef s, const SweepFilter flt, uint32_t& spare, uint32_t lvl, uint32_t v,
                                               uint32_t hint) {
  const uint8_t live = s.alive[v];
  const uint32_t nm = sweep_mask(s, v, s.nbrmask[v]);
  const u32x4_t hsrc = s.srcs[hint != SW_NONE ? hint : 0u];
  SW_D0();
  const unsigned long long old = SW_G_OR(&s.cstate[v], SW_DYING);
  SW_DT(0, (uint32_t)old + live + nm + hsrc.x);
  const uint32_t am = sweep_alive_nbrs(s, v, nm);
  SW_DT(1, am);
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
  if ((old & SW_SPILL) || (ghost && nc > 1)) {
    sweep_deadline_many(s, flt, spare, lvl, v, old, (old & SW_SPILL) ? sweep_spill_get(s, v) : 0ull, am, ghost);
    return;
  }
  if (nc == 1) {
    uint32_t c1 = cid[0];
#pragma unroll
    for (int i = 1; i < 4; i++) if (has[i]) c1 = cid[i];
    if (ghost) { sweep_emit_possible(s, flt, spare, lvl, v, c1); return; }   // its possible nodes, no deadlines
    int x, y, z;
    sweep_coords(s, v, x, y, z);
    const u32x4_t s1 = c1 == hint ? hsrc : s.srcs[c1];       // (the hinted record is here already)
    sweep_deadline_one(s, flt, spare, lvl, v, c1, s1, x, y, z, am);
    return;
  }
  int x, y, z;
  sweep_coords(s, v, x, y, z);
  u32x4_t src[4];
#pragma unroll
  for (int i = 0; i < 4; i++) src[i] = s.srcs[has[i] ? cid[i] : 0u];
  for (uint32_t m = am; m; m &= m - 1u) {
    const int k = __ffs((int)m) - 1;
    const uint32_t q = v + (uint32_t)s.g->off[k];
    int dx, dy, dz;
    dir_delta(k, dx, dy, dz);
    bool all = true;
    uint32_t tr = 0, rk[4];
    bool cov[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      cov[i] = false;
      rk[i] = 0u;
      if (has[i]) {
        cov[i] = sweep_eval(s, src[i], x + dx, y + dy, z + dz, rk[i]);
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

// event e of the level being processed: the newest chunk (chain[0]) holds `newest` events, the others are full
__device__ __forceinline__ u32x2_t sweep_event(SweepRef s, uint32_t e, uint32_t newest) {
  const uint32_t per = (1u << s.shift) - 1u;
  uint32_t c = 0, slot = e + 1u;
  if (e >= newest) { const uint32_t r = e - newest; c = 1u + r / per; slot = r - (c - 1u) * per + 1u; }
  return s.chunks[((size_t)s.chain[c] << s.shift) + slot];
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
__device__ __forceinline__ bool sweep_ball(SweepRef s, const uint32_t* path, uint32_t npath, const float* __restrict__ dbf,
                                                     float scale, float constant, float rmax, const uint32_t* list, uint32_t nf,
                                                     uint32_t* count, bool allow_ghosts = false) {
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  KH_AS_LDS SweepShared* sh = s.sh;
  const SweepFilter flt = sweep_filter(s, npath);
  uint32_t spare = SW_NONE;
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
    srcs[i] = u32x4_t{(uint32_t)x, (uint32_t)y, (uint32_t)z, __float_as_uint(r)};
  }
  __syncthreads();
  if (sh->bail) return false;
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
    }
    SW_T(0)   // commit
    if (wave == 0) {
      // the chunks of the level that has just been processed are free again (their ids are still in the chain list)
      const uint32_t nprev = sh->nprev;
      const int nfree0 = sh->nfree;          // >= 0: every lane that found the stack empty has put its claim back
      for (uint32_t i = (uint32_t)lane; i < nprev; i += 64u) s.fs[(uint32_t)nfree0 + i] = s.chain[i];
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
          sh->nfree = nfree0 + (int32_t)nprev;
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
    // ---- A: candidates
    for (uint32_t e = tid; e < nev; e += nthr) {
      const u32x2_t ev = sweep_event(s, e, newest);
      if (ev.y & SW_P) sweep_possible(s, lvl, ev.x, ev.y & 0x7fffu, (ev.y & SW_D) != 0u);
    }
    __syncthreads();
    SW_T(2)   // A
    // cascades (rare).  sh->na is stable whenever it is read here: appends only happen between the two barriers below
    for (uint32_t done = 0;;) {
      const uint32_t avail = sh->na < s.ncap ? sh->na : s.ncap;
      if (avail == done) break;
      if (tid == 0) sh->snap = avail;
      __syncthreads();
      const uint32_t end = sh->snap;
      for (uint32_t i = done + tid; i < end; i += nthr) {
        const unsigned long long it = s.wa[i];
        sweep_possible(s, lvl, (uint32_t)(it >> 32), (uint32_t)it, false);
      }
      done = end;
      __syncthreads();
    }
    SW_T(3)   // cascade of A
    // ---- B: deadlines (a voxel's candidates are complete now)
    for (uint32_t e = tid; e < nev; e += nthr) {
      const u32x2_t ev = sweep_event(s, e, newest);
      if (ev.y & SW_D) sweep_deadline(s, flt, spare, lvl, ev.x, (ev.y & SW_P) ? (ev.y & 0x7fffu) : SW_NONE);
    }
    __syncthreads();
    SW_T(6)   // B
    for (uint32_t done = 0;;) {
      const uint32_t avail = sh->nb < s.ncap ? sh->nb : s.ncap;
      if (avail == done) break;
      if (tid == 0) sh->snap = avail;
      __syncthreads();
      const uint32_t end = sh->snap;
      for (uint32_t i = done + tid; i < end; i += nthr) sweep_deadline(s, flt, spare, lvl, s.wb[i], SW_NONE);
      done = end;
      __syncthreads();
    }
    SW_T(4)   // cascade of B
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
    // undo: the killed voxels come back (ghosts as ghosts), every per-voxel word of the label is cleared
    if (tid == 0) sh->bail = bail;
    for (uint32_t i = tid; i < nk; i += nthr) { const uint32_t v = s.killed[i]; s.alive[v] = (uint8_t)(s.alive[v] == 0 ? 1 : s.alive[v]); }
    for (uint32_t i = tid; i < nf; i += nthr) s.cstate[list[i]] = 0ull;
    if (s.sched != nullptr) for (uint32_t i = tid; i < nf; i += nthr) s.sched[list[i]] = SW_SCHED_NONE;
    __syncthreads();
    return false;
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
  *count = nk - sh->nkg;
  return true;
}

}  // namespace kh
