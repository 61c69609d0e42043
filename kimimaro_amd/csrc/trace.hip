// trace.hip -- the per-label TEASAR searches for gfx950: one workgroup (256 threads) per label.
//
//   kh_edf_batch    a4  dijkstra3d.euclidean_distance_field   (kimimaro/trace.py:139-145, 302-307)
//   kh_trace_paths  a6-a11 compute_paths loop                 (kimimaro/trace.py:196-267):
//                     target finder   skeletontricks.pyx:995-1045
//                     railroad        dijkstra3d.railroad, trace.py:240-242
//                     invalidation    skeletontricks.pyx:373-418 -> dijkstra_invalidation.hpp:239-332
//                     rail edits      trace.py:220, 261-263
//
// Search = label-correcting relaxation with a near/far split (delta-stepping with an adaptive
// threshold): work items are (frontier voxel, direction) pairs spread over the 256 lanes, distances
// are float bit patterns updated with atomicMin, new frontier entries are (dist,voxel) pairs appended
// with LDS counters (hipcc folds the per-lane atomicAdd into one per wave), stale entries are
// recognised by dist[v] != entry.dist.  The distances converge to the unique Bellman fixpoint
// d[v] = min_u fl(d[u] + w), so they equal the oracle's heap Dijkstra bit for bit regardless of the
// relaxation order; paths are then recovered with the canonical predecessor rule of
// oracle/kimi_oracle.c (ko_pred), 26 lanes looking at the 26 neighbours at once.
//
// The invalidation flood is order dependent down to the tie order of
// std::priority_queue, so it is run as an exact emulation of the libstdc++ binary heap by one lane,
// with the 26 neighbour tests of each popped voxel evaluated by 26 lanes.
//
// No MFMA: irregular, latency/atomic bound integer+f32 work (north_star).  Labels are independent, so
// the chip is filled by running every label's workgroup concurrently (8 workgroups per CU).
#include "common.h"

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

struct Ctl {
  unsigned long long best_rail;
  unsigned long long red64[4];
  uint32_t n_cur, n_next, n_far, n_far2, n_touched;
  uint32_t status;
  float red_min[4];
  float red_sum[4];
  uint32_t red_cnt[4];
  float T;
  uint32_t u0, u1, u2, u3;
};

struct Queues {
  uint64_t* a;
  uint64_t* b;
  uint64_t* c;
  uint32_t* touched;  // capacity 2*cap
  uint32_t cap;
};

// Returns with ctl->best_rail set (RAIL) ; distances below the final threshold are exact.
template <bool RAIL>
__device__ void sssp(const Geometry& g, const uint32_t* __restrict__ nbrmask, const float* __restrict__ wfield,
                     float* dist, uint32_t source, Queues q, Ctl* ctl, float delta_floor) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  uint64_t* cur = q.a;
  uint64_t* next = q.b;
  uint64_t* far = q.c;
  float T = RAIL ? 1e-45f : delta_floor;
  if (tid == 0) {
    ctl->n_cur = 1; ctl->n_next = 0; ctl->n_far = 0; ctl->n_far2 = 0;
    ctl->n_touched = 0;
    ctl->best_rail = NONE64;
    cur[0] = pack(0.0f, source);
    st_f32_l2(&dist[source], 0.0f);
    if (RAIL) { q.touched[0] = source; ctl->n_touched = 1; }
  }
  __syncthreads();
  for (;;) {
    // ---- near phase: label-correct everything below T
    for (;;) {
      const uint32_t n = ctl->n_cur;
      if (n == 0) break;
      const uint64_t items = (uint64_t)n << 5;
      for (uint64_t w = tid; w < items; w += 256) {
        const int k = (int)(w & 31);
        if (k >= 26) continue;
        const uint64_t e = cur[w >> 5];
        const uint32_t u = (uint32_t)e;
        const uint32_t dbits = (uint32_t)(e >> 32);
        if (!((nbrmask[u] >> k) & 1u)) continue;
        if (__float_as_uint(ld_f32_l2(&dist[u])) != dbits) continue;  // stale entry
        const uint32_t v = u + (uint32_t)g.off[k];
        const float wn = RAIL ? wfield[v] : g.w[k];
        const float nd = __uint_as_float(dbits) + wn;
        const uint32_t nb = __float_as_uint(nd);
        const uint32_t old = atomicMin(reinterpret_cast<uint32_t*>(&dist[v]), nb);
        if (nb < old) {
          if (RAIL && old == INF_BITS) {
            const uint32_t t = atomicAdd(&ctl->n_touched, 1u);
            if (t < 2u * q.cap) q.touched[t] = v; else atomicOr(&ctl->status, KH_ST_QUEUE_OVERFLOW);
          }
          if (RAIL && wn == 0.0f) {  // a rail: absorbing
            atomicMin(&ctl->best_rail, pack(nd, v));
            continue;
          }
          if (nd < T) {
            const uint32_t p = atomicAdd(&ctl->n_next, 1u);
            if (p < q.cap) next[p] = pack(nd, v); else atomicOr(&ctl->status, KH_ST_QUEUE_OVERFLOW);
          } else {
            const uint32_t p = atomicAdd(&ctl->n_far, 1u);
            if (p < q.cap) far[p] = pack(nd, v); else atomicOr(&ctl->status, KH_ST_QUEUE_OVERFLOW);
          }
        }
      }
      __syncthreads();
      if (tid == 0) {
        ctl->n_cur = ctl->n_next < q.cap ? ctl->n_next : q.cap;
        ctl->n_next = 0;
        if (ctl->n_far > q.cap) ctl->n_far = q.cap;
      }
      uint64_t* t = cur; cur = next; next = t;
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
    // pass 1: min / mean of the live far entries
    float mn = KH_INF, sm = 0.0f;
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < nfar; i += 256) {
      const uint64_t e = far[i];
      const uint32_t dbits = (uint32_t)(e >> 32);
      if (__float_as_uint(ld_f32_l2(&dist[(uint32_t)e])) != dbits) continue;
      const float d = __uint_as_float(dbits);
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
    mn = fminf(fminf(ctl->red_min[0], ctl->red_min[1]), fminf(ctl->red_min[2], ctl->red_min[3]));
    sm = ctl->red_sum[0] + ctl->red_sum[1] + ctl->red_sum[2] + ctl->red_sum[3];
    cnt = ctl->red_cnt[0] + ctl->red_cnt[1] + ctl->red_cnt[2] + ctl->red_cnt[3];
    if (cnt == 0) {  // only stale entries left
      __syncthreads();
      if (tid == 0) ctl->n_far = 0;
      __syncthreads();
      break;
    }
    const float mean = sm / (float)cnt;
    float step = 0.5f * (mean - mn);
    if (!(step > delta_floor)) step = delta_floor;
    float Tn = mn + step;
    if (!(Tn > mn)) Tn = next_up(mn);
    if (Tn > tcap) Tn = tcap;
    if (RAIL && !(mn < tcap)) {  // nothing left at or below the best rail distance
      break;
    }
    T = Tn;
    // pass 2: split far -> cur (d < T) + compacted far (into the free `next` buffer)
    for (uint32_t i = tid; i < nfar; i += 256) {
      const uint64_t e = far[i];
      const uint32_t dbits = (uint32_t)(e >> 32);
      if (__float_as_uint(ld_f32_l2(&dist[(uint32_t)e])) != dbits) continue;
      if (__uint_as_float(dbits) < T) {
        const uint32_t p = atomicAdd(&ctl->n_cur, 1u);
        cur[p] = e;   // p < nfar <= cap
      } else {
        const uint32_t p = atomicAdd(&ctl->n_far2, 1u);
        next[p] = e;
      }
    }
    __syncthreads();
    if (tid == 0) { ctl->n_far = ctl->n_far2; ctl->n_far2 = 0; }
    uint64_t* t = far; far = next; next = t;
    __syncthreads();
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edf_batch_kernel(kh_label_t* tasks, int mode, const uint32_t* __restrict__ lists,
                                                        const uint32_t* __restrict__ nbrmask, Geometry g, float* field,
                                                        uint64_t* queues, float delta_floor) {
  __shared__ Ctl ctl;
  kh_label_t* task = &tasks[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (mode == 1 && task->root != 0xFFFFFFFFu) return;
  const uint32_t source = (mode == 2) ? task->root : task->source;
  const uint32_t* list = lists + task->list_offset;
  const uint32_t nf = task->count;
  if (tid == 0) ctl.status = 0;
  for (uint32_t i = tid; i < nf; i += 256) st_f32_l2(&field[list[i]], KH_INF);
  __syncthreads();
  Queues q;
  q.cap = task->q_capacity;
  q.a = queues + (uint64_t)task->q_offset * 4;
  q.b = q.a + q.cap;
  q.c = q.b + q.cap;
  q.touched = reinterpret_cast<uint32_t*>(q.c + q.cap);
  sssp<false>(g, nbrmask, nullptr, field, source, q, &ctl, delta_floor);
  // farthest voxel: max finite distance, ties -> smallest linear index
  unsigned long long best = 0;
  for (uint32_t i = tid; i < nf; i += 256) {
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
    for (int i = 1; i < 4; i++) if (ctl.red64[i] > best) best = ctl.red64[i];
    const uint32_t loc = 0xFFFFFFFFu - (uint32_t)best;
    task->max_loc = loc;
    task->max_val = __uint_as_float((uint32_t)(best >> 32));
    if (mode == 1) task->root = loc;
    task->status |= ctl.status;
  }
}

// ------------------------------------------------------------------------------------------------
// exact libstdc++ binary heap (bits/stl_heap.h __push_heap / __adjust_heap / __pop_heap) with the
// non-strict comparator of dijkstra_invalidation.hpp:233-237; run by ONE lane.
struct Heap {
  float* key;
  uint64_t* pay;
  uint32_t n, cap;
};

__device__ __forceinline__ bool heap_push(Heap& h, float k, uint64_t p) {
  if (h.n >= h.cap) return false;
  uint32_t hole = h.n++;
  while (hole > 0) {
    const uint32_t parent = (hole - 1) >> 1;
    const float pk = h.key[parent];
    if (!(pk >= k)) break;
    h.key[hole] = pk;
    h.pay[hole] = h.pay[parent];
    hole = parent;
  }
  h.key[hole] = k;
  h.pay[hole] = p;
  return true;
}

__device__ __forceinline__ void heap_pop(Heap& h) {
  uint32_t len = h.n;
  if (len > 1) {
    len--;
    const float vk = h.key[len];
    const uint64_t vp = h.pay[len];
    uint32_t hole = 0, child = 0;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      float ck = h.key[child];
      const float lk = h.key[child - 1];
      if (ck >= lk) { child--; ck = lk; }
      h.key[hole] = ck;
      h.pay[hole] = h.pay[child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      h.key[hole] = h.key[child - 1];
      h.pay[hole] = h.pay[child - 1];
      hole = child - 1;
    }
    while (hole > 0) {
      const uint32_t parent = (hole - 1) >> 1;
      const float pk = h.key[parent];
      if (!(pk >= vk)) break;
      h.key[hole] = pk;
      h.pay[hole] = h.pay[parent];
      hole = parent;
    }
    h.key[hole] = vk;
    h.pay[hole] = vp;
  }
  h.n--;
}

// wave 0 only.  Returns the number of voxels invalidated.
__device__ uint32_t invalidate_ball(const Geometry& g, const kh_label_t* task, const uint32_t* __restrict__ nbrmask,
                                    const float* __restrict__ dbf, uint8_t* alive, const uint32_t* path, uint32_t npath,
                                    float scale, float constant, float* hkeys, uint64_t* hpay, uint32_t hcap,
                                    uint32_t* status, uint32_t* pushes) {
  const int lane = threadIdx.x & 63;
  Heap h;
  h.key = hkeys; h.pay = hpay; h.n = 0; h.cap = hcap;
  uint32_t npush = 0;
  bool ovf = false;
  if (lane == 0) {
    for (uint32_t i = 0; i < npath; i++) {
      if (!heap_push(h, 0.0f, ((uint64_t)i << 32) | path[i])) ovf = true;
      npush++;
    }
  }
  const uint32_t sx = (uint32_t)g.sx, sxy = (uint32_t)g.sxy;
  const uint32_t xmin = task->xmin, xmax = task->xmax;
  int dx, dy, dz;
  dir_delta(lane < 26 ? lane : 0, dx, dy, dz);
  uint32_t count = 0;
  for (;;) {
    unsigned long long top = NONE64;
    if (lane == 0 && h.n > 0) { top = h.pay[0]; heap_pop(h); }
    top = __shfl(top, 0);
    if (top == NONE64) break;
    const uint32_t vox = (uint32_t)top, si = (uint32_t)(top >> 32);
    if (!alive[vox]) continue;
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
      if (k >= 0 && ((nbrmask[vox] >> k) & 1u)) {
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
    unsigned long long m = __ballot(want);
    while (m) {
      const int k = __ffsll((long long)m) - 1;
      m &= m - 1;
      const float kd = __shfl(nd, k);
      const uint32_t kq = __shfl(q, k);
      if (lane == 0) {
        if (!heap_push(h, kd, ((uint64_t)si << 32) | kq)) ovf = true;
        npush++;
      }
    }
  }
  if (lane == 0) {
    if (ovf) atomicOr(status, KH_ST_HEAP_OVERFLOW);
    *pushes += npush;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  return count;
}

// wave 0 only: canonical predecessor walk (oracle ko_pred / ko_railroad).  Writes the path (rail end
// first) to out[0..]; returns its length (0 on failure).
__device__ uint32_t backtrack(const Geometry& g, const uint32_t* __restrict__ nbrmask, const float* __restrict__ pdrf,
                              const float* dist, uint32_t rail_end, uint32_t target, uint32_t* out, uint32_t cap,
                              uint32_t* status) {
  const int lane = threadIdx.x & 63;
  uint32_t v = rail_end, n = 0;
  if (lane == 0) out[0] = v;
  n = 1;
  while (v != target) {
    const float dv = ld_f32_l2(&dist[v]);
    const float fv = pdrf[v];
    unsigned long long key = NONE64;
    if (lane < 26 && ((nbrmask[v] >> lane) & 1u)) {
      const uint32_t u = v + (uint32_t)g.off[lane];
      const float fu = pdrf[u];
      if (fu != 0.0f) {
        const float du = ld_f32_l2(&dist[u]);
        if (du != KH_INF) {
          const float c = du + fv;
          if (c == dv) key = pack(du, u);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long ok = __shfl_xor(key, o);
      if (ok < key) key = ok;
    }
    if (key == NONE64) { if (lane == 0) atomicOr(status, KH_ST_NO_RAIL); return 0; }
    const uint32_t u = (uint32_t)key;
    const float du = __uint_as_float((uint32_t)(key >> 32));
    if (du >= dv && v != rail_end) { if (lane == 0) atomicOr(status, KH_ST_PLATEAU); return 0; }
    if (n >= cap) { if (lane == 0) atomicOr(status, KH_ST_PATH_OVERFLOW); return 0; }
    if (lane == 0) out[n] = u;
    n++;
    v = u;
  }
  return n;
}

template <typename LT>
__global__ __launch_bounds__(256) void trace_paths_kernel(kh_label_t* tasks, const uint32_t* __restrict__ lists,
                                                          const float* __restrict__ list_daf,
                                                          const uint32_t* __restrict__ nbrmask, Geometry g,
                                                          const float* __restrict__ dbf, float* pdrf, float* dist,
                                                          uint8_t* alive, const uint32_t* __restrict__ manual_targets,
                                                          float scale, float constant, uint64_t* queues, float* heap_keys,
                                                          uint64_t* heap_payload, uint32_t* path_vertices,
                                                          uint32_t* path_lengths) {
  __shared__ Ctl ctl;
  kh_label_t* task = &tasks[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t* list = lists + task->list_offset;
  const float* ldaf = list_daf + task->list_offset;
  const uint32_t nf = task->count;
  Queues q;
  q.cap = task->q_capacity;
  q.a = queues + (uint64_t)task->q_offset * 4;
  q.b = q.a + q.cap;
  q.c = q.b + q.cap;
  q.touched = reinterpret_cast<uint32_t*>(q.c + q.cap);
  uint32_t* pverts = path_vertices + task->path_offset;
  uint32_t* plens = path_lengths + task->path_offset;
  const uint32_t pcap = task->path_capacity;
  const uint32_t root = task->root;
  const uint32_t* before = manual_targets + task->tgt_offset;
  const uint32_t* after = before + task->n_before;
  const bool implicit = task->n_before == 0;            // trace.py:171-172
  uint32_t nb = implicit ? 1u : task->n_before;
  uint32_t na = task->n_after;
  uint32_t valid = nf;                                  // trace.py:211
  const uint32_t max_paths = task->max_paths ? task->max_paths : nf;  // trace.py:214-215
  uint32_t npaths = 0, nverts = 0;
  if (tid == 0) { ctl.status = 0; ctl.u2 = 0; ctl.u3 = 0; }
  __syncthreads();
  if (nb + na >= max_paths) {                           // trace.py:217-218
    if (tid == 0) { task->n_paths = 0; task->n_vertices = 0; }
    return;
  }
  if (tid == 0) pdrf[root] = 0.0f;                      // trace.py:220
  __syncthreads();
  while ((valid > 0 || nb > 0 || na > 0) && npaths < max_paths) {
    // ---- target selection, trace.py:225-230
    uint32_t target;
    if (nb > 0) { nb--; target = implicit ? task->max_loc : before[nb]; }
    else if (valid == 0) { na--; target = after[na]; }
    else {
      // CachedTargetFinder.find_target: the valid voxel with the largest DAF (ties: largest index)
      unsigned long long best = 0;
      for (uint32_t i = tid; i < nf; i += 256) {
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
      for (int i = 1; i < 4; i++) if (ctl.red64[i] > best) best = ctl.red64[i];
      target = (uint32_t)best;
      __syncthreads();
    }
    // ---- railroad, trace.py:240-242
    uint32_t plen = 0;
    if (nverts >= pcap || npaths >= pcap) {
      if (tid == 0) atomicOr(&ctl.status, KH_ST_PATH_OVERFLOW);
      __syncthreads();
      break;
    }
    uint32_t* out = pverts + nverts;
    if (pdrf[target] == 0.0f) {
      if (tid == 0) out[0] = target;
      plen = 1;
    } else {
      sssp<true>(g, nbrmask, pdrf, dist, target, q, &ctl, 0.0f);
      const unsigned long long br = ctl.best_rail;
      if (tid == 0) { ctl.u0 = 0; ctl.u2 += ctl.n_touched; }
      __syncthreads();
      if (br == NONE64) {
        if (tid == 0) atomicOr(&ctl.status, KH_ST_NO_RAIL);
      } else if (wave == 0) {
        const uint32_t n = backtrack(g, nbrmask, pdrf, dist, (uint32_t)br, target, out, pcap - nverts, &ctl.status);
        if (lane == 0) ctl.u0 = n;
      }
      __syncthreads();
      plen = ctl.u0;
      // restore dist = +inf on everything the search touched
      const uint32_t nt = ctl.n_touched < 2u * q.cap ? ctl.n_touched : 2u * q.cap;
      for (uint32_t i = tid; i < nt; i += 256) st_f32_l2(&dist[q.touched[i]], KH_INF);
      __syncthreads();
      if (plen == 0) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    // ---- invalidation, trace.py:253-259
    if (valid > 0) {
      if (wave == 0) {
        const uint32_t c = invalidate_ball(g, task, nbrmask, dbf, alive, out, plen, scale, constant,
                                           heap_keys + task->heap_offset, heap_payload + task->heap_offset,
                                           task->heap_capacity, &ctl.status, &ctl.u3);
        if (lane == 0) ctl.u1 = c;
      }
      __syncthreads();
      valid -= ctl.u1;
    }
    // ---- rails, trace.py:261-263
    for (uint32_t i = tid; i < plen; i += 256) pdrf[out[i]] = 0.0f;
    if (tid == 0) plens[npaths] = plen;
    npaths++;
    nverts += plen;
    __syncthreads();
    if (ctl.status) break;
  }
  __syncthreads();
  if (tid == 0) {
    task->n_paths = npaths;
    task->n_vertices = nverts;
    task->status |= ctl.status;
    task->stat_settled = ctl.u2;
    task->stat_heap_pushes = ctl.u3;
  }
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
                            uint64_t* queues, void* stream) {
  if (int rc = require_device()) return rc;
  if (ntasks <= 0) return KH_OK;
  if (sx * sy * sz >= (1ll << 32)) { set_error("kh_edf_batch: volume must have < 2^32 voxels"); return KH_EINVAL; }
  Geometry g;
  make_geometry(g, sx, sy, sz, wx, wy, wz);
  float mn = wx < wy ? wx : wy;
  if (wz < mn) mn = wz;
  const float delta_floor = 2.0f * mn;
  hipLaunchKernelGGL(edf_batch_kernel, dim3(ntasks), dim3(256), 0, (hipStream_t)stream, tasks, mode, lists, nbrmask, g, field,
                     queues, delta_floor);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_trace_paths(kh_label_t* tasks, int ntasks, const uint32_t* lists, const float* list_daf,
                              const uint32_t* nbrmask, const void* labels, int label_bytes, int64_t sx, int64_t sy,
                              int64_t sz, float wx, float wy, float wz, const float* dbf, float* pdrf, float* dist,
                              uint8_t* alive, const uint32_t* manual_targets, float scale, float constant, uint64_t* queues,
                              float* heap_keys, uint64_t* heap_payload, uint32_t* path_vertices, uint32_t* path_lengths,
                              void* stream) {
  if (int rc = require_device()) return rc;
  if (ntasks <= 0) return KH_OK;
  if (sx * sy * sz >= (1ll << 32)) { set_error("kh_trace_paths: volume must have < 2^32 voxels"); return KH_EINVAL; }
  (void)labels; (void)label_bytes;
  Geometry g;
  make_geometry(g, sx, sy, sz, wx, wy, wz);
  hipLaunchKernelGGL((trace_paths_kernel<uint32_t>), dim3(ntasks), dim3(256), 0, (hipStream_t)stream, tasks, lists, list_daf,
                     nbrmask, g, dbf, pdrf, dist, alive, manual_targets, scale, constant, queues, heap_keys, heap_payload,
                     path_vertices, path_lengths);
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
