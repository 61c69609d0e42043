/* systolic_heap_sim.c -- CPU model of the lock-step pipelined heap of csrc/trace.hip (round 3), lane for lane.
 *
 * TEST / DESIGN INFRASTRUCTURE (never part of the product).  Three executions of the same flood
 * (dijkstra_invalidation.hpp:239-332) on a random blob, which must agree in the sequence of pops (key, voxel, source,
 * live?) and in the final heap array after every operation boundary that the sequential semantics defines:
 *
 *   A  libstdc++ literally: __push_heap / __adjust_heap as in bits/stl_heap.h with Compare = (a.key >= b.key);
 *   B  the early-stop form (hole walks down the smaller child, ties left, while child.key < last.key) -- what the GPU
 *      used in rounds 1-2, one operation after the other;
 *   C  the pipeline: lane l owns the hole of one in-flight pop at heap level l; every tick all holes move down one level
 *      (read phase, then write phase); a new pop enters at the root when the previous one is two levels down; pushes of a
 *      live pop wait only until no in-flight hole is an ancestor of the new leaves.  Hazard rules as in the HIP code:
 *        I1  injection needs lane 1 empty (RAW through the heap array between consecutive pops);
 *        I2  an operation touching slot set S (a pop's `last` slot, the new leaves of a push batch with their ancestor
 *            chains) starts only when no in-flight hole is an ancestor-or-self of a slot of S;
 *        I3  the prefetched `last` element is dropped when a tick wrote its slot.
 *
 *   gcc -O2 -o /tmp/systolic_heap_sim experiments/systolic_heap_sim.c -lm && /tmp/systolic_heap_sim 200
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t k, v, s, m; } node;   /* key bits, voxel, source voxel, max_dist bits */
#define NONE 0xFFFFFFFFu
#define INFB 0x7f800000u

static int SX, SY, SZ;
static float WX, WY, WZ;
static uint8_t* MASK0;

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* the pushes of a live pop, in direction order (shared by A, B, C: the heap is what is under test) */
static int fire(const uint8_t* alive, uint32_t vox, uint32_t src, uint32_t maxdb, node* out) {
  const int sxy = SX * SY;
  const int z = vox / sxy, r = vox % sxy, y = r / SX, x = r % SX;
  const int oz = src / sxy, orr = src % sxy, oy = orr / SX, ox = orr % SX;
  const float maxd = bitsf(maxdb);
  int n = 0;
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        if (!dx && !dy && !dz) continue;
        const int qx = x + dx, qy = y + dy, qz = z + dz;
        if (qx < 0 || qy < 0 || qz < 0 || qx >= SX || qy >= SY || qz >= SZ) continue;
        const uint32_t q = (uint32_t)(qx + SX * qy + sxy * qz);
        if (!alive[q]) continue;
        const float a = WX * (float)(qx - ox), b = WY * (float)(qy - oy), c = WZ * (float)(qz - oz);
        float s = a * a; const float t = b * b, u = c * c;
        s = s + t; s = s + u;
        const float nd = sqrtf(s);
        if (nd < maxd) { out[n].k = fbits(nd); out[n].v = q; out[n].s = src; out[n].m = maxdb; n++; }
      }
  return n;
}

/* ---------------------------------------------------------------- A: libstdc++ */
static void std_push(node* h, uint32_t* n, node x) {
  uint32_t hole = (*n)++;
  while (hole > 0) {
    const uint32_t par = (hole - 1) / 2;
    if (!(h[par].k >= x.k)) break;       /* comp(parent, value) */
    h[hole] = h[par];
    hole = par;
  }
  h[hole] = x;
}
static void std_pop(node* h, uint32_t* n) {
  const uint32_t len = --(*n);
  if (len == 0) return;
  const node value = h[len];
  uint32_t hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (h[child].k >= h[child - 1].k) child--;   /* comp(first + child, first + (child - 1)) */
    h[hole] = h[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    h[hole] = h[child - 1];
    hole = child - 1;
  }
  while (hole > 0) {                              /* __push_heap(first, hole, 0, value) */
    const uint32_t par = (hole - 1) / 2;
    if (!(h[par].k >= value.k)) break;
    h[hole] = h[par];
    hole = par;
  }
  h[hole] = value;
}

/* ---------------------------------------------------------------- B: early stop */
static void es_pop(node* h, uint32_t* n) {
  const uint32_t len = --(*n);
  if (len == 0) return;
  const node value = h[len];
  uint32_t hole = 0;
  for (;;) {
    const uint32_t c0 = 2 * hole + 1;
    if (c0 >= len) break;
    uint32_t c = c0;
    if (c0 + 1 < len && h[c0 + 1].k < h[c0].k) c = c0 + 1;
    if (h[c].k >= value.k) break;
    h[hole] = h[c];
    hole = c;
  }
  h[hole] = value;
}

typedef struct { uint32_t k, v, s; int live; } rec;

static long run_seq(int early, const uint8_t* mask, const node* srcs, int nsrc, uint8_t* alive, rec* log, long logcap, node* h) {
  memcpy(alive, mask, (size_t)SX * SY * SZ);
  uint32_t n = 0;
  long nl = 0;
  for (int i = 0; i < nsrc; i++) std_push(h, &n, srcs[i]);
  node out[26];
  while (n) {
    const node top = h[0];
    if (early) es_pop(h, &n); else std_pop(h, &n);
    const int live = alive[top.v];
    if (nl < logcap) { log[nl].k = top.k; log[nl].v = top.v; log[nl].s = top.s; log[nl].live = live; }
    nl++;
    if (!live) continue;
    alive[top.v] = 0;
    const int m = fire(alive, top.v, top.s, top.m, out);
    for (int i = 0; i < m; i++) std_push(h, &n, out[i]);
  }
  return nl;
}

/* ---------------------------------------------------------------- C: the pipeline, lane for lane */
#define NL 64
static int level_of(uint32_t slot) { int l = 0; while (((slot + 1) >> (l + 1)) != 0) l++; return l; }

/* does the hole `pos` of lane `l` (level l) lie on the ancestor chain (or at) a slot of [a, b]?  n >= 1 assumed */
static int on_chain(uint32_t pos, int l, uint32_t a, uint32_t b) {
  const int la = level_of(a), lb = level_of(b);
  for (int L = la; L <= lb; L++) {
    uint32_t lo = (1u << L) - 1, hi = (2u << L) - 2;
    if (lo < a) lo = a;
    if (hi > b) hi = b;
    if (L < l) continue;
    const uint32_t alo = (lo + 1) >> (L - l), ahi = (hi + 1) >> (L - l);
    if (pos + 1 >= alo && pos + 1 <= ahi) return 1;
  }
  return 0;
}

static long ticks_total, pops_total, stall_inject, stall_fire;

static long run_pipe(const uint8_t* mask, const node* srcs, int nsrc, uint8_t* alive, rec* log, long logcap, node* h) {
  memcpy(alive, mask, (size_t)SX * SY * SZ);
  uint32_t n = 0;
  long nl = 0;
  for (int i = 0; i < nsrc; i++) std_push(h, &n, srcs[i]);
  uint32_t pos[NL], plen[NL];
  node val[NL];
  for (int l = 0; l < NL; l++) pos[l] = NONE;
  /* prefetched last element: slot lslot, valid unless a tick wrote it since the load */
  node lastv; uint32_t lslot = NONE; int lvalid = 0;
  int firing = 0; node fout[26]; int fcnt = 0;
  for (;;) {
    int any = 0;
    for (int l = 0; l < NL; l++) any |= pos[l] != NONE;
    int injected = 0;
    if (!firing) {
      if (n == 0) { if (!any) break; }
      else if (pos[1] == NONE && pos[0] == NONE) {               /* I1 */
        const uint32_t s = n - 1;
        int conflict = 0;
        for (int l = 0; l < NL; l++) if (pos[l] != NONE && on_chain(pos[l], l, s, s)) conflict = 1;   /* I2 on the last slot */
        if (!conflict) {
          if (!(lvalid && lslot == s)) { lastv = h[s]; lslot = s; lvalid = 1; }    /* (re)load: exposed latency on the GPU */
          const node top = h[0];
          const int live = alive[top.v];
          if (nl < logcap) { log[nl].k = top.k; log[nl].v = top.v; log[nl].s = top.s; log[nl].live = live; }
          nl++; pops_total++;
          n = s;
          if (s > 0) { pos[0] = 0; plen[0] = s; val[0] = lastv; injected = 1; }
          /* prefetch the next last element */
          if (s > 0) { lastv = h[s - 1]; lslot = s - 1; lvalid = 1; } else lvalid = 0;
          if (live) {
            alive[top.v] = 0;
            fcnt = fire(alive, top.v, top.s, top.m, fout);
            firing = fcnt > 0;
          }
        } else stall_inject++;
      }
    } else {
      int conflict = 0;
      if (n < 64) { conflict = any; }                                /* tiny heap: drain */
      else for (int l = 0; l < NL; l++) if (pos[l] != NONE && on_chain(pos[l], l, n, n + (uint32_t)fcnt - 1)) conflict = 1;
      if (!conflict) {
        for (int i = 0; i < fcnt; i++) std_push(h, &n, fout[i]);
        firing = 0; lvalid = 0;
      } else stall_fire++;
    }
    /* ---- tick: read phase for every lane, then write phase */
    any = 0;
    for (int l = 0; l < NL; l++) any |= pos[l] != NONE;
    if (!any) continue;
    ticks_total++;
    node W[NL]; uint32_t np[NL];
    for (int l = 0; l < NL; l++) {
      np[l] = NONE;
      if (pos[l] == NONE) continue;
      const uint32_t c0 = 2 * pos[l] + 1;
      const int hasL = c0 < plen[l], hasR = c0 + 1 < plen[l];
      const node L = hasL ? h[c0] : (node){INFB, 0, 0, 0}, R = hasR ? h[c0 + 1] : (node){INFB, 0, 0, 0};
      const int pickR = R.k < L.k;
      const node P = pickR ? R : L;
      const int stop = !hasL || P.k >= val[l].k;
      W[l] = stop ? val[l] : P;
      np[l] = stop ? NONE : c0 + (uint32_t)pickR;
    }
    for (int l = 0; l < NL; l++) {
      if (pos[l] == NONE) continue;
      h[pos[l]] = W[l];
      if (lvalid && pos[l] == lslot) lvalid = 0;                     /* I3 */
    }
    for (int l = NL - 1; l >= 1; l--) { pos[l] = np[l - 1]; plen[l] = plen[l - 1]; val[l] = val[l - 1]; }
    pos[0] = NONE;
    (void)injected;
  }
  return nl;
}


/* ---------------------------------------------------------------- C2: as C, but a pop whose `last` slot is still on the
 * ancestor chain of an older in-flight hole enters WITHOUT its value ("val-less") instead of waiting at the door:
 *   - the value is fetched in the first tick in which no OLDER hole is an ancestor-or-self of its slot (then final);
 *   - until then its stop test cannot be evaluated; it is replaced by a floor: every candidate for the value -- the
 *     slot's current content or the value of an older in-flight pop that may still land there -- has a key >= kfl, so
 *     P.key < kfl means "continue" whatever the value turns out to be; otherwise the pop FREEZES together with everything
 *     younger (lanes below it) for this tick, older pops go on and resolve the question. */
static long freezes, valless;
static long run_pipe2(const uint8_t* mask, const node* srcs, int nsrc, uint8_t* alive, rec* log, long logcap, node* h) {
  memcpy(alive, mask, (size_t)SX * SY * SZ);
  uint32_t n = 0;
  long nl = 0;
  for (int i = 0; i < nsrc; i++) std_push(h, &n, srcs[i]);
  uint32_t pos[NL], plen[NL], vslot[NL], kfl[NL];
  int hasval[NL];
  node val[NL];
  for (int l = 0; l < NL; l++) pos[l] = NONE;
  int firing = 0; node fout[26]; int fcnt = 0;
  uint32_t ep_cur = 0xFFFFFFFFu, ep_prev = 0xFFFFFFFFu; int ep_count = 0;
  for (;;) {
    int any = 0;
    for (int l = 0; l < NL; l++) any |= pos[l] != NONE;
    if (!firing) {
      if (n == 0) { if (!any) break; }
      else if (pos[1] == NONE && pos[0] == NONE) {               /* I1 */
        const uint32_t s = n - 1;
        const node top = h[0];
        const int live = alive[top.v];
        if (nl < logcap) { log[nl].k = top.k; log[nl].v = top.v; log[nl].s = top.s; log[nl].live = live; }
        nl++; pops_total++;
        n = s;
        if (s > 0) {
          pos[0] = 0; plen[0] = s; vslot[0] = s; hasval[0] = 0;
          /* floor over the candidates: the slot's current content or the value of an older pop in flight -- which is
           * itself the content its slot had when it was injected, or an even older pop's value.  So the minimum key of the
           * slots' contents over (at least) the last 32 injections bounds every candidate from below: two scalar
           * accumulators that take turns every 32 injections (at most 32 / 2 = 16 pops are in flight). */
          if (getenv("EXACTFLOOR")) {
            uint32_t f = h[s].k;
            for (int l = 1; l < NL; l++) if (pos[l] != NONE) { const uint32_t k = hasval[l] ? val[l].k : kfl[l]; if (k < f) f = k; }
            kfl[0] = f;
          } else {
            if (++ep_count == 32) { ep_count = 0; ep_prev = ep_cur; ep_cur = 0xFFFFFFFFu; }
            if (h[s].k < ep_cur) ep_cur = h[s].k;
            kfl[0] = ep_cur < ep_prev ? ep_cur : ep_prev;
          }
        }
        if (live) {
          alive[top.v] = 0;
          fcnt = fire(alive, top.v, top.s, top.m, fout);
          firing = fcnt > 0;
        }
      }
    } else {
      int conflict = 0;
      if (n < 64) { conflict = any; }
      else for (int l = 0; l < NL; l++) if (pos[l] != NONE && (!hasval[l] || on_chain(pos[l], l, n, n + (uint32_t)fcnt - 1))) conflict = 1;
      /* (a val-less pop has not read its slot yet, and that slot is >= n: the new leaves would overwrite it) */
      if (!conflict) { for (int i = 0; i < fcnt; i++) std_push(h, &n, fout[i]); firing = 0; }
      else stall_fire++;
    }
    any = 0;
    for (int l = 0; l < NL; l++) any |= pos[l] != NONE;
    if (!any) continue;
    ticks_total++;
    /* value fetch for val-less lanes whose slot no OLDER hole can reach any more */
    for (int l = 0; l < NL; l++) {
      if (pos[l] == NONE || hasval[l]) continue;
      int conflict = 0;
      for (int o = l + 1; o < NL; o++) if (pos[o] != NONE && on_chain(pos[o], o, vslot[l], vslot[l])) conflict = 1;
      if (!conflict) { val[l] = h[vslot[l]]; hasval[l] = 1; } else valless++;
    }
    node W[NL]; uint32_t np[NL]; int frz = -1;
    for (int l = 0; l < NL; l++) {
      np[l] = NONE;
      if (pos[l] == NONE) continue;
      const uint32_t c0 = 2 * pos[l] + 1;
      const int hasL = c0 < plen[l], hasR = c0 + 1 < plen[l];
      const node L = hasL ? h[c0] : (node){INFB, 0, 0, 0}, R = hasR ? h[c0 + 1] : (node){INFB, 0, 0, 0};
      const int pickR = R.k < L.k;
      const node P = pickR ? R : L;
      if (!hasval[l]) {
        if (!hasL || P.k >= kfl[l]) { if (l > frz) frz = l; continue; }   /* cannot decide: freeze */
        W[l] = P; np[l] = c0 + (uint32_t)pickR;
        continue;
      }
      const int stop = !hasL || P.k >= val[l].k;
      W[l] = stop ? val[l] : P;
      np[l] = stop ? NONE : c0 + (uint32_t)pickR;
    }
    if (frz >= 0) freezes++;
    for (int l = frz + 1; l < NL; l++) if (pos[l] != NONE) h[pos[l]] = W[l];
    for (int l = NL - 1; l >= 1; l--) {
      if (l > frz + 1) { pos[l] = np[l - 1]; plen[l] = plen[l - 1]; val[l] = val[l - 1]; vslot[l] = vslot[l - 1]; hasval[l] = hasval[l - 1]; kfl[l] = kfl[l - 1]; }
      else if (l == frz + 1) pos[l] = NONE;
      /* l <= frz: keeps its own state */
    }
    if (frz < 0) pos[0] = NONE;
  }
  return nl;
}

int main(int argc, char** argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 50;
  srand(12345);
  long bad = 0;
  for (int t = 0; t < trials; t++) {
    SX = 10 + rand() % 40; SY = 10 + rand() % 30; SZ = 6 + rand() % 30;
    const int BIG = getenv("BIG") != NULL;       /* one deep heap (~10^5 nodes) instead of many shallow ones */
    if (BIG) { SX = 120; SY = 100; SZ = 60; }
    const int mode = t % 4;
    WX = mode == 0 ? 1 : mode == 1 ? 16 : mode == 2 ? 2 : 4; WY = mode == 0 ? 1 : mode == 1 ? 16 : mode == 2 ? 3 : 4; WZ = mode == 0 ? 1 : mode == 1 ? 40 : mode == 2 ? 5 : 40;
    const size_t nv = (size_t)SX * SY * SZ;
    uint8_t* mask = malloc(nv);
    /* a blobby object: union of random balls (many exact ties with integer anisotropies) */
    memset(mask, 0, nv);
    int cx = SX / 2, cy = SY / 2, cz = SZ / 2;
    if (BIG) memset(mask, 1, nv);
    for (int b = 0; b < 30; b++) {
      cx += rand() % 7 - 3; cy += rand() % 7 - 3; cz += rand() % 5 - 2;
      if (cx < 1) cx = 1; if (cy < 1) cy = 1; if (cz < 1) cz = 1;
      if (cx > SX - 2) cx = SX - 2; if (cy > SY - 2) cy = SY - 2; if (cz > SZ - 2) cz = SZ - 2;
      const int r = 2 + rand() % 5;
      for (int z = cz - r; z <= cz + r; z++) for (int y = cy - r; y <= cy + r; y++) for (int x = cx - r; x <= cx + r; x++) {
        if (x < 0 || y < 0 || z < 0 || x >= SX || y >= SY || z >= SZ) continue;
        if ((x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz) <= r * r) mask[x + SX * (y + SY * z)] = 1;
      }
    }
    /* sources: a short run of object voxels with radii that make balls overlap */
    node srcs[40]; int nsrc = 0;
    const int want = 1 + rand() % 24;
    for (size_t i = rand() % nv, tries = 0; nsrc < want && tries < 4 * nv; i = (i + 1 + rand() % 3) % nv, tries++)
      if (mask[i]) { const float r = (float)((BIG ? 25 : 2) + rand() % 12) * WX + (rand() % 2 ? 0.5f : 0.0f);
                     srcs[nsrc++] = (node){0u, (uint32_t)i, (uint32_t)i, fbits(r)}; }
    uint8_t *a1 = malloc(nv), *a2 = malloc(nv), *a3 = malloc(nv);
    const long cap = 4000000;
    rec *l1 = malloc(sizeof(rec) * cap), *l2 = malloc(sizeof(rec) * cap), *l3 = malloc(sizeof(rec) * cap);
    node* h = malloc(sizeof(node) * (27 * nv + 64));
    const long n1 = run_seq(0, mask, srcs, nsrc, a1, l1, cap, h);
    const long n2 = run_seq(1, mask, srcs, nsrc, a2, l2, cap, h);
    const long n3 = getenv("V1") ? run_pipe(mask, srcs, nsrc, a3, l3, cap, h) : run_pipe2(mask, srcs, nsrc, a3, l3, cap, h);
    int ok = n1 == n2 && n1 == n3 && !memcmp(a1, a2, nv) && !memcmp(a1, a3, nv);
    const long m = n1 < cap ? n1 : cap;
    if (ok) for (long i = 0; i < m; i++) {
      if (memcmp(&l1[i], &l2[i], sizeof(rec)) || memcmp(&l1[i], &l3[i], sizeof(rec))) { ok = 0; printf("trial %d: pop %ld differs\n", t, i); break; }
    }
    if (!ok) { bad++; printf("trial %d FAILED (%ld / %ld / %ld pops)\n", t, n1, n2, n3); }
    free(mask); free(a1); free(a2); free(a3); free(l1); free(l2); free(l3); free(h);
    MASK0 = NULL;
  }
  printf("freezes %ld, val-less lane-ticks %ld\n", freezes, valless);
  printf("%d trials, %ld failures; pops %ld, ticks %ld (%.2f per pop), injection stalls %ld, push stalls %ld\n", trials, bad, pops_total,
         ticks_total, (double)ticks_total / (double)(pops_total ? pops_total : 1), stall_inject, stall_fire);
  return bad != 0;
}
