/* Developer experiment v5 = v4 + CANDIDATE SPILL: a fifth-plus possible owner of a voxel is not stored in the voxel's word.
 * The pair (voxel, source) goes to a hash set (dedupe), it emits its possible nodes like any pair, and the voxel is marked
 * 'overflowed': it can still be killed for certain by a deadline, but emits no deadlines itself (its owner set is not known
 * to the deadline rule) -- sound: fewer deadlines and a superset of possible nodes.
 * v4 = v3 + GHOSTS: a voxel left in state M at the end of a call is not a failure, it enters the
 * next call as a ghost (f == 3: alive under some resolutions only).  A ghost takes candidates and emits possible nodes like
 * an alive voxel, a deadline kills it for certain, but it emits no deadlines (it may have been dead all along).
 * v3: the order-free sweep of cert_ball2.c restructured the way
 * the HIP kernel runs it: per level  A) candidate closure  B) deadline closure  C) commit  D) emission, with
 * merged possible+deadline events, at most KMAX candidates per voxel, per-voxel state words.
 * See cert_ball2.c for the model and the soundness argument.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifndef KMAX
#define KMAX 8
#endif
typedef struct { uint16_t s[KMAX]; } cw_t;
static int cw_any(const cw_t* w) { for (int i = 0; i < KMAX; i++) if (w->s[i]) return 1; return 0; }
#define EV_P 1u
#define EV_D 2u

typedef struct { float key; uint32_t vox; uint16_t src; uint16_t type; } ev_t;
typedef struct { ev_t* a; size_t n, cap; } pheap;

static void ph_push(pheap* h, ev_t x) {
  if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 4096; h->a = (ev_t*)realloc(h->a, h->cap * sizeof(ev_t)); }
  size_t i = h->n++;
  while (i > 0) { size_t p = (i - 1) / 2; if (!(x.key < h->a[p].key)) break; h->a[i] = h->a[p]; i = p; }
  h->a[i] = x;
}
static ev_t ph_pop(pheap* h) {
  ev_t top = h->a[0]; ev_t x = h->a[--h->n]; size_t i = 0;
  for (;;) { size_t c = 2 * i + 1; if (c >= h->n) break; if (c + 1 < h->n && h->a[c + 1].key < h->a[c].key) c++;
    if (!(h->a[c].key < x.key)) break; h->a[i] = h->a[c]; i = c; }
  if (h->n) h->a[i] = x; return top;
}
static const int8_t D[26][3]={{-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1},{-1,-1,0},{-1,1,0},{1,-1,0},{1,1,0},{0,-1,-1},{0,-1,1},{0,1,-1},{0,1,1},{-1,0,-1},{-1,0,1},{1,0,-1},{1,0,1},{-1,-1,-1},{1,-1,-1},{-1,1,-1},{-1,-1,1},{1,1,-1},{1,-1,1},{-1,1,1},{1,1,1}};
typedef struct { uint64_t* a; size_t n, cap; } vec64;
static void vpush(vec64* v, uint64_t x) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->a = (uint64_t*)realloc(v->a, v->cap * 8); } v->a[v->n++] = x; }

/* cstate word: 4 x u16 slots, slot = (src + 1) | 0x8000 when added in the current level; 0 = empty */
typedef struct { uint64_t* t; size_t cap, n; } hset;
static int hs_add(hset* h, uint64_t key) {   /* 1 = newly added */
  if (h->n * 2 >= h->cap) { size_t oc = h->cap; uint64_t* ot = h->t; h->cap = oc ? oc * 2 : 1024; h->t = (uint64_t*)calloc(h->cap, 8); h->n = 0;
    for (size_t i = 0; i < oc; i++) if (ot[i]) hs_add(h, ot[i] - 1); free(ot); }
  size_t i = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (h->cap - 1);
  for (;;) { if (!h->t[i]) { h->t[i] = key + 1; h->n++; return 1; } if (h->t[i] == key + 1) return 0; i = (i + 1) & (h->cap - 1); }
}
static int cs_count(cw_t w) { int n = 0; for (int i = 0; i < KMAX; i++) if (w.s[i] & 0x7fff) n++; return n; }

/* stats: [0] levels, [1] events processed, [2] bail (0 ok, 1 M left, 2 cand overflow, 3 too many sources), [3] dead,
 * [4] M voxels left, [5] max events in one level, [6] events emitted, [7] peak pending */
int64_t cert_ball6(uint8_t* f, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                   const uint64_t* src, const float* md, int64_t ns, int64_t* stats) {
  const int64_t sxy = sx * sy, nvox = sxy * sz;
  int64_t count = 0;
  memset(stats, 0, 8 * sizeof(int64_t));
  if (ns > 32766) { stats[2] = 3; return -1; }
  int32_t* sox = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)ns);
  cw_t* cst = (cw_t*)calloc((size_t)nvox, sizeof(cw_t));
  cw_t* dbg_own = NULL;
  float* dbg_lvl = NULL;
  vec64 wa = {0,0,0}, wb = {0,0,0}, np = {0,0,0}, dl = {0,0,0};
  pheap h = {0, 0, 0};
  hset ov = {0, 0, 0}; uint8_t* ovf = (uint8_t*)calloc((size_t)nvox, 1); int64_t nspill = 0, novf = 0;
  int64_t nM = 0, ghosts_resolved = 0; int isghost_tmp = 0;
  for (int64_t i = 0; i < ns; i++) {
    int64_t z = src[i] / sxy, r = src[i] % sxy;
    sox[3*i] = (int32_t)(r % sx); sox[3*i+1] = (int32_t)(r / sx); sox[3*i+2] = (int32_t)z;
    int dup = 0;
    for (int64_t j = 0; j < i; j++) if (src[j] == src[i]) { dup = 1; break; }
    if (!dup && f[src[i]]) { ev_t e = {0.0f, (uint32_t)src[i], (uint16_t)i, EV_P | EV_D}; ph_push(&h, e); }
  }
  int bail = 0;
#define DIST(c, nx, ny, nz, out) { float a_ = wx * (float)((nx) - sox[3*(c)]), b_ = wy * (float)((ny) - sox[3*(c)+1]), c_ = wz * (float)((nz) - sox[3*(c)+2]); \
    float s_ = a_ * a_; float t_ = b_ * b_; float u_ = c_ * c_; s_ = s_ + t_; s_ = s_ + u_; out = sqrtf(s_); }
#define NBR_LOOP(v) int64_t z_ = (v) / sxy, r_ = (v) % sxy, y_ = r_ / sx, x_ = r_ % sx; \
      for (int i_ = 0; i_ < 26; i_++) { int64_t nx = x_ + D[i_][0], ny = y_ + D[i_][1], nz = z_ + D[i_][2]; \
        if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue; int64_t q = nx + sx * ny + sxy * nz;
  while (h.n && !bail) {
    const float k = h.a[0].key;
    stats[0]++;
    if ((int64_t)h.n > stats[7]) stats[7] = (int64_t)h.n;
    wa.n = wb.n = np.n = dl.n = 0;
    int64_t nev = 0;
    while (h.n && h.a[0].key == k) {
      ev_t x = ph_pop(&h); nev++;
      if (x.type & EV_P) vpush(&wa, ((uint64_t)x.vox << 32) | x.src);
      if (x.type & EV_D) vpush(&wb, x.vox);
    }
    stats[1] += nev;
    /* A: candidates */
    for (size_t w = 0; w < wa.n && !bail; w++) {
      uint32_t v = (uint32_t)(wa.a[w] >> 32), c = (uint32_t)wa.a[w];
      if (!f[v]) continue;
      cw_t cw = cst[v]; int found = 0, freei = -1;
      for (int i = 0; i < KMAX; i++) { uint32_t s = cw.s[i] & 0x7fff; if (s == c + 1) found = 1; if (!s && freei < 0) freei = i; }
      if (found) continue;
      if (freei < 0) {
        if (!hs_add(&ov, ((uint64_t)v << 32) | c)) continue;
        nspill++; if (!ovf[v]) { ovf[v] = 1; novf++; }
      } else {
        if (!cw_any(&cw)) nM++;
        cst[v].s[freei] = (uint16_t)((c + 1) | 0x8000);
      }
      vpush(&np, ((uint64_t)v << 32) | c);
      NBR_LOOP(v)
        if (!f[q]) continue;
        float d; DIST(c, nx, ny, nz, d);
        if (d < md[c] && d <= k) vpush(&wa, ((uint64_t)q << 32) | c);
      }
    }
    if (bail) break;
    /* B: deadlines */
    for (size_t w = 0; w < wb.n; w++) {
      uint32_t v = (uint32_t)wb.a[w];
      if (!f[v] || (f[v] & 4)) continue;
      cw_t cw = cst[v];
      if (!cw_any(&cw)) { fprintf(stderr, "deadline on untouched voxel\n"); bail = 4; break; }
      int ghost = f[v] == 3;
      f[v] |= 4; vpush(&dl, v);
      if (ghost || ovf[v]) continue;   /* a ghost / an overflowed voxel emits no deadlines */
      NBR_LOOP(v)
        if (!f[q] || (f[q] & 4)) continue;
        int all = 1; float t = 0;
        for (int i = 0; i < KMAX; i++) { uint32_t s = cw.s[i] & 0x7fff; if (!s) continue; float d; DIST(s - 1, nx, ny, nz, d); if (!(d < md[s - 1])) { all = 0; break; } if (d > t) t = d; }
        if (all && t <= k) vpush(&wb, (uint64_t)q);
      }
    }
    if (bail) break;
    /* C: commit */
    for (size_t w = 0; w < dl.n; w++) { uint32_t v = (uint32_t)dl.a[w]; if ((f[v] & 3) == 3) ghosts_resolved++; else count++; isghost_tmp = (f[v] & 3) == 3; f[v] = isghost_tmp ? 8 : 0; nM--; }
    /* D: emission */
    for (size_t w = 0; w < dl.n; w++) {
      uint32_t v = (uint32_t)dl.a[w];
      cw_t cw = cst[v];
      int nc = cs_count(cw);
      if (f[v] == 8 || ovf[v]) continue;   /* ghost / overflowed: no deadlines; its possible nodes go out below */
      NBR_LOOP(v)
        if (!f[q] || f[q] == 8) continue;
        int all = 1; float t = 0; uint32_t one = 0; int isnew = 0;
        for (int i = 0; i < KMAX; i++) { uint32_t s = cw.s[i] & 0x7fff; if (!s) continue; one = s - 1; isnew = (int)((cw.s[i] >> 15) & 1);
          float d; DIST(s - 1, nx, ny, nz, d); if (!(d < md[s - 1])) { all = 0; break; } if (d > t) t = d; }
        if (!all) continue;
        ev_t e = {t, (uint32_t)q, (uint16_t)one, (uint16_t)((nc == 1 && isnew) ? (EV_P | EV_D) : EV_D)};
        ph_push(&h, e); stats[6]++;
      }
    }
    for (size_t w = 0; w < np.n; w++) {
      uint32_t v = (uint32_t)(np.a[w] >> 32), c = (uint32_t)np.a[w];
      cw_t cw = cst[v];
      if (!f[v] && cs_count(cw) == 1 && !ovf[v]) continue;    /* its possible nodes went out merged with the deadlines */
      NBR_LOOP(v)
        if (!f[q] || f[q] == 8) continue;
        float d; DIST(c, nx, ny, nz, d);
        if (!(d < md[c]) || d <= k) continue;
        ev_t e = {d, (uint32_t)q, (uint16_t)c, EV_P};
        ph_push(&h, e); stats[6]++;
      }
    }
    /* E: clean up */
    for (size_t w = 0; w < np.n; w++) { uint32_t v = (uint32_t)(np.a[w] >> 32); if (f[v]) for (int i = 0; i < KMAX; i++) cst[v].s[i] &= 0x7fff; }
    for (size_t w = 0; w < dl.n; w++) if (f[dl.a[w]] == 8) f[dl.a[w]] = 0;
    for (size_t w = 0; w < dl.n; w++) { memset(&cst[dl.a[w]], 0, sizeof(cw_t)); }
  }
  int64_t newghosts = 0;
  if (!bail) for (int64_t v = 0; v < nvox; v++) if (f[v] && cw_any(&cst[v])) { if (f[v] == 1) newghosts++; f[v] = 3; memset(&cst[v], 0, sizeof(cw_t)); }
  stats[4] = newghosts; stats[5] = ghosts_resolved; stats[6] = nspill; stats[7] = novf;
  stats[2] = bail; stats[3] = count;
  free(ov.t); free(ovf); free(dbg_own); free(dbg_lvl); free(h.a); free(cst); free(wa.a); free(wb.a); free(np.a); free(dl.a); free(sox);
  return bail ? -1 : count;
}
