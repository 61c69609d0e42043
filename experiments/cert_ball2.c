/* Developer experiment v2 (test infrastructure, CPU): order-free evaluation of
 * roll_invalidation_ball_inside_component with three voxel states and "deadlines".
 *
 * Model P of dijkstra_invalidation.hpp:239-332: a multiset of nodes (key, source, voxel); pop ANY node of
 * minimal key; if its voxel is alive, kill it (owner = node's source) and push every alive in-radius
 * neighbour with key = distance to the owner.  libstdc++'s heap is one resolution of "ANY".
 *
 * Abstract sweep over the levels k (distinct keys, increasing):
 *   state A (alive under every order), M (may be dead), D (dead under every order).
 *   possible node (key, c, m): under some order source c may own m from level max(key, level of emission) on.
 *     Processing it on a voxel that is not yet D adds c to Cand(m) (A -> M) and emits the possible nodes of
 *     (m, c) to the neighbours c covers (cascade when key <= k).
 *   deadline (t, m): m is dead under every order once level t is complete.  Emitted when a voxel v becomes D at
 *     level tv, for every neighbour m covered by ALL of Cand(v): t = max(tv, max_c d_c(m)).  Sources start with
 *     a deadline at level 0.
 * Certified iff no voxel is left in state M at the end; then the dead set is D under every order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define KMAX 8

typedef struct { float key; uint32_t src; uint32_t vox; } pnode;   /* src == 0xffffffff: deadline */
typedef struct { pnode* a; size_t n, cap; } pheap;

static void ph_push(pheap* h, pnode x) {
  if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 4096; h->a = (pnode*)realloc(h->a, h->cap * sizeof(pnode)); }
  size_t i = h->n++;
  while (i > 0) { size_t p = (i - 1) / 2; if (!(x.key < h->a[p].key)) break; h->a[i] = h->a[p]; i = p; }
  h->a[i] = x;
}
static pnode ph_pop(pheap* h) {
  pnode top = h->a[0]; pnode x = h->a[--h->n]; size_t i = 0;
  for (;;) { size_t c = 2 * i + 1; if (c >= h->n) break; if (c + 1 < h->n && h->a[c + 1].key < h->a[c].key) c++;
    if (!(h->a[c].key < x.key)) break; h->a[i] = h->a[c]; i = c; }
  if (h->n) h->a[i] = x; return top;
}

static const int8_t D[26][3]={{-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1},{-1,-1,0},{-1,1,0},{1,-1,0},{1,1,0},{0,-1,-1},{0,-1,1},{0,1,-1},{0,1,1},{-1,0,-1},{-1,0,1},{1,0,-1},{1,0,1},{-1,-1,-1},{1,-1,-1},{-1,1,-1},{-1,-1,1},{1,1,-1},{1,-1,1},{-1,1,1},{1,1,1}};

typedef struct { uint32_t vox; uint8_t ncand; uint8_t dnow; uint32_t cand[KMAX]; } tslot;
typedef struct { uint64_t* a; size_t n, cap; } vec64;
static void vpush(vec64* v, uint64_t x) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->a = (uint64_t*)realloc(v->a, v->cap * 8); } v->a[v->n++] = x; }

/* stats: [0] levels, [1] possible nodes processed, [2] bail (0 ok, 1 M left, 2 cand overflow), [3] dead,
 * [4] M voxels left, [5] voxels that had >1 candidate, [6] deadlines processed, [7] peak M voxels */
int64_t cert_ball2(uint8_t* f, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                   const uint64_t* src, const float* md, int64_t ns, int64_t* stats, float delta) {
  const int64_t sxy = sx * sy, nvox = sxy * sz;
  int64_t count = 0;
  memset(stats, 0, 8 * sizeof(int64_t));
  uint32_t* canon = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)ns);
  int32_t* sox = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)ns);
  for (int64_t i = 0; i < ns; i++) {
    canon[i] = (uint32_t)i;
    for (int64_t j = 0; j < i; j++) if (src[j] == src[i]) { canon[i] = (uint32_t)j; break; }
    int64_t z = src[i] / sxy, r = src[i] % sxy;
    sox[3*i] = (int32_t)(r % sx); sox[3*i+1] = (int32_t)(r / sx); sox[3*i+2] = (int32_t)z;
  }
  int32_t* slot = (int32_t*)malloc(sizeof(int32_t) * (size_t)nvox);
  memset(slot, 0xff, sizeof(int32_t) * (size_t)nvox);
  tslot* T = NULL; size_t tn = 0, tcap = 0;
  int32_t* freel = NULL; size_t fn = 0, fcap = 0;
  vec64 wl = {0,0,0}, dl = {0,0,0};
  pheap h = {0, 0, 0};
  int64_t nM = 0;
  for (int64_t i = 0; i < ns; i++) if (canon[i] == (uint32_t)i && f[src[i]]) {
    pnode x = {0.0f, (uint32_t)i, (uint32_t)src[i]}; ph_push(&h, x);
    pnode y = {0.0f, 0xffffffffu, (uint32_t)src[i]}; ph_push(&h, y);
  }
  int bail = 0;
#define DIST(c, nx, ny, nz, out) { float a_ = wx * (float)((nx) - sox[3*(c)]), b_ = wy * (float)((ny) - sox[3*(c)+1]), c_ = wz * (float)((nz) - sox[3*(c)+2]); \
    float s_ = a_ * a_; float t_ = b_ * b_; float u_ = c_ * c_; s_ = s_ + t_; s_ = s_ + u_; out = sqrtf(s_); }
#define QK(d) (delta > 0.0f ? (floorf((d) / delta) + 1.0f) * delta : (d))
#define GETSLOT(q, s2) { s2 = slot[q]; if (s2 < 0) { if (fn) s2 = freel[--fn]; else { if (tn == tcap) { tcap = tcap ? tcap * 2 : 1024; T = (tslot*)realloc(T, tcap * sizeof(tslot)); } s2 = (int32_t)tn++; } \
      slot[q] = s2; T[s2].vox = (uint32_t)(q); T[s2].ncand = 0; T[s2].dnow = 0; nM++; if (nM > stats[7]) stats[7] = nM; } }
  while (h.n && !bail) {
    const float k = h.a[0].key;
    stats[0]++;
    wl.n = 0; dl.n = 0;
    while (h.n && h.a[0].key == k) {
      pnode x = ph_pop(&h);
      if (!f[x.vox]) continue;
      if (x.src == 0xffffffffu) vpush(&dl, x.vox); else vpush(&wl, ((uint64_t)x.vox << 32) | x.src);
    }
    /* phase A: possible nodes of this level, cascades included */
    for (size_t w = 0; w < wl.n && !bail; w++) {
      uint32_t v = (uint32_t)(wl.a[w] >> 32), c = (uint32_t)wl.a[w];
      stats[1]++;
      int32_t s; GETSLOT(v, s);
      int found = 0;
      for (int q = 0; q < T[s].ncand; q++) if (T[s].cand[q] == c) found = 1;
      if (found) continue;
      if (T[s].ncand == KMAX) { bail = 2; break; }
      T[s].cand[T[s].ncand++] = c;
      if (T[s].ncand == 2) stats[5]++;
      int64_t z = v / sxy, r = v % sxy, y = r / sx, x = r % sx;
      for (int i = 0; i < 26; i++) {
        int64_t nx = x + D[i][0], ny = y + D[i][1], nz = z + D[i][2];
        if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
        int64_t q = nx + sx * ny + sxy * nz;
        if (!f[q]) continue;
        float d; DIST(c, nx, ny, nz, d);
        if (!(d < md[c])) continue;
        float qd = QK(d); if (qd <= k) vpush(&wl, ((uint64_t)q << 32) | c);
        else { pnode nn = {qd, c, (uint32_t)q}; ph_push(&h, nn); }
      }
    }
    if (bail) break;
    /* phase B: deadlines of this level, cascades included */
    for (size_t w = 0; w < dl.n; w++) {
      uint32_t v = (uint32_t)dl.a[w];
      int32_t s = slot[v];
      if (s < 0) { fprintf(stderr, "deadline on untouched voxel?!\n"); bail = 3; break; }
      if (T[s].dnow) continue;
      T[s].dnow = 1; stats[6]++;
      int64_t z = v / sxy, r = v % sxy, y = r / sx, x = r % sx;
      for (int i = 0; i < 26; i++) {
        int64_t nx = x + D[i][0], ny = y + D[i][1], nz = z + D[i][2];
        if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
        int64_t q = nx + sx * ny + sxy * nz;
        if (!f[q]) continue;
        float t = k; int all = 1;
        for (int qc = 0; qc < T[s].ncand; qc++) { uint32_t c = T[s].cand[qc]; float d; DIST(c, nx, ny, nz, d); if (!(d < md[c])) { all = 0; break; } float qd = QK(d); if (qd > t) t = qd; }
        if (!all) continue;
        if (t <= k) { int32_t s2 = slot[q]; if (s2 >= 0 && T[s2].dnow) continue; vpush(&dl, (uint64_t)q); }
        else { pnode nn = {t, 0xffffffffu, (uint32_t)q}; ph_push(&h, nn); }
      }
    }
    if (bail) break;
    /* commit */
    for (size_t w = 0; w < dl.n; w++) {
      uint32_t v = (uint32_t)dl.a[w];
      int32_t s = slot[v];
      if (s < 0) continue;   /* duplicate entry already committed */
      f[v] = 0; count++; slot[v] = -1; nM--;
      if (fn == fcap) { fcap = fcap ? fcap * 2 : 1024; freel = (int32_t*)realloc(freel, fcap * sizeof(int32_t)); }
      freel[fn++] = s;
    }
  }
  if (!bail && nM > 0) bail = 1;
  stats[2] = bail; stats[3] = count; stats[4] = nM;
  free(h.a); free(T); free(wl.a); free(dl.a); free(slot); free(canon); free(sox); free(freel);
  return bail ? -1 : count;
}
