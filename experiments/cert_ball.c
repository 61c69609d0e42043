/* Developer experiment (test infrastructure, CPU): a level-synchronous, order-free evaluation of
 * roll_invalidation_ball_inside_component with a SOUND certificate that the result does not depend
 * on how the heap breaks ties between equal keys.
 *
 * Model P of dijkstra_invalidation.hpp:239-332: a multiset of nodes (key, source, voxel); pop ANY node of
 * minimal key; if its voxel is alive, kill it (owner = node's source) and push every alive in-radius
 * neighbour with key = distance to the owner.  libstdc++'s heap is one resolution of "ANY".
 *
 * Abstract execution: levels = strictly increasing running maximum k of the popped keys.  In a level every
 * pending node with key <= k is processed, cascades (pushes with key <= k) included.  Per voxel touched in
 * the level: Cand = every source that can own it under some order (over-approximation); a node is DEFINITE
 * when it exists under every order (pushed by a voxel whose Cand is a singleton).  The level is certified
 * when every touched voxel is killed by a definite node; then the set of dead voxels after the level is the
 * same under every order.  Otherwise the call "bails" (the exact heap emulation has to be used).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#define KMAX 6


typedef struct { float key; uint32_t src; uint32_t vox; uint32_t def; uint32_t from; uint32_t fromc[KMAX]; uint32_t fromn; } pnode;
typedef struct { pnode* a; size_t n, cap; } pheap;

static int ph_push(pheap* h, pnode x) {
  if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 4096; h->a = (pnode*)realloc(h->a, h->cap * sizeof(pnode)); if (!h->a) return 1; }
  size_t i = h->n++;
  while (i > 0) { size_t p = (i - 1) / 2; if (!(x.key < h->a[p].key)) break; h->a[i] = h->a[p]; i = p; }
  h->a[i] = x; return 0;
}
static pnode ph_pop(pheap* h) {
  pnode top = h->a[0]; pnode x = h->a[--h->n]; size_t i = 0;
  for (;;) { size_t c = 2 * i + 1; if (c >= h->n) break; if (c + 1 < h->n && h->a[c + 1].key < h->a[c].key) c++;
    if (!(h->a[c].key < x.key)) break; h->a[i] = h->a[c]; i = c; }
  if (h->n) h->a[i] = x; return top;
}

static const int8_t D[26][3]={{-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1},{-1,-1,0},{-1,1,0},{1,-1,0},{1,1,0},{0,-1,-1},{0,-1,1},{0,1,-1},{0,1,1},{-1,0,-1},{-1,0,1},{1,0,-1},{1,0,1},{-1,-1,-1},{1,-1,-1},{-1,1,-1},{-1,-1,1},{1,1,-1},{1,-1,1},{-1,1,1},{1,1,1}};

typedef struct { uint32_t vox; uint8_t ncand; uint8_t def; uint32_t cand[KMAX]; } tslot;

/* stats[0]=levels, [1]=nodes processed, [2]=bail reason (0 ok,1 no definite,2 cand overflow), [3]=dead at bail,
 * [4]=max touched per level, [5]=ambiguous voxels, [6]=possible-only nodes pushed, [7]=bail level index */
int64_t cert_ball(uint8_t* f, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                  const uint64_t* src, const float* md, int64_t ns, int64_t* stats) {
  const int64_t sxy = sx * sy, nvox = sxy * sz;
  int64_t count = 0;
  memset(stats, 0, 8 * sizeof(int64_t));
  /* canonical source ids: identical voxel => identical (position, radius) => same candidate */
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
  uint64_t* wl = NULL; size_t wn = 0, wcap = 0;   /* (slot << 32) | cand */
  pheap h = {0, 0, 0};
  for (int64_t i = 0; i < ns; i++) if (canon[i] == (uint32_t)i) { pnode x = {0.0f, (uint32_t)i, (uint32_t)src[i], 1, 0, {0}, 0}; ph_push(&h, x); }
  int bail = 0;
#define DIST(c, nx, ny, nz, out) { float a_ = wx * (float)((nx) - sox[3*(c)]), b_ = wy * (float)((ny) - sox[3*(c)+1]), c_ = wz * (float)((nz) - sox[3*(c)+2]); \
    float s_ = a_ * a_; float t_ = b_ * b_; float u_ = c_ * c_; s_ = s_ + t_; s_ = s_ + u_; out = sqrtf(s_); }
  while (h.n && !bail) {
    const float k = h.a[0].key;
    stats[0]++;
    tn = 0; wn = 0;
    /* extract the level */
    while (h.n && h.a[0].key == k) {
      pnode x = ph_pop(&h);
      stats[1]++;
      if (!f[x.vox]) continue;
      int32_t s = slot[x.vox];
      if (s < 0) {
        if (tn == tcap) { tcap = tcap ? tcap * 2 : 1024; T = (tslot*)realloc(T, tcap * sizeof(tslot)); }
        s = (int32_t)tn++; slot[x.vox] = s; T[s].vox = x.vox; T[s].ncand = 0; T[s].def = 0;
      }
      if (x.def) T[s].def = 1;
      if (!x.def && getenv("CERT_DEBUG2")) { int64_t z = x.from / sxy, r = x.from % sxy; int64_t z2 = x.vox / sxy, r2 = x.vox % sxy;
        fprintf(stderr, "   possible node k=%g src=%u vox=(%ld,%ld,%ld) from (%ld,%ld,%ld) cands:", x.key, x.src, (long)(r2%sx),(long)(r2/sx),(long)z2, (long)(r%sx),(long)(r/sx),(long)z);
        for (uint32_t q = 0; q < x.fromn; q++) { float dd; DIST(x.fromc[q], (r%sx), (r/sx), z, dd); float dn; DIST(x.fromc[q], (r2%sx), (r2/sx), z2, dn); fprintf(stderr, " [%u d=%g r=%g -> dn=%g]", x.fromc[q], dd, md[x.fromc[q]], dn); } fprintf(stderr, "\n"); }
      int found = 0;
      for (int q = 0; q < T[s].ncand; q++) if (T[s].cand[q] == x.src) found = 1;
      if (!found) {
        if (T[s].ncand == KMAX) { bail = 2; break; }
        T[s].cand[T[s].ncand++] = x.src;
        if (wn == wcap) { wcap = wcap ? wcap * 2 : 1024; wl = (uint64_t*)realloc(wl, wcap * sizeof(uint64_t)); }
        wl[wn++] = ((uint64_t)s << 32) | x.src;
      }
    }
    if (bail) break;
    /* phase A: candidate closure over cascades (pushes with key <= k) */
    for (size_t w = 0; w < wn && !bail; w++) {
      int32_t s = (int32_t)(wl[w] >> 32); uint32_t c = (uint32_t)wl[w];
      uint32_t v = T[s].vox;
      int64_t z = v / sxy, r = v % sxy, y = r / sx, x = r % sx;
      for (int i = 0; i < 26; i++) {
        int64_t nx = x + D[i][0], ny = y + D[i][1], nz = z + D[i][2];
        if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
        int64_t q = nx + sx * ny + sxy * nz;
        if (!f[q]) continue;
        float d; DIST(c, nx, ny, nz, d);
        if (!(d < md[c]) || d > k) continue;
        int32_t s2 = slot[q];
        if (s2 < 0) {
          if (tn == tcap) { tcap = tcap ? tcap * 2 : 1024; T = (tslot*)realloc(T, tcap * sizeof(tslot)); }
          s2 = (int32_t)tn++; slot[q] = s2; T[s2].vox = (uint32_t)q; T[s2].ncand = 0; T[s2].def = 0;
        }
        int found = 0;
        for (int qq = 0; qq < T[s2].ncand; qq++) if (T[s2].cand[qq] == c) found = 1;
        if (!found) {
          if (T[s2].ncand == KMAX) { bail = 2; break; }
          T[s2].cand[T[s2].ncand++] = c;
          if (wn == wcap) { wcap = wcap ? wcap * 2 : 1024; wl = (uint64_t*)realloc(wl, wcap * sizeof(uint64_t)); }
          wl[wn++] = ((uint64_t)s2 << 32) | c;
        }
        stats[1]++;
      }
    }
    if (bail) break;
    /* phase B: definite deaths: seeded by definite pending nodes, propagated through cascades of singleton owners */
    {
      wn = 0;
      for (size_t s = 0; s < tn; s++) if (T[s].def) { if (wn == wcap) { wcap = wcap ? wcap * 2 : 1024; wl = (uint64_t*)realloc(wl, wcap * sizeof(uint64_t)); } wl[wn++] = s; }
      for (size_t w = 0; w < wn; w++) {
        int32_t s = (int32_t)wl[w];
        if (T[s].ncand != 1) continue;
        uint32_t c = T[s].cand[0];
        uint32_t v = T[s].vox;
        int64_t z = v / sxy, r = v % sxy, y = r / sx, x = r % sx;
        for (int i = 0; i < 26; i++) {
          int64_t nx = x + D[i][0], ny = y + D[i][1], nz = z + D[i][2];
          if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
          int64_t q = nx + sx * ny + sxy * nz;
          if (!f[q]) continue;
          float d; DIST(c, nx, ny, nz, d);
          if (!(d < md[c]) || d > k) continue;
          int32_t s2 = slot[q];
          if (!T[s2].def) { T[s2].def = 1; if (wn == wcap) { wcap = wcap ? wcap * 2 : 1024; wl = (uint64_t*)realloc(wl, wcap * sizeof(uint64_t)); } wl[wn++] = (uint64_t)s2; }
        }
      }
      for (size_t s = 0; s < tn; s++) if (!T[s].def) { bail = 1;
        if (getenv("CERT_DEBUG")) { uint32_t v = T[s].vox; int64_t z = v / sxy, r = v % sxy, y = r / sx, x = r % sx;
          fprintf(stderr, "  undecided voxel (%ld,%ld,%ld) level k=%g ncand=%d:", (long)x,(long)y,(long)z, k, T[s].ncand);
          for (int q = 0; q < T[s].ncand; q++) { float d; DIST(T[s].cand[q], x, y, z, d); fprintf(stderr, " [src %u d=%g r=%g]", T[s].cand[q], d, md[T[s].cand[q]]); }
          fprintf(stderr, "\n"); } }
    }
    if ((int64_t)tn > stats[4]) stats[4] = (int64_t)tn;
    if (bail) { for (size_t s = 0; s < tn; s++) slot[T[s].vox] = -1; break; }
    /* commit */
    for (size_t s = 0; s < tn; s++) { f[T[s].vox] = 0; count++; if (T[s].ncand > 1) stats[5]++; }
    for (size_t s = 0; s < tn; s++) {
      uint32_t v = T[s].vox;
      int64_t z = v / sxy, r = v % sxy, y = r / sx, x = r % sx;
      for (int qc = 0; qc < T[s].ncand; qc++) {
        uint32_t c = T[s].cand[qc];
        for (int i = 0; i < 26; i++) {
          int64_t nx = x + D[i][0], ny = y + D[i][1], nz = z + D[i][2];
          if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
          int64_t q = nx + sx * ny + sxy * nz;
          if (!f[q]) continue;
          float d; DIST(c, nx, ny, nz, d);
          if (!(d < md[c])) continue;
          pnode nn = {d, c, (uint32_t)q, (uint32_t)(T[s].ncand == 1), v, {0}, T[s].ncand}; memcpy(nn.fromc, T[s].cand, sizeof(nn.fromc));
          if (!nn.def) stats[6]++;
          ph_push(&h, nn);
        }
      }
      slot[v] = -1;
    }
  }
  stats[2] = bail; stats[3] = count; stats[7] = stats[0];
  free(h.a); free(T); free(wl); free(slot); free(canon); free(sox);
  return bail ? -1 : count;
}
