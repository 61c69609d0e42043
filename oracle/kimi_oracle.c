/*
 * oracle/kimi_oracle.c  --  CPU restatement of the kimimaro TEASAR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kimimaro_amd/ (the product) may
 * import, link or execute this file; it is the checker used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Where the arithmetic lives in a third-party package whose
 * source is NOT in the reference tree (edt>=3.0.0, dijkstra3d>=1.15.0, see
 * pyproject.toml:60-62) the published algorithm is restated and the parity
 * status is "unpinned by reference code" -- pinned instead by the mathematical
 * definition (brute force / scipy) and by the reference's own end-to-end known
 * answer tests (automated_test.py:48-199).  Functions that follow
 * ext/skeletontricks are pinned against the compiled reference
 * (oracle/_ref, tests/test_oracle_vs_ref.py) and the golden vectors in
 * tests/golden/.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC   (see oracle/build.py)
 * All arrays are Fortran ordered: loc = x + sx*(y + sy*z)   (skeletontricks.pyx:398).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KO_OK 0
#define KO_EINVAL 1
#define KO_ENOMEM 2
#define KO_ENOPATH 3
#define KO_EPLATEAU 4

/* ------------------------------------------------------------------------- */
/* 26-neighbourhood, in the order of dijkstra_invalidation.hpp:60-124:
 * 0 -x, 1 +x, 2 -y, 3 +y, 4 -z, 5 +z, 6..9 xy diagonals (-x-y,-x+y,+x-y,+x+y),
 * 10..13 yz (-y-z,-y+z,+y-z,+y+z), 14..17 xz (-x-z,-x+z,+x-z,+x+z),
 * 18..25 corners (---, +--, -+-, --+, ++-, +-+, -++, +++).                    */
static const int8_t KO_DIR[26][3] = {
  {-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1},
  {-1,-1,0},{-1,1,0},{1,-1,0},{1,1,0},
  {0,-1,-1},{0,-1,1},{0,1,-1},{0,1,1},
  {-1,0,-1},{-1,0,1},{1,0,-1},{1,0,1},
  {-1,-1,-1},{1,-1,-1},{-1,1,-1},{-1,-1,1},{1,1,-1},{1,-1,1},{-1,1,1},{1,1,1}
};

/* voxel_graph= of the dijkstra3d calls (kimimaro/trace.py:139-145,155,240-242; third-party source absent, PARITY UNPINNED): a
 * uint32 per voxel, bit ko_vg_bit[i] = "a step from THIS voxel in direction i is allowed" (the layout cc3d writes and
 * dijkstra_invalidation.hpp:152-190 reads).  Restated the way the invalidation reads it: the word of the voxel being EXPANDED
 * gates the step, so an asymmetric graph gives one-way edges; a predecessor u of v needs the edge u -> v.  Set for the
 * duration of a call by the Python wrappers (test infrastructure: single threaded). */
static const int ko_vg_bit[26] = {1, 0, 3, 2, 5, 4,   9, 7, 8, 6,   17, 13, 16, 12,   15, 11, 14, 10,
                                  25, 24, 23, 21, 22, 20, 19, 18};
static const uint32_t* ko_vg = 0;
void ko_set_voxel_graph(const uint32_t* g) { ko_vg = g; }
static inline int ko_edge(uint64_t u, int dir) { return ko_vg == 0 || ((ko_vg[u] >> ko_vg_bit[dir]) & 1u); }
static int ko_opp(int dir) {   /* the direction with the negated offset */
  for (int j = 0; j < 26; j++)
    if (KO_DIR[j][0] == -KO_DIR[dir][0] && KO_DIR[j][1] == -KO_DIR[dir][1] && KO_DIR[j][2] == -KO_DIR[dir][2]) return j;
  return dir;
}


/* centre-to-centre edge lengths, float arithmetic, no contraction
 * (dijkstra_invalidation.hpp:45-52 `_s`, `_c`). */
void ko_weights26(float wx, float wy, float wz, float* w) {
  for (int i = 0; i < 26; i++) {
    float a = KO_DIR[i][0] ? wx : 0.0f;
    float b = KO_DIR[i][1] ? wy : 0.0f;
    float c = KO_DIR[i][2] ? wz : 0.0f;
    float s = a * a;
    float t = b * b;
    float u = c * c;
    s = s + t;
    s = s + u;
    w[i] = sqrtf(s);
  }
}

/* ------------------------------------------------------------------------- */
/* a1: multi-label anisotropic exact EDT.
 * Reference call sites: kimimaro/intake.py:178-183, kimimaro/trace.py:112-117.
 * Arithmetic lives in PyPI `edt` (>=3.0.0, source absent): restated from its
 * published algorithm: x pass = distance to the nearest
 * label change inside the row (w*k, squared), y/z pass = min over j in the
 * same-label segment of f[j] + (w*(i-j))^2, clamped by the segment ends
 * (label change, or volume border when black_border), sqrt last.
 * The min is taken EXACTLY over the float expressions (windowed search, the
 * window closes as soon as (w*k)^2 >= best), so the result is the unique value
 * defined by the formula -- the HIP kernel evaluates the same expressions.   */
static inline uint64_t ko_lab(const void* p, int bytes, int64_t i) {
  switch (bytes) {
    case 1: return ((const uint8_t*)p)[i];
    case 2: return ((const uint16_t*)p)[i];
    case 4: return ((const uint32_t*)p)[i];
    default: return ((const uint64_t*)p)[i];
  }
}

/* instrumentation (DESIGN.md sizing): search steps taken by the last ko_edt call, per axis pass */
int64_t ko_edt_steps[2] = {0, 0};
static int ko_edt_axis_id = 0;
int64_t ko_get_edt_steps(int pass) { return ko_edt_steps[pass & 1]; }

static void ko_edt_axis(const void* labels, int lb, float* f, int64_t n, int64_t stride,
                        int64_t base, float w, int black_border, float* tmp) {
  /* one line: positions base + i*stride, i in [0,n) ; tmp holds the input f */
  for (int64_t i = 0; i < n; i++) tmp[i] = f[base + i * stride];
  for (int64_t i = 0; i < n; i++) {
    uint64_t L = ko_lab(labels, lb, base + i * stride);
    if (L == 0) { f[base + i * stride] = 0.0f; continue; }
    float best = tmp[i];
    int left_open = 1, right_open = 1;
    for (int64_t k = 1; left_open || right_open; k++) {
      float d = w * (float)k;
      float t = d * d;
      if (t >= best) break;
      ko_edt_steps[ko_edt_axis_id]++;
      if (left_open) {
        int64_t j = i - k;
        if (j < 0) { left_open = 0; if (black_border && t < best) best = t; }
        else if (ko_lab(labels, lb, base + j * stride) != L) { left_open = 0; if (t < best) best = t; }
        else { float c = tmp[j] + t; if (c < best) best = c; }
      }
      if (right_open) {
        int64_t j = i + k;
        if (j >= n) { right_open = 0; if (black_border && t < best) best = t; }
        else if (ko_lab(labels, lb, base + j * stride) != L) { right_open = 0; if (t < best) best = t; }
        else { float c = tmp[j] + t; if (c < best) best = c; }
      }
    }
    f[base + i * stride] = best;
  }
}

int ko_edt_nd(const void* labels, int label_bytes, int ndim, int64_t sx, int64_t sy, int64_t sz,
              float wx, float wy, float wz, int black_border, float* out);
int ko_edt(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz,
           float wx, float wy, float wz, int black_border, float* out) {
  return ko_edt_nd(labels, label_bytes, 3, sx, sy, sz, wx, wy, wz, black_border, out);
}
/* ndim = dimensionality of the caller's array: edt.edt on a 2-D array is a 2-D transform, a missing axis has no
 * border (kimimaro/intake.py:568 calls it on the faces of the volume with black_border=True).               */
int ko_edt_nd(const void* labels, int label_bytes, int ndim, int64_t sx, int64_t sy, int64_t sz,
              float wx, float wy, float wz, int black_border, float* out) {
  if (!(label_bytes == 1 || label_bytes == 2 || label_bytes == 4 || label_bytes == 8)) return KO_EINVAL;
  const int64_t sxy = sx * sy;
  /* x pass: two sweeps */
  for (int64_t z = 0; z < sz; z++) for (int64_t y = 0; y < sy; y++) {
    const int64_t b = sx * y + sxy * z;
    /* left sweep: distance (in voxels) to nearest differing voxel on the left */
    int64_t run = 0; int have = black_border; uint64_t prev = 0;
    for (int64_t x = 0; x < sx; x++) {
      uint64_t L = ko_lab(labels, label_bytes, b + x);
      if (x > 0 && L != prev) { run = 0; have = 1; }
      run++; prev = L;
      if (L == 0) out[b + x] = 0.0f;
      else if (have) { float d = wx * (float)run; out[b + x] = d * d; }
      else out[b + x] = INFINITY;
    }
    run = 0; have = black_border; prev = 0;
    for (int64_t x = sx - 1; x >= 0; x--) {
      uint64_t L = ko_lab(labels, label_bytes, b + x);
      if (x < sx - 1 && L != prev) { run = 0; have = 1; }
      run++; prev = L;
      if (L != 0 && have) { float d = wx * (float)run; float t = d * d; if (t < out[b + x]) out[b + x] = t; }
    }
  }
  int64_t m = sy > sz ? sy : sz;
  float* tmp = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
  if (!tmp) return KO_ENOMEM;
  ko_edt_steps[0] = ko_edt_steps[1] = 0;
  ko_edt_axis_id = 0;
  if (ndim >= 2 && (sy > 1 || black_border))
    for (int64_t z = 0; z < sz; z++) for (int64_t x = 0; x < sx; x++)
      ko_edt_axis(labels, label_bytes, out, sy, sx, x + sxy * z, wy, black_border, tmp);
  ko_edt_axis_id = 1;
  if (ndim >= 3 && (sz > 1 || black_border))
    for (int64_t y = 0; y < sy; y++) for (int64_t x = 0; x < sx; x++)
      ko_edt_axis(labels, label_bytes, out, sz, sxy, x + sx * y, wz, black_border, tmp);
  free(tmp);
  const int64_t n = sxy * sz;
  for (int64_t i = 0; i < n; i++) out[i] = sqrtf(out[i]);
  return KO_OK;
}

/* ------------------------------------------------------------------------- */
/* generic binary min-heap on (key, loc) -- a TOTAL order, so the settle order
 * (and therefore every result below) is independent of heap internals.       */
typedef struct { float k; uint64_t v; } ko_hn;
typedef struct { ko_hn* a; size_t n, cap; } ko_heap;

static inline int ko_hless(ko_hn x, ko_hn y) { return x.k < y.k || (x.k == y.k && x.v < y.v); }
static int ko_hpush(ko_heap* h, float k, uint64_t v) {
  if (h->n == h->cap) {
    size_t nc = h->cap ? h->cap * 2 : 1024;
    ko_hn* na = (ko_hn*)realloc(h->a, nc * sizeof(ko_hn));
    if (!na) return KO_ENOMEM;
    h->a = na; h->cap = nc;
  }
  size_t i = h->n++;
  ko_hn x = {k, v};
  while (i > 0) { size_t p = (i - 1) / 2; if (!ko_hless(x, h->a[p])) break; h->a[i] = h->a[p]; i = p; }
  h->a[i] = x;
  return KO_OK;
}
static ko_hn ko_hpop(ko_heap* h) {
  ko_hn top = h->a[0];
  ko_hn x = h->a[--h->n];
  size_t i = 0;
  for (;;) {
    size_t c = 2 * i + 1;
    if (c >= h->n) break;
    if (c + 1 < h->n && ko_hless(h->a[c + 1], h->a[c])) c++;
    if (!ko_hless(h->a[c], x)) break;
    h->a[i] = h->a[c]; i = c;
  }
  if (h->n) h->a[i] = x;
  return top;
}

/* ------------------------------------------------------------------------- */
/* a4: dijkstra3d.euclidean_distance_field (call sites kimimaro/trace.py:139-145,
 * 302-307).  Third-party source absent: restated as the Bellman fixpoint
 *     d[src] = 0 ; d[v] = min_{u in N26(v), mask[u]} fl(d[u] + w(u,v))
 * over the 26-connected foreground with float32 accumulation, w = anisotropic
 * centre-to-centre distance.  The fixpoint is unique (fl(a+w) is monotone), so
 * any correct search gives these bits.  Background/unreachable = +inf.
 * max location: largest finite d, ties -> smallest linear index (documented
 * canonical tie-break; the reference's is heap-order dependent).
 * free_space_radius (soma mode, trace.py:134) is not restated here (row f3).  */
int ko_edf(const uint8_t* mask, int64_t sx, int64_t sy, int64_t sz,
           float wx, float wy, float wz, uint64_t source, float free_space_radius,
           float* out, uint64_t* max_loc, float* max_val) {
  const int64_t sxy = sx * sy, n = sxy * sz;
  float w26[26];
  ko_weights26(wx, wy, wz, w26);
  for (int64_t i = 0; i < n; i++) out[i] = INFINITY;
  if ((int64_t)source >= n || !mask[source]) return KO_EINVAL;
  ko_heap h = {0, 0, 0};
  out[source] = 0.0f;
  if (ko_hpush(&h, 0.0f, source)) return KO_ENOMEM;
  /* free_space_radius (soma mode, trace.py:134,142): dijkstra3d source absent; restated from its
   * documentation: foreground voxels closer than the radius (straight line, anisotropic) get the straight
   * line distance and all of them seed the search.  PARITY UNPINNED. */
  if (free_space_radius > 0.0f) {
    const int64_t sz0 = (int64_t)(source / (uint64_t)sxy), sr = (int64_t)(source % (uint64_t)sxy), sy0 = sr / sx, sx0 = sr % sx;
    for (int64_t z = 0; z < sz; z++) for (int64_t y = 0; y < sy; y++) for (int64_t x = 0; x < sx; x++) {
      const int64_t q = x + sx * y + sxy * z;
      if (!mask[q] || (uint64_t)q == source) continue;
      float a = wx * (float)(x - sx0), b = wy * (float)(y - sy0), c = wz * (float)(z - sz0);
      float s2 = a * a; float t2 = b * b; float u2 = c * c;
      s2 = s2 + t2; s2 = s2 + u2;
      const float sd = sqrtf(s2);
      if (sd < free_space_radius) { out[q] = sd; if (ko_hpush(&h, sd, (uint64_t)q)) { free(h.a); return KO_ENOMEM; } }
    }
  }
  while (h.n) {
    ko_hn t = ko_hpop(&h);
    if (t.k > out[t.v]) continue; /* stale */
    int64_t z = (int64_t)(t.v / (uint64_t)sxy), r = (int64_t)(t.v % (uint64_t)sxy), y = r / sx, x = r % sx;
    for (int i = 0; i < 26; i++) {
      int64_t nx = x + KO_DIR[i][0], ny = y + KO_DIR[i][1], nz = z + KO_DIR[i][2];
      if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
      int64_t q = nx + sx * ny + sxy * nz;
      if (!mask[q] || !ko_edge(t.v, i)) continue;
      float nd = t.k + w26[i];
      if (nd < out[q]) { out[q] = nd; if (ko_hpush(&h, nd, (uint64_t)q)) { free(h.a); return KO_ENOMEM; } }
    }
  }
  free(h.a);
  float best = -1.0f; uint64_t bl = source;
  for (int64_t i = 0; i < n; i++) if (mask[i] && out[i] != INFINITY && out[i] > best) { best = out[i]; bl = (uint64_t)i; }
  if (max_loc) *max_loc = bl;
  if (max_val) *max_val = best;
  return KO_OK;
}

/* ------------------------------------------------------------------------- */
/* a5: compute_pdrf, kimimaro/trace.py:315-356.  Every operation is rounded to
 * float32 exactly as numpy 2.x does (SURVEY B-9): P = 1 - DBF*M ; P = P^e by
 * repeated squaring when e is a power of two < 2^16 (:343-345) else powf ;
 * P *= scale ; if max_daf != 0: DAF *= (1/max_daf) ; P += DAF.  DAF IS MUTATED
 * (:353).  M = f32(1/dbf_max**1.01) is computed by the caller with numpy.     */
int ko_pdrf(const float* dbf, float* daf, int64_t n, float M, int exponent,
            float scale, float max_daf, float* out) {
  int pow2 = exponent > 0 && (exponent & (exponent - 1)) == 0 && exponent < 65536;
  int nsq = 0; for (int e = exponent; pow2 && e > 1; e >>= 1) nsq++;
  float inv = 0.0f;
  if (max_daf != 0.0f) inv = 1.0f / max_daf;
  for (int64_t i = 0; i < n; i++) {
    float p = dbf[i] * M;
    p = 1.0f - p;
    if (pow2) { for (int s = 0; s < nsq; s++) p = p * p; }
    else p = powf(p, (float)exponent);
    p = p * scale;
    if (max_daf != 0.0f) { float d = daf[i] * inv; daf[i] = d; p = p + d; }
    out[i] = p;
  }
  return KO_OK;
}

/* ------------------------------------------------------------------------- */
/* a6: CachedTargetFinder.__init__, skeletontricks.pyx:1001-1006:
 *   idx = flatnonzero(mask F-order) ; order = flip(argsort(daf[idx])).
 * numpy's default argsort is unstable, so the tie order in the reference is
 * unspecified; canonical restatement = flip(stable ascending argsort), i.e.
 * descending DAF, ties by DESCENDING linear index.  Returns count.            */
typedef struct { float k; uint32_t v; } ko_kv;
static int ko_kv_cmp(const void* a, const void* b) {
  const ko_kv* x = (const ko_kv*)a; const ko_kv* y = (const ko_kv*)b;
  if (x->k > y->k) return -1;
  if (x->k < y->k) return 1;
  return (x->v > y->v) ? -1 : (x->v < y->v);
}
int64_t ko_target_order(const uint8_t* mask, const float* daf, int64_t n, uint32_t* order) {
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) if (mask[i]) m++;
  ko_kv* a = (ko_kv*)malloc(sizeof(ko_kv) * (size_t)(m ? m : 1));
  if (!a) return -1;
  m = 0;
  for (int64_t i = 0; i < n; i++) if (mask[i]) { a[m].k = daf[i]; a[m].v = (uint32_t)i; m++; }
  qsort(a, (size_t)m, sizeof(ko_kv), ko_kv_cmp);
  for (int64_t i = 0; i < m; i++) order[i] = a[i].v;
  free(a);
  return m;
}

/* ------------------------------------------------------------------------- */
/* a7/a8: weighted 26-connected shortest paths over a voxel field
 * (dijkstra3d.parental_field / path_from_parents / railroad; call sites
 * kimimaro/trace.py:155, 240-244).  Third-party source absent; restated:
 *   cost of a step = value of the voxel being ENTERED, float32 accumulation,
 *   +inf voxels are walls.
 * Distances are the unique Bellman fixpoint
 *     d[src] = 0 ; d[v] = min_u fl(d[u] + f[v]).
 * Path choice under ties is heap-order dependent in the reference; canonical
 * restatement (identical in the HIP kernels):
 *     pred(v) = the neighbour u with fl(d[u] + f[v]) == d[v] minimising
 *               (d[u], linear index of u)
 * and a path is the pred chain.  Rails (f == 0, trace.py:220,263) are
 * absorbing: they receive a distance but are not expanded, so the search ends
 * at the first rail, which is the rail voxel with the smallest distance
 * (ties -> smallest linear index).                                            */
static int ko_pred(const float* f, const float* d, int64_t sx, int64_t sy, int64_t sz,
                   uint64_t v, int rails_absorb, uint64_t* pred) {
  const int64_t sxy = sx * sy;
  int64_t z = (int64_t)(v / (uint64_t)sxy), r = (int64_t)(v % (uint64_t)sxy), y = r / sx, x = r % sx;
  float dv = d[v], fv = f[v];
  int found = 0; float bd = 0; uint64_t bu = 0;
  for (int i = 0; i < 26; i++) {
    int64_t nx = x + KO_DIR[i][0], ny = y + KO_DIR[i][1], nz = z + KO_DIR[i][2];
    if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
    uint64_t u = (uint64_t)(nx + sx * ny + sxy * nz);
    float du = d[u];
    if (du == INFINITY || !ko_edge(u, ko_opp(i))) continue;
    if (rails_absorb && f[u] == 0.0f) continue; /* rails never expand; the source is never a rail here */
    float c = du + fv;
    if (c != dv) continue;
    if (!found || du < bd || (du == bd && u < bu)) { found = 1; bd = du; bu = u; }
  }
  if (!found) return KO_ENOPATH;
  *pred = bu;
  return KO_OK;
}

/* Dijkstra from `source` over field f.  stop_at_rail: rails absorb and the
 * search stops once the heap minimum exceeds the best rail distance.          */
static int ko_field_sssp(const float* f, int64_t sx, int64_t sy, int64_t sz, uint64_t source,
                         int stop_at_rail, float* d, uint64_t* rail_end, int64_t* settled) {
  const int64_t sxy = sx * sy, n = sxy * sz;
  for (int64_t i = 0; i < n; i++) d[i] = INFINITY;
  ko_heap h = {0, 0, 0};
  d[source] = 0.0f;
  if (ko_hpush(&h, 0.0f, source)) return KO_ENOMEM;
  float best_rail = INFINITY; uint64_t best_loc = 0; int have_rail = 0;
  int64_t nset = 0;
  while (h.n) {
    if (stop_at_rail && have_rail && h.a[0].k > best_rail) break;
    ko_hn t = ko_hpop(&h);
    if (t.k > d[t.v]) continue;
    nset++;
    if (stop_at_rail && f[t.v] == 0.0f && t.v != source) {
      if (!have_rail || t.k < best_rail || (t.k == best_rail && t.v < best_loc)) { have_rail = 1; best_rail = t.k; best_loc = t.v; }
      continue; /* absorbing */
    }
    int64_t z = (int64_t)(t.v / (uint64_t)sxy), r = (int64_t)(t.v % (uint64_t)sxy), y = r / sx, x = r % sx;
    for (int i = 0; i < 26; i++) {
      int64_t nx = x + KO_DIR[i][0], ny = y + KO_DIR[i][1], nz = z + KO_DIR[i][2];
      if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
      int64_t q = nx + sx * ny + sxy * nz;
      float fq = f[q];
      if (fq == INFINITY || !ko_edge(t.v, i)) continue;
      float nd = t.k + fq;
      if (nd < d[q]) { d[q] = nd; if (ko_hpush(&h, nd, (uint64_t)q)) { free(h.a); return KO_ENOMEM; } }
    }
  }
  free(h.a);
  if (settled) *settled = nset;
  if (stop_at_rail) { if (!have_rail) return KO_ENOPATH; *rail_end = best_loc; }
  return KO_OK;
}

/* Predecessor walk from `start` back to the search source `stop` (canonical, identical in the HIP kernel):
 *   - normally: pred(v) = achieving neighbour (fl(d[u]+f[v]) == d[v]) with d[u] < d[v], minimising (d[u], index);
 *   - at the rail end of a railroad (f == 0): every achieving neighbour has d[u] == d[v]; take the smallest index;
 *   - float-absorption plateau (a step cost below half an ulp of the accumulated distance: every achieving
 *     neighbour has d[u] == d[v]): breadth-first search over the equal-distance achieving neighbours
 *     (FIFO, neighbours in direction order 0..25) to the first voxel that has a strictly smaller achieving
 *     predecessor; the BFS route is followed.  A real Dijkstra resolves such plateaus by its heap order. */
static int ko_pred_strict(const float* f, const float* d, int64_t sx, int64_t sy, int64_t sz,
                          uint64_t v, int rails_absorb, int allow_equal, uint64_t* pred) {
  const int64_t sxy = sx * sy;
  int64_t z = (int64_t)(v / (uint64_t)sxy), r = (int64_t)(v % (uint64_t)sxy), y = r / sx, x = r % sx;
  const float dv = d[v], fv = f[v];
  int found = 0; float bd = 0; uint64_t bu = 0;
  for (int i = 0; i < 26; i++) {
    int64_t nx = x + KO_DIR[i][0], ny = y + KO_DIR[i][1], nz = z + KO_DIR[i][2];
    if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
    uint64_t u = (uint64_t)(nx + sx * ny + sxy * nz);
    float du = d[u];
    if (du == INFINITY || !ko_edge(u, ko_opp(i))) continue;
    if (rails_absorb && f[u] == 0.0f) continue;
    if (du + fv != dv) continue;
    if (!allow_equal && !(du < dv)) continue;
    if (!found || du < bd || (du == bd && u < bu)) { found = 1; bd = du; bu = u; }
  }
  if (found) *pred = bu;
  return found;
}

int64_t ko_plateau_count = 0;  /* instrumentation: plateau searches since the last reset */
int64_t ko_get_plateau_count(int reset) { int64_t v = ko_plateau_count; if (reset) ko_plateau_count = 0; return v; }

static int ko_walk(const float* f, const float* d, int64_t sx, int64_t sy, int64_t sz, uint64_t start, uint64_t stop,
                   int rails_absorb, uint64_t* path, int64_t* npath) {
  const int64_t sxy = sx * sy, n = sxy * sz;
  int64_t k = 0; uint64_t v = start;
  path[k++] = v;
  uint8_t* seen = 0; uint64_t* q = 0; int64_t* par = 0;
  int rc = KO_OK;
  while (v != stop) {
    uint64_t u;
    if (k >= n) { rc = KO_ENOPATH; break; }
    if (rails_absorb && v == start) {  /* the rail end: equal distances are the rule */
      if (!ko_pred_strict(f, d, sx, sy, sz, v, 1, 1, &u)) { rc = KO_ENOPATH; break; }
      path[k++] = u; v = u; continue;
    }
    if (ko_pred_strict(f, d, sx, sy, sz, v, rails_absorb, 0, &u)) { path[k++] = u; v = u; continue; }
    /* plateau */
    ko_plateau_count++;
    if (!seen) {
      seen = (uint8_t*)calloc((size_t)n, 1); q = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
      par = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
      if (!seen || !q || !par) { rc = KO_ENOMEM; break; }
    }
    int64_t head = 0, tail = 0, found = -1;
    q[tail] = v; par[tail] = -1; tail++; seen[v] = 1;
    while (head < tail) {
      const uint64_t x = q[head];
      uint64_t tmp;
      if (head > 0 && ko_pred_strict(f, d, sx, sy, sz, x, rails_absorb, 0, &tmp)) { found = head; break; }
      int64_t xz = (int64_t)(x / (uint64_t)sxy), xr = (int64_t)(x % (uint64_t)sxy), xy = xr / sx, xx = xr % sx;
      for (int i = 0; i < 26; i++) {
        int64_t nx = xx + KO_DIR[i][0], ny = xy + KO_DIR[i][1], nz = xz + KO_DIR[i][2];
        if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy || nz >= sz) continue;
        uint64_t w = (uint64_t)(nx + sx * ny + sxy * nz);
        if (seen[w] || d[w] != d[x] || !ko_edge(w, ko_opp(i))) continue;
        if (rails_absorb && f[w] == 0.0f) continue;
        if (d[w] + f[x] != d[x]) continue;
        seen[w] = 1; q[tail] = w; par[tail] = head; tail++;
      }
      head++;
    }
    for (int64_t i = 0; i < tail; i++) seen[q[i]] = 0;
    if (found < 0) { rc = KO_ENOPATH; break; }
    /* route v -> ... -> q[found]: collect backwards, append forwards */
    int64_t len = 0;
    for (int64_t i = found; i > 0; i = par[i]) len++;
    if (k + len > n) { rc = KO_ENOPATH; break; }
    int64_t pos = k + len - 1;
    for (int64_t i = found; i > 0; i = par[i]) path[pos--] = q[i];
    k += len;
    v = q[found];
  }
  free(seen); free(q); free(par);
  *npath = k;
  return rc;
}

/* a8: dijkstra3d.railroad(field, source=target) (trace.py:240-242).
 * Returns the path ordered rail end -> ... -> target (trace.py:249-250 keeps
 * path[:1] as the junction).  If the target itself is a rail the path is the
 * single vertex.  d (scratch, n floats) is caller provided.                   */
int ko_railroad(const float* f, int64_t sx, int64_t sy, int64_t sz, uint64_t target,
                float* d, uint64_t* path, int64_t* npath, int64_t* settled) {
  const int64_t n = sx * sy * sz;
  if ((int64_t)target >= n || f[target] == INFINITY) return KO_EINVAL;
  if (f[target] == 0.0f) { path[0] = target; *npath = 1; if (settled) *settled = 1; return KO_OK; }
  uint64_t e = 0;
  int rc = ko_field_sssp(f, sx, sy, sz, target, 1, d, &e, settled);
  if (rc) return rc;
  return ko_walk(f, d, sx, sy, sz, e, target, 1, path, npath);
}

/* fix_branching=False (trace.py:155,244): distances of the one Dijkstra from the root ... */
int ko_field_distances(const float* f, int64_t sx, int64_t sy, int64_t sz, uint64_t source, float* d) {
  const int64_t n = sx * sy * sz;
  if ((int64_t)source >= n || f[source] == INFINITY) return KO_EINVAL;
  return ko_field_sssp(f, sx, sy, sz, source, 0, d, 0, 0);
}
/* ... and dijkstra3d.path_from_parents(parents, target) as a predecessor walk target -> source on those
 * distances, returned source -> ... -> target (handles absorption plateaus, see ko_walk).      */
int ko_path_to_source(const float* f, const float* d, int64_t sx, int64_t sy, int64_t sz, uint64_t source,
                      uint64_t target, uint64_t* path, int64_t* npath) {
  if (d[target] == INFINITY) return KO_ENOPATH;
  int rc = ko_walk(f, d, sx, sy, sz, target, source, 0, path, npath);
  if (rc) return rc;
  const int64_t k = *npath;
  for (int64_t i = 0; i < k / 2; i++) { uint64_t t = path[i]; path[i] = path[k - 1 - i]; path[k - 1 - i] = t; }
  return KO_OK;
}

/* a7: dijkstra3d.parental_field(field, source) (trace.py:155): parents[v] =
 * linear index of pred(v) + 1, 0 = none (source / unreachable).               */
int ko_parental_field(const float* f, int64_t sx, int64_t sy, int64_t sz, uint64_t source,
                      float* d, uint32_t* parents) {
  const int64_t n = sx * sy * sz;
  if ((int64_t)source >= n || f[source] == INFINITY) return KO_EINVAL;
  int rc = ko_field_sssp(f, sx, sy, sz, source, 0, d, 0, 0);
  if (rc) return rc;
  for (int64_t v = 0; v < n; v++) {
    parents[v] = 0;
    if ((uint64_t)v == source || d[v] == INFINITY) continue;
    uint64_t u;
    if (ko_pred(f, d, sx, sy, sz, (uint64_t)v, 0, &u) == KO_OK) {
      if (d[u] >= d[v]) return KO_EPLATEAU;
      parents[v] = (uint32_t)(u + 1);
    }
  }
  return KO_OK;
}

/* dijkstra3d.path_from_parents(parents, target) (trace.py:244): pointer chase,
 * returned source -> ... -> target.                                           */
int ko_path_from_parents(const uint32_t* parents, int64_t n, uint64_t target,
                         uint64_t* path, int64_t* npath) {
  int64_t k = 0; uint64_t v = target;
  path[k++] = v;
  while (parents[v] != 0) { v = parents[v] - 1; if (k >= n) return KO_EPLATEAU; path[k++] = v; }
  for (int64_t i = 0; i < k / 2; i++) { uint64_t t = path[i]; path[i] = path[k - 1 - i]; path[k - 1 - i] = t; }
  *npath = k;
  return KO_OK;
}

/* ------------------------------------------------------------------------- */
/* a9: roll_invalidation_ball_inside_component, skeletontricks.pyx:373-418 ->
 * _roll_invalidation_ball, dijkstra_invalidation.hpp:239-332.
 *
 * The flood is ORDER DEPENDENT: ownership of a voxel goes to the
 * source whose node pops first, and among equal keys the pop order is decided
 * by std::priority_queue with the NON-STRICT comparator `t1.dist >= t2.dist`
 * (dijkstra_invalidation.hpp:233-237).  To be bit-exact the heap below restates
 * the libstdc++ binary heap algorithm (bits/stl_heap.h: __push_heap,
 * __adjust_heap, __pop_heap) operation for operation, and the neighbour
 * enumeration restates dijkstra_invalidation.hpp:60-124 INCLUDING its quirk
 * that the corner entries 18..25 are gated on y and z only (so at an x border
 * they degenerate into a duplicate of the yz diagonal and are pushed twice).
 * Pinned against the compiled reference in tests/test_oracle_vs_ref.py.       */
typedef struct { float dist; uint64_t src; uint64_t val; float maxd; } ko_inode;
typedef struct { ko_inode* a; size_t n, cap; } ko_iheap;

static int ko_ipush(ko_iheap* h, ko_inode x) {
  if (h->n == h->cap) {
    size_t nc = h->cap ? h->cap * 2 : 4096;
    ko_inode* na = (ko_inode*)realloc(h->a, nc * sizeof(ko_inode));
    if (!na) return KO_ENOMEM;
    h->a = na; h->cap = nc;
  }
  /* emplace_back + push_heap: __push_heap(first, holeIndex=n, topIndex=0, value) */
  size_t hole = h->n++;
  while (hole > 0) {
    size_t parent = (hole - 1) / 2;
    if (!(h->a[parent].dist >= x.dist)) break;        /* comp(parent, value) */
    h->a[hole] = h->a[parent];
    hole = parent;
  }
  h->a[hole] = x;
  return KO_OK;
}
static void ko_ipop(ko_iheap* h) {
  /* pop_heap(begin,end) then pop_back */
  size_t len = h->n;
  if (len > 1) {
    len--;                                /* last now points at the old back */
    ko_inode value = h->a[len];
    h->a[len] = h->a[0];
    /* __adjust_heap(first, 0, len, value) */
    size_t hole = 0, child = 0;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (h->a[child].dist >= h->a[child - 1].dist) child--;   /* comp(right,left) */
      h->a[hole] = h->a[child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      h->a[hole] = h->a[child - 1];
      hole = child - 1;
    }
    while (hole > 0) {
      size_t parent = (hole - 1) / 2;
      if (!(h->a[parent].dist >= value.dist)) break;
      h->a[hole] = h->a[parent];
      hole = parent;
    }
    h->a[hole] = value;
  }
  h->n--;
}

/* neighbour offsets exactly as dijkstra_invalidation.hpp:60-124 (0 = absent) */
static void ko_nhood26(int64_t* nb, int64_t x, int64_t y, int64_t z, int64_t sx, int64_t sy, int64_t sz) {
  const int64_t sxy = sx * sy;
  nb[0] = -1 * (x > 0);
  nb[1] = (x < sx - 1);
  nb[2] = -sx * (y > 0);
  nb[3] = sx * (y < sy - 1);
  nb[4] = -sxy * (z > 0);
  nb[5] = sxy * (z < sz - 1);
  nb[6] = (nb[0] + nb[2]) * (nb[0] && nb[2]);
  nb[7] = (nb[0] + nb[3]) * (nb[0] && nb[3]);
  nb[8] = (nb[1] + nb[2]) * (nb[1] && nb[2]);
  nb[9] = (nb[1] + nb[3]) * (nb[1] && nb[3]);
  nb[10] = (nb[2] + nb[4]) * (nb[2] && nb[4]);
  nb[11] = (nb[2] + nb[5]) * (nb[2] && nb[5]);
  nb[12] = (nb[3] + nb[4]) * (nb[3] && nb[4]);
  nb[13] = (nb[3] + nb[5]) * (nb[3] && nb[5]);
  nb[14] = (nb[0] + nb[4]) * (nb[0] && nb[4]);
  nb[15] = (nb[0] + nb[5]) * (nb[0] && nb[5]);
  nb[16] = (nb[1] + nb[4]) * (nb[1] && nb[4]);
  nb[17] = (nb[1] + nb[5]) * (nb[1] && nb[5]);
  nb[18] = (nb[0] + nb[2] + nb[4]) * (nb[2] && nb[4]);
  nb[19] = (nb[1] + nb[2] + nb[4]) * (nb[2] && nb[4]);
  nb[20] = (nb[0] + nb[3] + nb[4]) * (nb[3] && nb[4]);
  nb[21] = (nb[0] + nb[2] + nb[5]) * (nb[2] && nb[5]);
  nb[22] = (nb[1] + nb[3] + nb[4]) * (nb[3] && nb[4]);
  nb[23] = (nb[1] + nb[2] + nb[5]) * (nb[2] && nb[5]);
  nb[24] = (nb[0] + nb[3] + nb[5]) * (nb[3] && nb[5]);
  nb[25] = (nb[1] + nb[3] + nb[5]) * (nb[3] && nb[5]);
}

/* instrumentation for DESIGN.md sizing: peak heap size of the last call */
int64_t ko_last_heap_peak = 0;
int64_t ko_get_last_heap_peak(void) { return ko_last_heap_peak; }

/* voxel_connectivity_graph bit of neighbourhood entry i (dijkstra_invalidation.hpp:152-190; the layout is cc3d's):
 * an entry whose bit is clear in the graph word of the CURRENT voxel is dropped -- after the neighbourhood helper has
 * produced it, so a corner entry that degenerated into a yz diagonal at an x face is gated by the corner's bit. */
static const int ko_graph_bit[26] = {1, 0, 3, 2, 5, 4,   9, 7, 8, 6,   17, 13, 16, 12,   15, 11, 14, 10,
                                     25, 24, 23, 21, 22, 20, 19, 18};

int ko_invalidate_ball_graph(uint8_t* field, int64_t sx, int64_t sy, int64_t sz,
                             float wx, float wy, float wz,
                             const uint64_t* sources, const float* max_distances, int64_t nsrc,
                             const uint32_t* graph, int64_t* invalidated, int64_t* heap_ops);

int ko_invalidate_ball(uint8_t* field, int64_t sx, int64_t sy, int64_t sz,
                       float wx, float wy, float wz,
                       const uint64_t* sources, const float* max_distances, int64_t nsrc,
                       int64_t* invalidated, int64_t* heap_ops) {
  return ko_invalidate_ball_graph(field, sx, sy, sz, wx, wy, wz, sources, max_distances, nsrc, NULL, invalidated, heap_ops);
}

/* skeletontricks.pyx:373-418 with voxel_connectivity_graph (nullable) -> dijkstra_invalidation.hpp:239-332 */
int ko_invalidate_ball_graph(uint8_t* field, int64_t sx, int64_t sy, int64_t sz,
                             float wx, float wy, float wz,
                             const uint64_t* sources, const float* max_distances, int64_t nsrc,
                             const uint32_t* graph, int64_t* invalidated, int64_t* heap_ops) {
  const int64_t sxy = sx * sy;
  ko_iheap h = {0, 0, 0};
  int64_t ops = 0;
  for (int64_t i = 0; i < nsrc; i++) {
    ko_inode nd = {0.0f, sources[i], sources[i], max_distances[i]};
    if (ko_ipush(&h, nd)) { free(h.a); return KO_ENOMEM; }
    ops++;
  }
  int64_t count = 0;
  int64_t nb[26];
  size_t peak = h.n;
  while (h.n) {
    if (h.n > peak) peak = h.n;
    const float maxd = h.a[0].maxd;
    const uint64_t src = h.a[0].src;
    const uint64_t loc = h.a[0].val;
    ko_ipop(&h);
    if (!field[loc]) continue;
    field[loc] = 0;
    count++;
    int64_t z = (int64_t)(loc / (uint64_t)sxy), r = (int64_t)(loc % (uint64_t)sxy), y = r / sx, x = r % sx;
    int64_t oz = (int64_t)(src / (uint64_t)sxy), orr = (int64_t)(src % (uint64_t)sxy), oy = orr / sx, ox = orr % sx;
    ko_nhood26(nb, x, y, z, sx, sy, sz);
    if (graph) {
      const uint32_t gw = graph[loc];
      for (int i = 0; i < 26; i++) if (!((gw >> ko_graph_bit[i]) & 1u)) nb[i] = 0;
    }
    for (int i = 0; i < 26; i++) {
      if (nb[i] == 0) continue;
      uint64_t q = (uint64_t)((int64_t)loc + nb[i]);
      if (field[q] == 0) continue;
      int64_t qz = (int64_t)(q / (uint64_t)sxy), qr = (int64_t)(q % (uint64_t)sxy), qy = qr / sx, qx = qr % sx;
      float a = wx * (float)(qx - ox);
      float b = wy * (float)(qy - oy);
      float c = wz * (float)(qz - oz);
      float s = a * a; float t = b * b; float u = c * c;
      s = s + t; s = s + u;
      float nd = sqrtf(s);
      if (nd < maxd) {
        ko_inode node = {nd, src, q, maxd};
        if (ko_ipush(&h, node)) { free(h.a); return KO_ENOMEM; }
        ops++;
      }
    }
  }
  free(h.a);
  ko_last_heap_peak = (int64_t)peak;
  *invalidated = count;
  if (heap_ops) *heap_ops = ops;
  return KO_OK;
}

/* radii for a9: (scale * DBF[x,y,z] + constant) evaluated through numpy float32
 * scalars, skeletontricks.pyx:393-395 (numpy 2.x: every op rounds to f32).    */
void ko_ball_radii(const float* dbf, const uint64_t* path, int64_t n, float scale, float constant, float* r) {
  for (int64_t i = 0; i < n; i++) { float t = scale * dbf[path[i]]; t = t + constant; r[i] = t; }
}

/* ------------------------------------------------------------------------- */
/* a10: roll_invalidation_cube, skeletontricks.pyx:766-836 ->
 * _roll_invalidation_cube, skeletontricks.hpp:42-155.  Semantics: union of
 * inclusive boxes lo = max(0, trunc(c - r/w)), hi = min(s-1, trunc(0.5 + c + r/w))
 * with r = scale*DBF[c] + constant (float), zero everything inside, count the
 * non-zero voxels erased.  (The reference's difference-array trick, including
 * its direct handling of minx==maxx boxes, computes exactly this union.)      */
int ko_invalidate_cube(uint8_t* labels, const float* dbf, int64_t sx, int64_t sy, int64_t sz,
                       float wx, float wy, float wz, const uint64_t* path, int64_t npath,
                       float scale, float constant, int64_t* invalidated) {
  const int64_t sxy = sx * sy;
  int64_t count = 0;
  for (int64_t i = 0; i < npath; i++) {
    uint64_t loc = path[i];
    float radius = scale * dbf[loc];
    radius = radius + constant;
    int64_t z = (int64_t)(loc / (uint64_t)sxy), r = (int64_t)(loc % (uint64_t)sxy), y = r / sx, x = r % sx;
    float rx = radius / wx, ry = radius / wy, rz = radius / wz;
    int64_t lo[3], hi[3];
    int64_t c[3] = {x, y, z}; int64_t s[3] = {sx, sy, sz}; float rr[3] = {rx, ry, rz};
    for (int a = 0; a < 3; a++) {
      float fl = (float)c[a] - rr[a];
      int64_t l = (int64_t)fl; if (l < 0) l = 0;
      float fh = (float)c[a] + rr[a];
      double dh = 0.5 + (double)fh;
      int64_t hh = (int64_t)dh; if (hh > s[a] - 1) hh = s[a] - 1;
      lo[a] = l; hi[a] = hh;
    }
    for (int64_t zz = lo[2]; zz <= hi[2]; zz++)
      for (int64_t yy = lo[1]; yy <= hi[1]; yy++)
        for (int64_t xx = lo[0]; xx <= hi[0]; xx++) {
          int64_t q = xx + sx * yy + sxy * zz;
          if (labels[q]) { count++; labels[q] = 0; }
        }
  }
  *invalidated = count;
  return KO_OK;
}

/* ------------------------------------------------------------------------- */
/* small helpers: skeletontricks.pyx:177-224 (inf2zero / zero2inf),
 * :307-326 (first_label, F-order raster).                                     */
void ko_zero2inf(float* f, int64_t n) { for (int64_t i = 0; i < n; i++) if (f[i] == 0.0f) f[i] = INFINITY; }
void ko_inf2zero(float* f, int64_t n) { for (int64_t i = 0; i < n; i++) if (f[i] == INFINITY) f[i] = 0.0f; }
int64_t ko_first_label(const uint8_t* m, int64_t n) { for (int64_t i = 0; i < n; i++) if (m[i]) return i; return -1; }

/* ------------------------------------------------------------------------- */
/* preamble helper (row f1, host side): 26-connected multi-label connected
 * components, restating cc3d.connected_components as called at
 * kimimaro/utility.py:74-77 (third-party, source absent).  Component ids are
 * assigned 1..N in order of first appearance in the F-order raster.           */
static uint32_t ko_find(uint32_t* p, uint32_t i) { while (p[i] != i) { p[i] = p[p[i]]; i = p[i]; } return i; }
int64_t ko_ccl26(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz, uint32_t* out) {
  const int64_t sxy = sx * sy, n = sxy * sz;
  uint32_t* parent = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n + 1));
  if (!parent) return -1;
  /* the 13 already-visited neighbours in raster order */
  static const int8_t back[13][3] = {
    {-1,0,0},{-1,-1,0},{0,-1,0},{1,-1,0},
    {-1,-1,-1},{0,-1,-1},{1,-1,-1},{-1,0,-1},{0,0,-1},{1,0,-1},{-1,1,-1},{0,1,-1},{1,1,-1}};
  for (int64_t z = 0; z < sz; z++) for (int64_t y = 0; y < sy; y++) for (int64_t x = 0; x < sx; x++) {
    int64_t i = x + sx * y + sxy * z;
    uint64_t L = ko_lab(labels, label_bytes, i);
    parent[i] = (uint32_t)i;
    if (L == 0) continue;
    for (int k = 0; k < 13; k++) {
      int64_t nx = x + back[k][0], ny = y + back[k][1], nz = z + back[k][2];
      if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy) continue;
      int64_t j = nx + sx * ny + sxy * nz;
      if (ko_lab(labels, label_bytes, j) != L) continue;
      uint32_t a = ko_find(parent, (uint32_t)i), b = ko_find(parent, (uint32_t)j);
      if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }
    }
  }
  int64_t next = 0;
  for (int64_t i = 0; i < n; i++) {
    if (ko_lab(labels, label_bytes, i) == 0) { out[i] = 0; continue; }
    uint32_t r = ko_find(parent, (uint32_t)i);
    if (r == (uint32_t)i) out[i] = (uint32_t)(++next); /* roots are the smallest index => first appearance */
    else out[i] = out[r];
  }
  free(parent);
  return next;
}
