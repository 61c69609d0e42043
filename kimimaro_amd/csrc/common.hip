// common.hip -- error plumbing, device check, geometry and the host-side preamble helper.
#include "common.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <new>
#include <utility>
#include <vector>

namespace kh {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int g_device_ok = 0;

int require_device() {
  if (g_device_ok) return KH_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("libkimi_hip: no HIP device visible (%s); this library has no CPU fallback",
              e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return KH_ENODEVICE;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    set_error("libkimi_hip: hipGetDeviceProperties failed");
    return KH_ENODEVICE;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    set_error("libkimi_hip: built for gfx950 (MI355X) only, found %s", p.gcnArchName);
    return KH_ENODEVICE;
  }
  g_device_ok = 1;
  return KH_OK;
}

void make_geometry(Geometry& g, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz) {
  g.sx = (int32_t)sx;
  g.sy = (int32_t)sy;
  g.sz = (int32_t)sz;
  g.sxy = (int32_t)(sx * sy);
  g.wx = wx;
  g.wy = wy;
  g.wz = wz;
  for (int i = 0; i < 26; i++) {
    int dx, dy, dz;
    dir_delta(i, dx, dy, dz);
    g.off[i] = (int32_t)(dx + sx * dy + sx * sy * dz);
    // dijkstra_invalidation.hpp:45-52 (_s, _c): sqrt(wa*wa + wb*wb + wc*wc), float, no contraction
    const float a = dx ? wx : 0.0f, b = dy ? wy : 0.0f, c = dz ? wz : 0.0f;
    float s = a * a;
    const float t = b * b;
    const float u = c * c;
    s = s + t;
    s = s + u;
    g.w[i] = sqrtf(s);
  }
}

}  // namespace kh

extern "C" int kh_version(void) { return 100; }

extern "C" int kh_last_error(char* buf, int len) {
  const int n = (int)strlen(kh::g_err);
  if (buf && len > 0) {
    strncpy(buf, kh::g_err, (size_t)len - 1);
    buf[len - 1] = 0;
  }
  return n;
}

extern "C" int kh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

// ---- preamble row f1 (host side for now): 26-connected multi-label CCL, union-find over the 13
// already visited neighbours, ids 1..N by first appearance in the F-order raster
// (cc3d.connected_components as called at kimimaro/utility.py:74-77).
namespace {
template <typename LT>
int64_t ccl26(const LT* lab, int64_t sx, int64_t sy, int64_t sz, uint32_t* out) {
  const int64_t sxy = sx * sy, n = sxy * sz;
  uint32_t* parent = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
  if (!parent) return -1;
  auto find = [&](uint32_t i) {
    while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
    return i;
  };
  auto unite = [&](uint32_t i, uint32_t j) {
    const uint32_t a = find(i), b = find(j);
    if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }
  };
  static const int8_t below[9][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {0, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};
  for (int64_t z = 0; z < sz; z++)
    for (int64_t y = 0; y < sy; y++)
      for (int64_t x = 0; x < sx; x++) {
        const int64_t i = x + sx * y + sxy * z;
        parent[i] = (uint32_t)i;
        const LT L = lab[i];
        if (L == 0) continue;
        // the four visited neighbours of the plane, without the links that earlier pixels have made already: (x, y-1) touches
        // the other three; (x-1, y) touches (x-1, y-1)  [the faces of the volume are planes: this is the whole job there]
        const bool has_up = y > 0, has_l = x > 0, has_r = x + 1 < sx;
        if (has_up && lab[i - sx] == L) unite((uint32_t)i, (uint32_t)(i - sx));
        else {
          if (has_l && lab[i - 1] == L) unite((uint32_t)i, (uint32_t)(i - 1));
          else if (has_l && has_up && lab[i - sx - 1] == L) unite((uint32_t)i, (uint32_t)(i - sx - 1));
          if (has_r && has_up && lab[i - sx + 1] == L) unite((uint32_t)i, (uint32_t)(i - sx + 1));
        }
        if (z == 0) continue;
        for (int k = 0; k < 9; k++) {
          const int64_t nx = x + below[k][0], ny = y + below[k][1];
          if (nx < 0 || ny < 0 || nx >= sx || ny >= sy) continue;
          const int64_t j = nx + sx * ny + sxy * (z - 1);
          if (lab[j] == L) unite((uint32_t)i, (uint32_t)j);
        }
      }
  int64_t next = 0;
  for (int64_t i = 0; i < n; i++) {
    if (lab[i] == 0) { out[i] = 0; continue; }
    const uint32_t r = find((uint32_t)i);
    out[i] = (r == (uint32_t)i) ? (uint32_t)(++next) : out[r];
  }
  free(parent);
  return next;
}
}  // namespace

extern "C" int64_t kh_host_ccl26(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz, uint32_t* out) {
  switch (label_bytes) {
    case 1: return ccl26((const uint8_t*)labels, sx, sy, sz, out);
    case 2: return ccl26((const uint16_t*)labels, sx, sy, sz, out);
    case 4: return ccl26((const uint32_t*)labels, sx, sy, sz, out);
    case 8: return ccl26((const uint64_t*)labels, sx, sy, sz, out);
    default: kh::set_error("kh_host_ccl26: label_bytes must be 1, 2, 4 or 8"); return -1;
  }
}

// ---- row a12 (host side): Skeleton.from_path per path + simple_merge + consolidate (kimimaro/trace.py:182-184) for every
// component of a result group (see kimi_hip.h)
extern "C" int64_t kh_host_consolidate_paths(int64_t nslots, const int64_t* voff, const int64_t* loff, const uint32_t* locs,
                                             const uint32_t* lens, const float* radii, int64_t sx, int64_t sy, int64_t sz,
                                             float* out_verts, float* out_radii, uint32_t* out_edges, int64_t* vstart,
                                             int64_t* estart) {
  std::vector<std::pair<uint64_t, uint32_t>> keyed;   // (vertex key, position in the slot)
  std::vector<uint32_t> inv, firstpos, rank;
  std::vector<uint64_t> rows;
  std::vector<uint8_t> used;
  int64_t nv = 0, ne = 0;
  try {
    for (int64_t s = 0; s < nslots; s++) {
      vstart[s] = nv;
      estart[s] = ne;
      const int64_t v0 = voff[s], n = voff[s + 1] - v0;
      if (n <= 0) continue;
      // unique vertices, sorted like np.unique(vertices, axis=0): by (x, y, z); the first occurrence names the radius
      keyed.resize((size_t)n);
      for (int64_t i = 0; i < n; i++) {
        const uint64_t l = locs[v0 + i];
        const uint64_t x = l % (uint64_t)sx, y = (l / (uint64_t)sx) % (uint64_t)sy, z = l / ((uint64_t)sx * (uint64_t)sy);
        keyed[(size_t)i] = {(x * (uint64_t)sy + y) * (uint64_t)sz + z, (uint32_t)i};
      }
      std::sort(keyed.begin(), keyed.end());
      inv.resize((size_t)n);
      firstpos.clear();
      for (int64_t i = 0; i < n; i++) {
        if (i == 0 || keyed[(size_t)i].first != keyed[(size_t)i - 1].first) firstpos.push_back(keyed[(size_t)i].second);
        inv[keyed[(size_t)i].second] = (uint32_t)(firstpos.size() - 1);
      }
      const int64_t nu = (int64_t)firstpos.size();
      // consecutive vertices inside a path are edges; rows (lo, hi), lo != hi, unique, sorted
      rows.clear();
      int64_t pos = 0;
      for (int64_t p = loff[s]; p < loff[s + 1]; p++) {
        const int64_t len = lens[p];
        for (int64_t i = pos; i + 1 < pos + len; i++) {
          uint32_t a = inv[(size_t)i], b = inv[(size_t)i + 1];
          if (a == b) continue;
          if (a > b) { const uint32_t t = a; a = b; b = t; }
          rows.push_back(((uint64_t)a << 32) | b);
        }
        pos += len;
      }
      std::sort(rows.begin(), rows.end());
      rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
      // vertices no edge refers to are dropped (Skeleton.consolidate); local index = rank among the used ones
      used.assign((size_t)nu, 0);
      for (uint64_t r : rows) { used[(size_t)(r >> 32)] = 1; used[(size_t)(r & 0xffffffffu)] = 1; }
      rank.resize((size_t)nu);
      uint32_t k = 0;
      for (int64_t u = 0; u < nu; u++) {
        rank[(size_t)u] = k;
        if (!used[(size_t)u]) continue;
        const uint64_t l = locs[v0 + firstpos[(size_t)u]];
        out_verts[3 * (nv + k)] = (float)(l % (uint64_t)sx);
        out_verts[3 * (nv + k) + 1] = (float)((l / (uint64_t)sx) % (uint64_t)sy);
        out_verts[3 * (nv + k) + 2] = (float)(l / ((uint64_t)sx * (uint64_t)sy));
        out_radii[nv + k] = radii[v0 + firstpos[(size_t)u]];
        k++;
      }
      for (uint64_t r : rows) {
        out_edges[2 * ne] = rank[(size_t)(r >> 32)];
        out_edges[2 * ne + 1] = rank[(size_t)(r & 0xffffffffu)];
        ne++;
      }
      nv += k;
    }
    vstart[nslots] = nv;
    estart[nslots] = ne;
  } catch (const std::bad_alloc&) {
    return -1;
  }
  return nv;
}

// ---- row a12 (host side): kimimaro/intake.py:587-593 for every label of a volume (see kimi_hip.h)
extern "C" int64_t kh_host_merge_components(int64_t nlabels, const int64_t* part_of_label, const int64_t* vstart, const int64_t* estart,
                                            const float* verts, const float* radii, const uint32_t* edges, int64_t sy, int64_t sz,
                                            float ax, float ay, float az, float* out_verts, float* out_radii, uint32_t* out_edges) {
  std::vector<std::pair<int64_t, uint32_t>> keyed;       // (vertex key, position in the label's concatenation)
  std::vector<uint32_t> rank;
  std::vector<uint64_t> rows;
  std::vector<uint32_t> first, placed;
  try {
    for (int64_t l = 0; l < nlabels; l++) {
      const int64_t p0 = part_of_label[l], p1 = part_of_label[l + 1];
      if (p1 <= p0) continue;
      const int64_t v0 = vstart[p0], v1 = vstart[p1], e0 = estart[p0], e1 = estart[p1];
      const int64_t n = v1 - v0;
      if (p1 - p0 == 1) {                                 // one component: sorted already, indices local already
        for (int64_t i = v0; i < v1; i++) {
          out_verts[3 * i] = verts[3 * i] * ax; out_verts[3 * i + 1] = verts[3 * i + 1] * ay; out_verts[3 * i + 2] = verts[3 * i + 2] * az;
          out_radii[i] = radii[i];
        }
        for (int64_t j = 2 * e0; j < 2 * e1; j++) out_edges[j] = edges[j];
        continue;
      }
      keyed.resize((size_t)n);
      for (int64_t i = 0; i < n; i++) {
        const float* v = verts + 3 * (v0 + i);
        keyed[(size_t)i] = {((int64_t)v[0] * sy + (int64_t)v[1]) * sz + (int64_t)v[2], (uint32_t)i};
      }
      // every part is sorted already (consolidate_paths_batch): merge the runs; anything else is sorted from scratch
      // (keys are distinct: the components are disjoint voxel sets)
      bool runs = true;
      for (int64_t p = p0; p < p1 && runs; p++)
        runs = std::is_sorted(keyed.begin() + (vstart[p] - v0), keyed.begin() + (vstart[p + 1] - v0));
      if (runs) for (int64_t p = p0 + 1; p < p1; p++)
        std::inplace_merge(keyed.begin(), keyed.begin() + (vstart[p] - v0), keyed.begin() + (vstart[p + 1] - v0));
      else std::sort(keyed.begin(), keyed.end());
      rank.resize((size_t)n);
      for (int64_t i = 0; i < n; i++) {
        const int64_t src = v0 + keyed[(size_t)i].second, dst = v0 + i;
        rank[keyed[(size_t)i].second] = (uint32_t)i;
        out_verts[3 * dst] = verts[3 * src] * ax; out_verts[3 * dst + 1] = verts[3 * src + 1] * ay; out_verts[3 * dst + 2] = verts[3 * src + 2] * az;
        out_radii[dst] = radii[src];
      }
      // rows (lo, hi) sorted lexicographically: a counting sort on lo (a vertex starts a handful of edges), then the few
      // rows of each lo by hi
      const int64_t m = e1 - e0;
      rows.resize((size_t)m);
      first.assign((size_t)n + 1, 0u);
      {
        int64_t r = 0;
        for (int64_t p = p0; p < p1; p++) {
          const uint32_t base = (uint32_t)(vstart[p] - v0);
          for (int64_t j = estart[p]; j < estart[p + 1]; j++, r++) {
            uint32_t a = rank[base + edges[2 * j]], b = rank[base + edges[2 * j + 1]];
            if (a > b) { const uint32_t t = a; a = b; b = t; }
            rows[(size_t)r] = ((uint64_t)a << 32) | b;
            first[a + 1]++;
          }
        }
      }
      for (int64_t i = 0; i < n; i++) first[(size_t)i + 1] += first[(size_t)i];
      placed.assign(first.begin(), first.end() - 1);
      for (int64_t r = 0; r < m; r++) {
        const uint32_t a = (uint32_t)(rows[(size_t)r] >> 32);
        const int64_t at = e0 + (int64_t)placed[a]++;
        out_edges[2 * at] = a;
        out_edges[2 * at + 1] = (uint32_t)rows[(size_t)r];
      }
      for (int64_t i = 0; i < n; i++) {                    // insertion sort of every lo's rows by hi
        const int64_t lo = e0 + first[(size_t)i], hi = e0 + first[(size_t)i + 1];
        for (int64_t a = lo + 1; a < hi; a++) {
          const uint32_t x = out_edges[2 * a + 1];
          int64_t b = a;
          while (b > lo && out_edges[2 * (b - 1) + 1] > x) { out_edges[2 * b + 1] = out_edges[2 * (b - 1) + 1]; b--; }
          out_edges[2 * b + 1] = x;
        }
      }
    }
  } catch (const std::bad_alloc&) {
    return -1;
  }
  return 0;
}

// ---- row f2 (host side): skeletontricks.find_border_targets (skeletontricks.pyx:591-647, with compute_centroids
// :528-588 and compute_tiebreaker_maxima :650-760) on one face of the volume.
//
// What the reference computes, said without its control flow: for every component of the plane, the pixel with the largest
// (non-zero) distance-transform value; among several such pixels the one with the smallest tie-break tuple
//     ( squared distance to the component's centroid pixel, squared distance to the centre of the plane,
//       "cornerness" = squared distance to the nearest corner of the plane, "edgeness" = distance to the nearest edge ),
// compared lexicographically, and among equal tuples the first pixel of the (y outer, x inner) raster -- the reference's
// scan replaces its current pick only on a strictly smaller tuple, which is exactly "first minimum".  The ids are reported in
// the order in which the components first show a non-zero value in that raster (the insertion order of the reference's dict).
// Here: pass 1 = per-component statistics (coordinate sums, pixel count, maximum, first appearance), pass 2 = one
// first-minimum reduction over the pixels that hold their component's maximum.  The arithmetic of the tuple follows the
// compiled Cython exactly (which operands are float, which double), because its bits decide ties -- including the
// reference's fourth "corner", which is at (-0.5, sx - 0.5) instead of (-0.5, sy - 0.5) (:744).
namespace {
struct PlaneGeometry {
  float wx, wy;        // pixel pitch
  float fsx, fsy;      // extent as floats
  float mid_x, mid_y;  // centre of the plane in physical units
};

inline float scaled_sq(float ax, float ay, float bx, float by, const PlaneGeometry& g) {
  const float ex = g.wx * (ax - bx);
  const float ey = g.wy * (ay - by);
  const float e2x = ex * ex;
  const float e2y = ey * ey;
  return e2x + e2y;
}

struct TieKey {
  float k[4];
  bool operator<(const TieKey& o) const {
    for (int i = 0; i < 4; i++) {
      if (k[i] < o.k[i]) return true;
      if (!(k[i] == o.k[i])) return false;
    }
    return false;
  }
};

inline TieKey tie_key(float x, float y, float cpx, float cpy, const PlaneGeometry& g) {
  TieKey t;
  t.k[0] = scaled_sq(x, y, cpx, cpy, g);
  t.k[1] = scaled_sq(x, y, g.mid_x, g.mid_y, g);
  // nearest of the four "corners" (the last one as the reference has it, see above)
  const float hx = (float)((double)g.fsx - 0.5), hy = (float)((double)g.fsy - 0.5);
  const float corner[4][2] = {{-0.5f, -0.5f}, {hx, -0.5f}, {hx, hy}, {-0.5f, hx}};
  float near = scaled_sq(x, y, corner[0][0], corner[0][1], g);
  for (int c = 1; c < 4; c++) {
    const float d = scaled_sq(x, y, corner[c][0], corner[c][1], g);
    if (d < near) near = d;
  }
  t.k[2] = near;
  // nearest edge, in double like the reference's expression, rounded once
  const double to_edge[4] = {(double)g.wx * ((double)x - 0.5), (double)g.wx * ((double)g.fsx - 0.5 - (double)x),
                             (double)g.wy * ((double)y - 0.5), (double)g.wy * ((double)g.fsy - 0.5 - (double)y)};
  double e = to_edge[0];
  for (int c = 1; c < 4; c++) if (to_edge[c] < e) e = to_edge[c];
  t.k[3] = (float)e;
  return t;
}
}  // namespace

// dt, cc: [sx, sy] Fortran ordered.  For every component id 1..nlab that has a maximum:
// out_xy[2*id], out_xy[2*id+1] = its (x, y) (as floats, to be truncated by int()), and `order`
// lists the ids in dict insertion order.  Returns the number of ids written to `order`.
extern "C" int64_t kh_host_find_border_targets(const float* dt, const uint32_t* cc, int64_t sx, int64_t sy, float wx,
                                               float wy, int64_t nlab, float* out_xy, int32_t* order) {
  struct Stat { float sum_x, sum_y; uint32_t pixels; float peak; int64_t first_seen; float cpx, cpy; bool picked; TieKey best; };
  const size_t n1 = (size_t)nlab + 1;
  Stat* st = (Stat*)calloc(n1, sizeof(Stat));
  if (!st) return -1;
  for (size_t l = 0; l < n1; l++) st[l].first_seen = -1;
  PlaneGeometry g;
  g.wx = wx; g.wy = wy;
  g.fsx = (float)(uint64_t)sx; g.fsy = (float)(uint64_t)sy;
  g.mid_x = (float)((double)(wx * g.fsx) / 2.0);
  g.mid_y = (float)((double)(wy * g.fsy) / 2.0);
  // pass 1a: coordinate sums in the reference's accumulation order (x outer, y inner: float sums are order sensitive)
  for (int64_t x = 0; x < sx; x++)
    for (int64_t y = 0; y < sy; y++) {
      const uint32_t l = cc[x + sx * y];
      if (!l) continue;
      Stat& s = st[l];
      s.sum_x = s.sum_x + (float)(uint64_t)x;
      s.sum_y = s.sum_y + (float)(uint64_t)y;
      s.pixels++;
    }
  // pass 1b: peak value and first appearance of a non-zero value, in the raster the ids are reported in
  for (int64_t i = 0, n = sx * sy; i < n; i++) {
    const uint32_t l = cc[i];
    const float d = dt[i];
    if (!l || d == 0) continue;
    Stat& s = st[l];
    if (s.first_seen < 0) s.first_seen = i;
    if (d > s.peak) s.peak = d;
  }
  // centroid pixel: the mean position, moved one pixel up when it lies below the centre of the plane, truncated
  for (size_t l = 1; l < n1; l++) {
    Stat& s = st[l];
    if (!s.pixels) continue;
    float px = wx * s.sum_x / (float)s.pixels;
    float py = wy * s.sum_y / (float)s.pixels;
    if (!(px - g.mid_x >= 0)) px = px + wx;
    if (!(py - g.mid_y >= 0)) py = py + wy;
    s.cpx = (float)(int)(px / wx);
    s.cpy = (float)(int)(py / wy);
  }
  // pass 2: first minimum of the tie-break tuple over the pixels at their component's peak
  int64_t norder = 0;
  for (int64_t y = 0; y < sy; y++)
    for (int64_t x = 0; x < sx; x++) {
      const int64_t i = x + sx * y;
      const uint32_t l = cc[i];
      if (!l) continue;
      Stat& s = st[l];
      if (s.first_seen == i) order[norder++] = (int32_t)l;
      if (dt[i] == 0 || dt[i] != s.peak) continue;
      const TieKey t = tie_key((float)x, (float)y, s.cpx, s.cpy, g);
      if (!s.picked || t < s.best) {
        s.picked = true;
        s.best = t;
        out_xy[2 * l] = (float)x;
        out_xy[2 * l + 1] = (float)y;
      }
    }
  free(st);
  return norder;
}
