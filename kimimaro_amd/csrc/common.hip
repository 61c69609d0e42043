// common.hip -- error plumbing, device check, geometry and the host-side preamble helper.
#include "common.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

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
  static const int8_t back[13][3] = {{-1, 0, 0}, {-1, -1, 0}, {0, -1, 0}, {1, -1, 0}, {-1, -1, -1}, {0, -1, -1}, {1, -1, -1},
                                     {-1, 0, -1}, {0, 0, -1}, {1, 0, -1}, {-1, 1, -1}, {0, 1, -1}, {1, 1, -1}};
  for (int64_t z = 0; z < sz; z++)
    for (int64_t y = 0; y < sy; y++)
      for (int64_t x = 0; x < sx; x++) {
        const int64_t i = x + sx * y + sxy * z;
        parent[i] = (uint32_t)i;
        const LT L = lab[i];
        if (L == 0) continue;
        for (int k = 0; k < 13; k++) {
          const int64_t nx = x + back[k][0], ny = y + back[k][1], nz = z + back[k][2];
          if (nx < 0 || ny < 0 || nz < 0 || nx >= sx || ny >= sy) continue;
          const int64_t j = nx + sx * ny + sxy * nz;
          if (lab[j] != L) continue;
          const uint32_t a = find((uint32_t)i), b = find((uint32_t)j);
          if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }
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

// ---- row f2 (host side): skeletontricks.find_border_targets (skeletontricks.pyx:591-647) with
// compute_centroids (:528-588) and compute_tiebreaker_maxima (:650-760) restated op for op
// (float vs double arithmetic exactly as the Cython source compiles, including the reference's
// use of sx in the fourth corner of `cornerness`, :744).
namespace {
inline float bt_distsq(float p1x, float p1y, float p2x, float p2y, float wx, float wy) {
  p1x = wx * (p1x - p2x);
  p1y = wy * (p1y - p2y);
  const float a = p1x * p1x;
  const float b = p1y * p1y;
  return a + b;
}
inline float bt_min4f(float a, float b, float c, float d) {
  float m = a;
  if (b < m) m = b;
  if (c < m) m = c;
  if (d < m) m = d;
  return m;
}
inline float bt_cornerness(float x, float y, float sx, float sy, float wx, float wy) {
  const float a = bt_distsq(x, y, -0.5f, -0.5f, wx, wy);
  const float b = bt_distsq(x, y, (float)((double)sx - 0.5), -0.5f, wx, wy);
  const float c = bt_distsq(x, y, (float)((double)sx - 0.5), (float)((double)sy - 0.5), wx, wy);
  const float d = bt_distsq(x, y, -0.5f, (float)((double)sx - 0.5), wx, wy);
  return bt_min4f(a, b, c, d);
}
inline float bt_edgeness(float x, float y, float sx, float sy, float wx, float wy) {
  const double a = (double)wx * ((double)x - 0.5);
  const double b = (double)wx * ((double)sx - 0.5 - (double)x);
  const double c = (double)wy * ((double)y - 0.5);
  const double d = (double)wy * ((double)sy - 0.5 - (double)y);
  double m = a;
  if (b < m) m = b;
  if (c < m) m = c;
  if (d < m) m = d;
  return (float)m;
}
}  // namespace

// dt, cc: [sx, sy] Fortran ordered.  For every component id 1..nlab that has a maximum:
// out_xy[2*id], out_xy[2*id+1] = its (x, y) (as floats, to be truncated by int()), and `order`
// lists the ids in dict insertion order.  Returns the number of ids written to `order`.
extern "C" int64_t kh_host_find_border_targets(const float* dt, const uint32_t* cc, int64_t sx, int64_t sy, float wx,
                                               float wy, int64_t nlab, float* out_xy, int32_t* order) {
  const int64_t n1 = nlab + 1;
  float* xsum = (float*)calloc((size_t)n1, sizeof(float));
  float* ysum = (float*)calloc((size_t)n1, sizeof(float));
  uint32_t* ct = (uint32_t*)calloc((size_t)n1, sizeof(uint32_t));
  double* mx = (double*)calloc((size_t)n1, sizeof(double));
  uint8_t* have = (uint8_t*)calloc((size_t)n1, 1);
  float* cent = (float*)calloc((size_t)n1 * 2, sizeof(float));
  if (!xsum || !ysum || !ct || !mx || !have || !cent) return -1;
  for (int64_t x = 0; x < sx; x++)
    for (int64_t y = 0; y < sy; y++) {
      const uint32_t L = cc[x + sx * y];
      if (L == 0) continue;
      xsum[L] = xsum[L] + (float)(uint64_t)x;
      ysum[L] = ysum[L] + (float)(uint64_t)y;
      ct[L] += 1;
    }
  const float cx = (float)((double)(wx * (float)(uint64_t)sx) / 2.0);
  const float cy = (float)((double)(wy * (float)(uint64_t)sy) / 2.0);
  for (int64_t L = 0; L < n1; L++) {
    if (ct[L] == 0) continue;
    float px = wx * xsum[L] / (float)ct[L];
    float py = wy * ysum[L] / (float)ct[L];
    if (!(px - cx >= 0)) px = px + wx;
    if (!(py - cy >= 0)) py = py + wy;
    cent[2 * L] = (float)(int)(px / wx);
    cent[2 * L + 1] = (float)(int)(py / wy);
  }
  int64_t norder = 0;
  const float fsx = (float)(uint64_t)sx, fsy = (float)(uint64_t)sy;
  for (int64_t y = 0; y < sy; y++)
    for (int64_t x = 0; x < sx; x++) {
      const uint32_t L = cc[x + sx * y];
      if (L == 0) continue;
      const float d = dt[x + sx * y];
      if (d == 0) continue;
      if ((double)d > mx[L]) {
        mx[L] = (double)d;
        if (!have[L]) { have[L] = 1; order[norder++] = (int32_t)L; }
        out_xy[2 * L] = (float)x;
        out_xy[2 * L + 1] = (float)y;
      } else if (mx[L] == (double)d) {
        const float px = out_xy[2 * L], py = out_xy[2 * L + 1];
        const float fx = (float)x, fy = (float)y;
        const float centx = cent[2 * L], centy = cent[2 * L + 1];
        bool take = false;
        float d1 = bt_distsq(px, py, centx, centy, wx, wy);
        float d2 = bt_distsq(fx, fy, centx, centy, wx, wy);
        if (d2 < d1) take = true;
        else if (d1 == d2) {
          d1 = bt_distsq(px, py, cx, cy, wx, wy);
          d2 = bt_distsq(fx, fy, cx, cy, wx, wy);
          if (d2 < d1) take = true;
          else if (d1 == d2) {
            d1 = bt_cornerness(px, py, fsx, fsy, wx, wy);
            d2 = bt_cornerness(fx, fy, fsx, fsy, wx, wy);
            if (d2 < d1) take = true;
            else if (d1 == d2) {
              d1 = bt_edgeness(px, py, fsx, fsy, wx, wy);
              d2 = bt_edgeness(fx, fy, fsx, fsy, wx, wy);
              if (d2 < d1) take = true;
            }
          }
        }
        if (take) { out_xy[2 * L] = fx; out_xy[2 * L + 1] = fy; }
      }
    }
  free(xsum); free(ysum); free(ct); free(mx); free(have); free(cent);
  return norder;
}
