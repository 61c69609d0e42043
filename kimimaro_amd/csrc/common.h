// common.h -- shared host/device helpers for libkimi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/kimi_hip.h"

namespace kh {

void set_error(const char* fmt, ...);

#define KH_HIP_CHECK(expr)                                                         \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      kh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return KH_EHIP;                                                              \
    }                                                                              \
  } while (0)

#define KH_LAUNCH_CHECK()                                                          \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      kh::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return KH_EHIP;                                                              \
    }                                                                              \
  } while (0)

int require_device();  // KH_OK or KH_ENODEVICE (+message); cached after the first success

static constexpr float KH_INF = __builtin_huge_valf();

// Address spaces in pointer TYPES.  A pointer that reaches a kernel's inner code through a struct in LDS or a function boundary is
// a generic one, and a generic (flat) access counts on the LDS counter as well as on the vector-memory one: every LDS read after
// it -- a field of an LDS-resident record, an entry of the geometry table -- waits for it to RETURN (s_waitcnt vmcnt(0)
// lgkmcnt(0)), which turns loads that could travel together into a chain of round trips.  With the address space in the type the
// ISA shows global_load / global_atomic / ds_* and only real dependences wait.
#define KH_AS_GLOBAL __attribute__((address_space(1)))
#define KH_AS_LDS __attribute__((address_space(3)))
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));   // (HIP's uint2 / uint4 are structs: no copies across address spaces)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// 26-neighbourhood in the order of dijkstra_invalidation.hpp:60-124
__host__ __device__ constexpr inline void dir_delta(int i, int& dx, int& dy, int& dz) {
  // packed table: 2 bits per component (0 -> -1, 1 -> 0, 2 -> +1)
  constexpr int8_t T[26][3] = {
      {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
      {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0},
      {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1},
      {-1, 0, -1}, {-1, 0, 1}, {1, 0, -1}, {1, 0, 1},
      {-1, -1, -1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}};
  dx = T[i][0];
  dy = T[i][1];
  dz = T[i][2];
}

// voxel_connectivity_graph bit of direction i (directions in the order of dijkstra_invalidation.hpp:60-124, bits as the reference
// reads them at dijkstra_invalidation.hpp:152-190 -- cc3d's layout): -x 1, +x 0, -y 3, +y 2, -z 5, +z 4, xy diagonals 9 7 8 6,
// yz diagonals 17 13 16 12, xz diagonals 15 11 14 10, corners 25 24 23 21 22 20 19 18.
__host__ __device__ inline uint32_t graph_to_directions(uint32_t gw) {
  constexpr int BIT[26] = {1, 0, 3, 2, 5, 4, 9, 7, 8, 6, 17, 13, 16, 12, 15, 11, 14, 10, 25, 24, 23, 21, 22, 20, 19, 18};
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 26; i++) m |= ((gw >> BIT[i]) & 1u) << i;
  return m;
}

struct Geometry {
  int32_t sx, sy, sz;
  int32_t sxy;          // sx*sy (volumes are < 2^32 voxels, slices < 2^31)
  int32_t off[26];      // linear offset of neighbour i
  float w[26];          // centre-to-centre length of neighbour i (f32, no contraction)
  float wx, wy, wz;
};

// host: fill a Geometry (edge lengths as dijkstra_invalidation.hpp:45-52)
void make_geometry(Geometry& g, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz);

}  // namespace kh
