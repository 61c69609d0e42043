// Developer self-test of the wave code of the invalidation heap (kimimaro_amd/csrc/trace.hip) on the GPU: a script of pushes, batched
// "fires" and pops runs through heap_push_wave / the batched append / heap_pop_wave on one wave; the host compares the pop sequence
// and the final array with a literal transcription of bits/stl_heap.h (tools/selftest/heap_selftest.py).
#include "../../kimimaro_amd/csrc/trace.hip"

namespace kh {
// script: words.  0 = pop; 1, key, id = push; 2, mask_lo, mask_hi, key[26], id0 = a fired voxel (ids id0, id0+1, ... in lane order)
__global__ __launch_bounds__(64) void heap_selftest_kernel(hnode_t* nodes, uint32_t cap, const uint32_t* script, uint32_t nwords,
                                                           uint32_t* popped, uint32_t* out_n, uint32_t wlds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  kh_label_t task;
  task.heap_offset = 0;
  task.heap_capacity = cap;
  Heap h;
  heap_setup(h, nodes, &task, lds, wlds, lane);
  uint32_t npop = 0;
  for (uint32_t i = 0; i < nwords;) {
    const uint32_t op = script[i];
    if (op == 0u) {
      if (h.n > 0) {
        const hnode_t top = *h.root;
        if (lane == 0) popped[npop] = top.y;
        npop++;
        heap_pop_wave(h, lane);
      }
      i += 1;
    } else if (op == 1u) {
      heap_push_wave(h, script[i + 1], script[i + 2], 7u, lane);
      i += 3;
    } else {
      unsigned long long m = (unsigned long long)script[i + 1] | ((unsigned long long)script[i + 2] << 32);
      const uint32_t ndb = lane < 26 ? script[i + 3 + lane] : 0u;
      const uint32_t id0 = script[i + 29];
      const uint32_t q = id0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      const uint32_t si = 7u;
      // ---- the batched append of invalidate_ball (kept textually in step with it)
      while (m) {
        const uint32_t base = h.n;
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (base < 64u || base + cnt > h.cap) {
          const int k = __ffsll((long long)m) - 1;
          m &= m - 1;
          heap_push_wave(h, rdlane_u32(ndb, k), rdlane_u32(q, k), si, lane);
          continue;
        }
        const bool mine = (m >> lane) & 1ull;
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        const uint32_t leaf = base + (uint32_t)__popcll(below);
        const uint32_t par = (leaf - 1u) >> 1;
        const uint32_t kp = h.node[mine ? par : 0u].x;
        const bool right = mine && (leaf & 1u) == 0u;
        const uint32_t kold = h.node[right && leaf == base ? leaf - 1u : 0u].x;
        const int prev = below ? 63 - __clzll((long long)below) : 0;
        const uint32_t kprev = (uint32_t)__shfl((int)ndb, prev);
        const bool stay = mine && kp < ndb;
        const unsigned long long climbers = m & ~ballot64(stay);
        const int c = climbers ? __ffsll((long long)climbers) - 1 : 64;
        const unsigned long long run = c < 64 ? (m & ((1ull << c) - 1ull)) : m;
        const bool inrun = (run >> lane) & 1ull;
        if (inrun) {
          const hnode_t fresh = {ndb, q, si, 0u};
          h.node[leaf] = fresh;
        }
        const uint32_t kl = leaf == base ? kold : kprev;
        heap_set_bit(h, inrun, par, right && ndb < kl);
        const uint32_t nrun = (uint32_t)__popcll(run);
        h.n = base + nrun;
        m &= ~run;
        if (c < 64) {
          m &= ~(1ull << c);
          heap_push_wave(h, rdlane_u32(ndb, c), rdlane_u32(q, c), si, lane);
        }
      }
      i += 30;
    }
  }
  if (lane == 0) { out_n[0] = h.n; out_n[1] = npop; }
}
}  // namespace kh

extern "C" int heap_selftest(void* nodes, uint32_t cap, const uint32_t* script, uint32_t nwords, uint32_t* popped, uint32_t* out_n,
                             uint32_t wlds) {
  const size_t lds = 16 + (size_t)wlds * 8;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&kh::heap_selftest_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 2;
  hipLaunchKernelGGL(kh::heap_selftest_kernel, dim3(1), dim3(64), lds, 0, (kh::hnode_t*)nodes, cap, script, nwords, popped, out_n, wlds);
  if (hipGetLastError() != hipSuccess) return 3;
  return hipDeviceSynchronize() == hipSuccess ? 0 : 4;
}
