/* Developer experiment (round 5): the "decision-bit" form of the libstdc++ heap emulation, checked on the CPU before it was written
 * for the wave (experiments/bitheap_r5.patch: built, bit exact on the GPU, measured, and taken out again -- DESIGN.md 3.4.7).
 *
 * libstdc++'s pop walks the hole from the root to a LEAF along the smaller child (ties: left) whatever the keys are, then pushes the
 * former last element up from there.  Which child is the smaller one is ONE BIT per internal node, and a pop or a push changes that
 * bit only for the parents of the few slots whose content it changes.  Keeping the bits (here: bit[i] = 1 iff the right child of
 * i exists and right.key < left.key) makes the walk independent of the keys: the wave reads one 64-bit word per six levels instead
 * of 126 speculative nodes, and the nodes on the path and their siblings are loaded in ONE parallel round trip afterwards.
 *
 * This file states the update rules the kernel uses and checks them against a literal transcription of
 * bits/stl_heap.h (__push_heap / __adjust_heap, comparator `a.dist >= b.dist` of dijkstra_invalidation.hpp:233-237) on random
 * streams with heavy ties: identical arrays after every operation, and every bit equal to its definition.
 * Build and run: gcc -O2 -o /tmp/bitheap_sim experiments/bitheap_sim.c && /tmp/bitheap_sim */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CAP (1 << 20)
typedef struct { uint32_t key, id; } node;

/* ---- reference: bits/stl_heap.h */
static node ra[CAP]; static size_t rn;
static void ref_push(node x) {
  size_t hole = rn++;
  while (hole > 0) { size_t p = (hole - 1) / 2; if (!(ra[p].key >= x.key)) break; ra[hole] = ra[p]; hole = p; }
  ra[hole] = x;
}
static void ref_pop(void) {
  size_t len = rn;
  if (len > 1) {
    len--;
    node value = ra[len];
    size_t hole = 0, child = 0;
    while (child < (len - 1) / 2) { child = 2 * (child + 1); if (ra[child].key >= ra[child - 1].key) child--; ra[hole] = ra[child]; hole = child; }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); ra[hole] = ra[child - 1]; hole = child - 1; }
    while (hole > 0) { size_t p = (hole - 1) / 2; if (!(ra[p].key >= value.key)) break; ra[hole] = ra[p]; hole = p; }
    ra[hole] = value;
  }
  rn--;
}

/* ---- the decision-bit heap */
static node a[CAP]; static size_t n; static uint8_t bit[CAP];
static long n_bit_writes;
/* bit of parent p after the content of one of its children changed; len = current heap size */
static void fix_bit(size_t p, size_t len) {
  const size_t l = 2 * p + 1, r = l + 1;
  bit[p] = (uint8_t)((r < len) && (a[r].key < a[l].key));
  n_bit_writes++;
}
static void bh_push(node x) {
  const size_t pos = n++;
  /* ancestors with key >= x.key form a prefix of the chain leaf -> root */
  size_t m = 0, c = pos;
  while (c > 0 && a[(c - 1) / 2].key >= x.key) { c = (c - 1) / 2; m++; }
  /* the chain nodes c_0 = pos, c_1, ..., c_m : c_{g-1} := old c_g, c_m := x */
  size_t hole = pos;
  for (size_t g = 0; g < m; g++) { const size_t p = (hole - 1) / 2; a[hole] = a[p]; hole = p; }
  a[hole] = x;
  /* bits: the parents of the changed slots c_0 .. c_m */
  c = pos;
  for (size_t e = 0; e <= m; e++) { if (c == 0) break; fix_bit((c - 1) / 2, n); c = (c - 1) / 2; }
}
static void bh_pop(void) {
  if (n <= 1) { n = 0; return; }
  const size_t len = n - 1;
  n = len;
  const node last = a[len];
  /* the heap shrinks first: if the removed slot was a right child its parent has only the left one now */
  if ((len & 1) == 0) bit[(len - 2) / 2] = 0;
  /* walk to a leaf by the bits alone */
  size_t path[64]; int D = 0;
  path[0] = 0;
  for (size_t h = 0; 2 * h + 1 < len;) { h = 2 * h + 1 + bit[h]; path[++D] = h; }
  /* path nodes with key < last.key move up one slot, last lands behind them */
  int m = 0;
  while (m < D && a[path[m + 1]].key < last.key) m++;
  for (int e = 0; e < m; e++) a[path[e]] = a[path[e + 1]];
  a[path[m]] = last;
  /* bits: parents of the changed slots path[1..m] (path[0] has no parent) */
  for (int e = 1; e <= m; e++) fix_bit(path[e - 1], len);
}

static int check(const char* what, long step) {
  if (n != rn) { printf("%s step %ld: size %zu vs %zu\n", what, step, n, rn); return 1; }
  for (size_t i = 0; i < n; i++) if (a[i].key != ra[i].key || a[i].id != ra[i].id) { printf("%s step %ld: slot %zu differs\n", what, step, i); return 1; }
  for (size_t p = 0; 2 * p + 1 < n; p++) {
    const size_t l = 2 * p + 1, r = l + 1;
    const uint8_t want = (uint8_t)((r < n) && (a[r].key < a[l].key));
    if (bit[p] != want) { printf("%s step %ld: bit of %zu is %d, should be %d (n = %zu)\n", what, step, p, bit[p], want, n); return 1; }
  }
  return 0;
}

int main(void) {
  uint64_t s = 88172645463325252ull;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
  long ops = 0;
  for (int trial = 0; trial < 400; trial++) {
    n = rn = 0; memset(bit, 0xAA, 4096);     /* stale bits must not matter */
    const int nkeys = 1 + (int)(RND() % (trial % 3 == 0 ? 4 : 60));      /* heavy ties */
    const long steps = 200 + (long)(RND() % 3000);
    uint32_t id = 0, base = 0;
    for (long t = 0; t < steps; t++) {
      const int burst = 1 + (int)(RND() % 14);
      if (n == 0 || RND() % 100 < 45) {
        for (int b = 0; b < burst && n < CAP - 1; b++) { node x = {base + (uint32_t)(RND() % nkeys), id++}; ref_push(x); bh_push(x); ops++; if ((trial & 7) == 0 && check("push", t)) return 1; }
      } else {
        ref_pop(); bh_pop(); ops++;
        if (RND() % 50 == 0) base++;            /* keys drift upwards like the flood's levels */
      }
      if (check("op", t)) return 1;
    }
    while (n) { ref_pop(); bh_pop(); ops++; if (check("drain", (long)n)) return 1; }
  }
  printf("bitheap_sim: %ld operations, arrays and bits identical to bits/stl_heap.h throughout; %.2f bit writes per operation\n",
         ops, (double)n_bit_writes / (double)ops);
  return 0;
}
