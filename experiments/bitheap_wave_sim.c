/* Developer experiment (round 5): lane-level CPU emulation of the wave code of the decision-bit heap (experiments/bitheap_r5.patch:
 * heap_push_wave, heap_pop_wave and the batched append of invalidate_ball), statement by statement over 64 lanes, against a
 * literal transcription of bits/stl_heap.h.  Memory is bounds-checked: an out-of-range access aborts with the statement.
 * gcc -O2 -o /tmp/bitheap_wave_sim experiments/bitheap_wave_sim.c && /tmp/bitheap_wave_sim */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CAPN (1u << 19)
typedef struct { uint32_t x, y, z, w; } hnode;
static hnode ra[CAPN]; static size_t rn;
static void ref_push(hnode v) { size_t hole = rn++; while (hole > 0) { size_t p = (hole - 1) / 2; if (!(ra[p].x >= v.x)) break; ra[hole] = ra[p]; hole = p; } ra[hole] = v; }
static void ref_pop(void) {
  size_t len = rn;
  if (len > 1) { len--; hnode value = ra[len]; size_t hole = 0, child = 0;
    while (child < (len - 1) / 2) { child = 2 * (child + 1); if (ra[child].x >= ra[child - 1].x) child--; ra[hole] = ra[child]; hole = child; }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); ra[hole] = ra[child - 1]; hole = child - 1; }
    while (hole > 0) { size_t p = (hole - 1) / 2; if (!(ra[p].x >= value.x)) break; ra[hole] = ra[p]; hole = p; }
    ra[hole] = value; }
  rn--;
}
/* ---- device state */
static hnode node[CAPN]; static uint64_t W[300000]; static hnode root; static uint32_t hn, hcap = CAPN - 64, wlds = 65, wcap = 300000;
static uint64_t pmask[64], ppat[64];
#define NODE(i, why) (*({ uint32_t i_ = (i); if (i_ >= CAPN) { printf("OOB node %u at %s\n", i_, why); exit(1); } &node[i_]; }))
static uint64_t* WP(uint32_t flat, const char* why) { if (flat >= wcap) { printf("OOB word %u at %s\n", flat, why); exit(1); } return &W[flat]; }
static void bit_loc(uint32_t q, uint32_t* flat, uint32_t* rel) {
  const uint32_t dq = 31u - (uint32_t)__builtin_clz(q + 1u), cq = (dq * 43u) >> 8, sh = dq - 6u * cq, root1 = (q + 1u) >> sh;
  *rel = (q + 1u) - (root1 << sh) + (1u << sh) - 1u;
  const uint32_t first1 = 1u << (6u * cq);
  *flat = (first1 - 1u) / 63u + (root1 - first1);
}
/* per-lane: first all ANDs, then all ORs (the order of the two instructions) */
static void set_bits(const int* on, const uint32_t* q, const int* value) {
  uint32_t flat[64], rel[64];
  for (int l = 0; l < 64; l++) bit_loc(on[l] ? q[l] : 0u, &flat[l], &rel[l]);
  for (int l = 0; l < 64; l++) if (on[l]) *WP(flat[l], "set_bit and") &= ~(1ull << rel[l]);
  for (int l = 0; l < 64; l++) if (on[l] && value[l]) *WP(flat[l], "set_bit or") |= 1ull << rel[l];
}
static int ffsll_(uint64_t v) { return v ? __builtin_ctzll(v) + 1 : 0; }
static int push_wave(uint32_t kbits, uint32_t vox, uint32_t src) {
  if (hn >= hcap) return 0;
  const uint32_t pos = hn++, len = pos + 1u;
  uint32_t ce[64], ai[64], sib[64], sk[64]; int chain[64], valid[64], sib_ok[64]; hnode a[64]; uint64_t climb = 0;
  for (int lane = 0; lane < 64; lane++) {
    const int sh = lane < 31 ? lane : 31;
    ce[lane] = ((pos + 1u) >> sh) - 1u;
    chain[lane] = (lane < 32) && ((pos + 1u) >> sh) >= 1u;
    valid[lane] = chain[lane] && ce[lane] != 0u;
    ai[lane] = valid[lane] ? (ce[lane] - 1u) >> 1 : 0u;
    a[lane] = NODE(ai[lane], "push anc");
    sib[lane] = (ce[lane] & 1u) ? ce[lane] + 1u : ce[lane] - 1u;
    sib_ok[lane] = valid[lane] && sib[lane] < len;
    sk[lane] = NODE(sib_ok[lane] ? sib[lane] : 0u, "push sib").x;
    if (valid[lane] && a[lane].x >= kbits) climb |= 1ull << lane;
  }
  const int m = ffsll_(~climb) - 1;
  const hnode fresh = {kbits, vox, src, 0u};
  int on[64], bitv[64];
  hnode val[64];
  for (int lane = 0; lane < 64; lane++) val[lane] = lane < m ? a[lane] : fresh;
  for (int lane = 0; lane < 64; lane++) if (chain[lane] && lane <= m) { NODE(ce[lane], "push store") = val[lane]; if (ce[lane] == 0u) root = val[lane]; }
  for (int lane = 0; lane < 64; lane++) {
    const uint32_t mykey = val[lane].x; const int right_is_me = (ce[lane] & 1u) == 0u;
    bitv[lane] = sib_ok[lane] && (right_is_me ? mykey < sk[lane] : sk[lane] < mykey);
    on[lane] = valid[lane] && lane <= m;
  }
  set_bits(on, ai, bitv);
  return 1;
}
static uint64_t chunk_word(int c, uint32_t r) { const uint32_t first1 = 1u << (6 * c); return *WP((first1 - 1u) / 63u + (r + 1u - first1), "chunk word"); }
static void pop_wave(void) {
  const uint32_t len = hn - 1u; hn = len;
  if (len == 0) return;
  const hnode last = NODE(len, "pop last");
  uint32_t sflat = 0xFFFFFFFFu, srel = 0;
  if ((len & 1u) == 0u) { bit_loc((len - 2u) >> 1, &sflat, &srel); *WP(sflat, "shrink") &= ~(1ull << srel); }
  uint32_t myp[64] = {0}; int mine[64] = {0};
  uint32_t r = 0;
  for (int c = 0; c < 5; c++) {
    if (2u * r + 1u >= len) break;
    uint64_t w = chunk_word(c, r);
    uint64_t hit = 0;
    for (int l = 0; l < 64; l++) if (((w ^ ppat[l]) & pmask[l]) == 0ull) hit |= 1ull << l;
    const uint32_t t1 = 64u + (uint32_t)(ffsll_(hit) - 1);
    uint32_t idx[64]; int ex[64]; uint32_t cnt = 0;
    for (int lane = 0; lane < 64; lane++) {
      const int d = lane - 6 * c; const int inch = d >= 1 && d <= 6; const int dd = inch ? d : 1;
      idx[lane] = ((r + 1u) << dd) - 1u + ((t1 >> (6 - dd)) - (1u << dd));
      ex[lane] = inch && idx[lane] < len;
      if (inch) { myp[lane] = idx[lane]; mine[lane] = ex[lane]; }
      cnt += ex[lane];
    }
    if (cnt < 6u) break;
    r = idx[6 * c + 6];
  }
  hnode nd[64]; uint32_t sib[64], sk[64]; int sib_ok[64]; uint32_t m = 0;
  for (int lane = 0; lane < 64; lane++) {
    nd[lane] = NODE(mine[lane] ? myp[lane] : 0u, "pop path");
    sib[lane] = (myp[lane] & 1u) ? myp[lane] + 1u : myp[lane] - 1u;
    sib_ok[lane] = mine[lane] && sib[lane] < len;
    sk[lane] = NODE(sib_ok[lane] ? sib[lane] : 0u, "pop sib").x;
    if (mine[lane] && nd[lane].x < last.x) m++;
  }
  int mover[64], bitv[64]; uint32_t par[64];
  for (int lane = 0; lane < 64; lane++) { mover[lane] = mine[lane] && (uint32_t)lane <= m; par[lane] = (myp[lane] - 1u) >> 1; }
  for (int lane = 0; lane < 64; lane++) if (mover[lane]) { NODE(par[lane], "pop move") = nd[lane]; if (lane == 1) root = nd[lane]; }
  for (int lane = 0; lane < 64; lane++) if ((uint32_t)lane == m) { NODE(myp[lane], "pop last store") = last; if (m == 0u) root = last; }
  for (int lane = 0; lane < 64; lane++) {
    const uint32_t nxt = lane < 63 ? nd[lane + 1].x : nd[lane].x;
    const uint32_t mykey = (uint32_t)lane == m ? last.x : nxt;
    const int right_is_me = (myp[lane] & 1u) == 0u;
    bitv[lane] = sib_ok[lane] && (right_is_me ? mykey < sk[lane] : sk[lane] < mykey);
  }
  set_bits(mover, par, bitv);
}
/* the pushes of one fired voxel: keys k[0..cnt) in lane order (lanes = set bits of m) */
static void fire(uint64_t m, const uint32_t* ndb, uint32_t* idp) {
  while (m) {
    const uint32_t base = hn, cnt = (uint32_t)__builtin_popcountll(m);
    if (base < 64u || base + cnt > hcap) { const int k = ffsll_(m) - 1; m &= m - 1; push_wave(ndb[k], (*idp)++, 7); continue; }
    uint32_t leaf[64], par[64], kp[64], kold[64], kprev[64]; int mine[64], right[64], stay[64]; uint64_t stayb = 0;
    for (int lane = 0; lane < 64; lane++) {
      mine[lane] = (m >> lane) & 1ull;
      const uint64_t below = m & ((1ull << lane) - 1ull);
      leaf[lane] = base + (uint32_t)__builtin_popcountll(below);
      par[lane] = (leaf[lane] - 1u) >> 1;
      kp[lane] = NODE(mine[lane] ? par[lane] : 0u, "fire parent").x;
      right[lane] = mine[lane] && (leaf[lane] & 1u) == 0u;
      kold[lane] = NODE(right[lane] && leaf[lane] == base ? leaf[lane] - 1u : 0u, "fire left sibling").x;
      const int prev = below ? 63 - __builtin_clzll(below) : 0;
      kprev[lane] = ndb[prev];
      stay[lane] = mine[lane] && kp[lane] < ndb[lane];
      if (stay[lane]) stayb |= 1ull << lane;
    }
    const uint64_t climbers = m & ~stayb;
    const int c = climbers ? ffsll_(climbers) - 1 : 64;
    const uint64_t run = c < 64 ? (m & ((1ull << c) - 1ull)) : m;
    int inrun[64], bitv[64];
    uint32_t ids[64];
    for (int lane = 0; lane < 64; lane++) { inrun[lane] = (run >> lane) & 1ull; if (inrun[lane]) ids[lane] = (*idp)++; }
    for (int lane = 0; lane < 64; lane++) if (inrun[lane]) { hnode fresh = {ndb[lane], ids[lane], 7, 0}; NODE(leaf[lane], "fire store") = fresh; }
    for (int lane = 0; lane < 64; lane++) { const uint32_t kl = leaf[lane] == base ? kold[lane] : kprev[lane]; bitv[lane] = right[lane] && ndb[lane] < kl; }
    set_bits(inrun, par, bitv);
    const uint32_t nrun = (uint32_t)__builtin_popcountll(run);
    hn = base + nrun;
    m &= ~run;
    if (c < 64) { m &= ~(1ull << c); push_wave(ndb[c], (*idp)++, 7); }
  }
}
static int check(const char* what, long t) {
  if (hn != rn) { printf("%s %ld: size %u vs %zu\n", what, t, hn, rn); return 1; }
  for (size_t i = 0; i < hn; i++) if (node[i].x != ra[i].x || node[i].y != ra[i].y) { printf("%s %ld: slot %zu differs (n=%u)\n", what, t, i, hn); return 1; }
  if (hn && (root.x != ra[0].x || root.y != ra[0].y)) { printf("%s %ld: root copy differs\n", what, t); return 1; }
  return 0;
}
int main(void) {
  for (int j = 0; j < 64; j++) { uint64_t mask = 0, pat = 0; uint32_t t = 63u + j;
    for (int d = 0; d < 6; d++) { uint32_t a = ((t + 1u) >> (6 - d)) - 1u, nx = ((t + 1u) >> (5 - d)) - 1u; mask |= 1ull << a; pat |= (uint64_t)((nx & 1u) == 0u) << a; }
    pmask[j] = mask; ppat[j] = pat; }
  uint64_t s = 88172645463325252ull;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
  long ops = 0;
  for (int trial = 0; trial < 60; trial++) {
    hn = 0; rn = 0; memset(W, 0x5A, sizeof(W));
    const int nkeys = 1 + (int)(RND() % (trial % 3 == 0 ? 4 : 60));
    const long steps = 500 + (long)(RND() % (trial < 50 ? 6000 : 120000));
    uint32_t id = 0, base = 0, ndb[64];
    for (int i = 0; i < 1 + (int)(RND() % 40); i++) { hnode x = {0, id, 7, 0}; ref_push(x); push_wave(0, id++, 7); ops++; }
    if (check("init", 0)) return 1;
    for (long t = 0; t < steps && hn; t++) {
      ref_pop(); pop_wave(); ops++;
      if (check("pop", t)) return 1;
      if (hn < 400000u && RND() % 100 < (trial < 50 ? 30 : 55)) {          /* a live pop: up to 26 pushes in lane order */
        uint64_t m = 0;
        for (int l = 0; l < 26; l++) if (RND() % 100 < 45) { m |= 1ull << l; ndb[l] = base + (uint32_t)(RND() % nkeys); }
        uint32_t id2 = id;
        for (int l = 0; l < 26; l++) if ((m >> l) & 1) { hnode x = {ndb[l], id2++, 7, 0}; ref_push(x); ops++; }
        fire(m, ndb, &id);
        if (id != id2) { printf("id mismatch\n"); return 1; }
        if (check("fire", t)) return 1;
      }
      if (RND() % 40 == 0) base++;
    }
  }
  printf("bitheap_wave_sim: %ld operations, heap sizes up to the hundreds of thousands: identical to bits/stl_heap.h\n", ops);
  return 0;
}
