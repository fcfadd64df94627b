// hostcheck_logic.h -- serial forms of the length-limited coding and of the dynamic-header plan, used ONLY by
// tests/hostcheck (CPU cross-checks against the oracle).  The product runs the wave-parallel forms
// (zip-ada_amd/csrc/zada_llhc_wave.h, k_block_analyze).
#pragma once
#include "../../zip-ada_amd/csrc/zada_logic.h"

namespace zada {

// ----- Huffman.Encoding.Length_Limited_Coding, lane-serial form -----
// huffman-encoding-length_limited_coding.adb:46-280.  Explicit stacks replace the recursion of
// Boundary_PM (:131-163) and Quick_sort (:196-223); node pool with the reference's garbage
// collection (:97-122).  Scratch lives in LDS on the GPU.
struct LlhcScratch {
  uint32_t w[480];          // pool weight
  uint16_t cnt[480];        // pool count
  int16_t tail[480];        // pool tail (-1 = null)
  uint8_t inuse[480];
  uint32_t leafw[288];
  uint16_t leafsym[288];
  int16_t lists[15][2];
  uint8_t stk[40];          // pending Boundary_PM calls (list index)
  uint16_t qlo[24], qn[24]; // quicksort stack
};

ZADA_HD int llhc_get_free_node(LlhcScratch *S, int pool_last, int max_bits, bool use_lists, int &pool_next) {
  for (;;) {
    if (pool_next > pool_last) {
      for (int i = 0; i <= pool_last; i++) S->inuse[i] = 0;
      if (use_lists) {
        for (int i = 0; i < max_bits * 2; i++) {
          int node = S->lists[i >> 1][i & 1];
          while (node >= 0) { S->inuse[node] = 1; node = S->tail[node]; }
        }
      }
      pool_next = 0;
    }
    if (!S->inuse[pool_next]) break;
    pool_next++;
  }
  pool_next++;
  return pool_next - 1;
}

// freq[n] -> bl[n].  n <= 288, max_bits <= 15.
template <typename FreqT>
ZADA_HD void llhc_serial(const FreqT *freq, int n, int max_bits, uint8_t *bl, LlhcScratch *S) {
  int ns = 0;
  for (int a = 0; a < n; a++) {
    bl[a] = 0;
    if (freq[a] > 0) { S->leafw[ns] = (uint32_t)freq[a]; S->leafsym[ns] = (uint16_t)a; ns++; }
  }
  if (ns == 0) return;
  if (ns == 1) { bl[S->leafsym[0]] = 1; return; }
  // Quick_sort :196-223 (Hoare partition, pivot a(n/2), compares weights only)
  {
    int sp = 0;
    S->qlo[0] = 0; S->qn[0] = (uint16_t)ns; sp = 1;
    while (sp > 0) {
      sp--;
      int lo = S->qlo[sp], m = S->qn[sp];
      while (m >= 2) {
        uint32_t p = S->leafw[lo + m / 2];
        int i = 0, j = m - 1;
        for (;;) {
          while (S->leafw[lo + i] < p) i++;
          while (p < S->leafw[lo + j]) j--;
          if (i >= j) break;
          uint32_t tw = S->leafw[lo + i]; S->leafw[lo + i] = S->leafw[lo + j]; S->leafw[lo + j] = tw;
          uint16_t ts = S->leafsym[lo + i]; S->leafsym[lo + i] = S->leafsym[lo + j]; S->leafsym[lo + j] = ts;
          i++; j--;
        }
        // recurse on (lo, i) and (lo + i, m - i): push the larger, iterate on the smaller
        int n1 = i, n2 = m - i;
        if (n1 > n2) { if (n1 >= 2) { S->qlo[sp] = (uint16_t)lo; S->qn[sp] = (uint16_t)n1; sp++; } lo = lo + i; m = n2; }
        else { if (n2 >= 2) { S->qlo[sp] = (uint16_t)(lo + i); S->qn[sp] = (uint16_t)n2; sp++; } m = n1; }
      }
    }
  }
  const int pool_last = 2 * max_bits * (max_bits + 1) - 1;
  int pool_next = 0;
  for (int i = 0; i <= pool_last; i++) { S->inuse[i] = 0; S->tail[i] = -1; }
  // Init_Lists :167-174
  {
    int node0 = llhc_get_free_node(S, pool_last, max_bits, false, pool_next);
    S->w[node0] = S->leafw[0]; S->cnt[node0] = 1; S->tail[node0] = -1; S->inuse[node0] = 1;
    int node1 = llhc_get_free_node(S, pool_last, max_bits, false, pool_next);
    S->w[node1] = S->leafw[1]; S->cnt[node1] = 2; S->tail[node1] = -1; S->inuse[node1] = 1;
    for (int i = 0; i < max_bits; i++) { S->lists[i][0] = (int16_t)node0; S->lists[i][1] = (int16_t)node1; }
  }
  const int runs = 2 * ns - 4;
  for (int r = 1; r <= runs; r++) {
    // Boundary_PM (max_bits - 1, final = (r == runs)), recursion unrolled on S->stk
    int sp = 0;
    S->stk[sp++] = (uint8_t)(max_bits - 1);
    bool top = true;
    while (sp > 0) {
      int index = S->stk[--sp];
      bool fin = top && (r == runs);
      top = false;
      int lastcount = S->cnt[S->lists[index][1]];
      if (index == 0 && lastcount >= ns) continue;
      int newchain = llhc_get_free_node(S, pool_last, max_bits, true, pool_next);
      int oldchain = S->lists[index][1];
      S->lists[index][0] = (int16_t)oldchain; S->lists[index][1] = (int16_t)newchain;
      if (index == 0) {
        S->w[newchain] = S->leafw[lastcount]; S->cnt[newchain] = (uint16_t)(lastcount + 1); S->tail[newchain] = -1; S->inuse[newchain] = 1;
      } else {
        uint32_t sum = S->w[S->lists[index - 1][0]] + S->w[S->lists[index - 1][1]];
        if (lastcount < ns && sum > S->leafw[lastcount]) {
          S->w[newchain] = S->leafw[lastcount]; S->cnt[newchain] = (uint16_t)(lastcount + 1); S->tail[newchain] = S->tail[oldchain]; S->inuse[newchain] = 1;
        } else {
          S->w[newchain] = sum; S->cnt[newchain] = (uint16_t)lastcount; S->tail[newchain] = S->lists[index - 1][1]; S->inuse[newchain] = 1;
          if (!fin) { S->stk[sp++] = (uint8_t)(index - 1); S->stk[sp++] = (uint8_t)(index - 1); }
        }
      }
    }
  }
  // Extract_Bit_Lengths :180-189
  for (int node = S->lists[max_bits - 1][1]; node >= 0; node = S->tail[node])
    for (int i = 0; i < (int)S->cnt[node]; i++) bl[S->leafsym[i]]++;
}

// Builds the plan (cost_analysis = True path).  ll[288], dd[32] = code lengths.
ZADA_HD void header_plan(const uint8_t *ll, const uint8_t *dd, HeaderPlan *hp, LlhcScratch *S) {
  int max_ll = 0, max_d = 0, idx = 0;
  for (int a = 287; a >= 0; a--) if (ll[a] > 0) { max_ll = a; break; }
  for (int a = 31; a >= 0; a--) if (dd[a] > 0) { max_d = a; break; }
  for (int a = 0; a <= max_ll; a++) hp->cs_bl[idx++] = ll[a];
  for (int a = 0; a <= max_d; a++) hp->cs_bl[idx++] = dd[a];
  hp->last_cs_bl = (uint16_t)idx;
  hp->hlit_m257 = (uint8_t)(max_ll - 256);
  hp->hdist_m1 = (uint8_t)max_d;
  for (int a = 0; a < 19; a++) hp->truc_freq[a] = 0;
  uint32_t *tf = hp->truc_freq;
  header_rle_walk(hp->cs_bl, idx, [tf](int x, uint32_t) { tf[x]++; });
  llhc_serial(hp->truc_freq, 19, 7, hp->truc_bl, S);
  int anz = 3;
  for (int a = 0; a <= 18; a++) if (a > anz && hp->truc_bl[header_perm(a)] > 0) anz = a;
  hp->a_non_zero = (uint8_t)anz;
  uint32_t bits = 14 + (uint32_t)(1 + anz) * 3;
  for (int a = 0; a <= 18; a++) bits += hp->truc_freq[a] * (uint32_t)(hp->truc_bl[a] + header_extra_bits(a));
  hp->bits = bits;
}


}  // namespace zada
