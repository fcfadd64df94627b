// zada_llhc_wave.h -- Huffman.Encoding.Length_Limited_Coding on ONE WAVE (64 lanes), device only.
//
// Same result, bit for bit, as the reference's sequential procedure
// (zip_lib/huffman-encoding-length_limited_coding.adb:46-280), obtained from two observations that
// tests/hostcheck (hc_llhc_pm) checks against the oracle on tens of thousands of inputs:
//
//  * Quick_sort (:196-223) is a Hoare partition around a(n/2) comparing weights only.  Its result
//    on a sub-array has a closed form: with I = positions holding weight >= pivot (ascending) and
//    J = positions holding weight <= pivot (descending), exactly the pairs (I[k], J[k]) with
//    I[k] < J[k] are swapped, and the split point is min(I[K], J[K-1]).  That is a few ballots per
//    64 elements, so the reference's tie-breaking order is reproduced without running its loop.
//  * Boundary_PM (:131-163) builds, lazily, the lists of the classic package-merge algorithm in
//    which a package precedes a leaf of equal weight (":148 sum > leaves (lastcount).weight").
//    The lists are built here level by level with parallel merges (one merge-path chunk per lane),
//    each level keeping one bit per item (leaf / package); Extract_Bit_Lengths (:180-189) becomes a
//    prefix popcount per level.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zada {

constexpr int LLHC_WAVE_SCRATCH = 4736;    // bytes of LDS per instance

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// serial reference quicksort on a small sub-array (run by single lanes, many sub-arrays at once)
__device__ inline void small_qsort(uint32_t *lw, uint16_t *ls, int lo0, int m0, uint16_t *stk /* 8 entries */) {
  int sp = 0, lo = lo0, m = m0;
  for (;;) {
    while (m >= 2) {
      const uint32_t p = lw[lo + m / 2];
      int i = 0, j = m - 1;
      for (;;) {
        while (lw[lo + i] < p) i++;
        while (p < lw[lo + j]) j--;
        if (i >= j) break;
        uint32_t tw = lw[lo + i]; lw[lo + i] = lw[lo + j]; lw[lo + j] = tw;
        uint16_t ts = ls[lo + i]; ls[lo + i] = ls[lo + j]; ls[lo + j] = ts;
        i++; j--;
      }
      const int n1 = i, n2 = m - i;
      if (n1 > n2) { if (n1 >= 2) { stk[2 * sp] = (uint16_t)lo; stk[2 * sp + 1] = (uint16_t)n1; sp++; } lo += i; m = n2; }
      else { if (n2 >= 2) { stk[2 * sp] = (uint16_t)(lo + i); stk[2 * sp + 1] = (uint16_t)n2; sp++; } m = n1; }
    }
    if (sp == 0) break;
    sp--; lo = stk[2 * sp]; m = stk[2 * sp + 1];
  }
}

// freq[n] (LDS or global, read once) -> bl[n].  All 64 lanes of the wave must call it together.
template <int max_bits>
__device__ __forceinline__ void llhc_wave(const uint32_t *freq, int n, uint8_t *bl, uint8_t *scratch, int lane) {
  uint32_t *lw = (uint32_t *)scratch;                 // 288 leaf weights
  uint16_t *ls = (uint16_t *)(lw + 288);              // 288 leaf symbols
  uint8_t *u = scratch + 1728;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long gt = ~((2ull << lane) - 1ull);
  // ---- leaves in symbol order (:230-235) ----
  int ns = 0;
  for (int base = 0; base < n; base += 64) {
    const int a = base + lane;
    const uint32_t f = a < n ? freq[a] : 0u;
    const bool nz = f > 0;
    const unsigned long long mask = __ballot(nz);
    if (a < n) bl[a] = 0;
    if (nz) { const int r = ns + __popcll(mask & lt); lw[r] = f; ls[r] = (uint16_t)a; }
    ns += __popcll(mask);
  }
  wave_sync();
  if (ns == 0) return;
  if (ns == 1) { if (lane == 0) bl[ls[0]] = 1; wave_sync(); return; }             // :243-246

  // ---- Quick_sort, same permutation as the reference ----
  {
    uint16_t *tmpI = (uint16_t *)u, *tmpJ = tmpI + 288;
    uint16_t *stk = tmpJ + 288;                       // 32 x (lo, m)
    uint16_t *small = stk + 64;                       // 160 x (lo, m)
    uint16_t *lstk = small + 320;                     // 64 lanes x 8
    int sp = 0, nsmall = 0;
    auto route = [&](int lo, int m) {                 // uniform; sp / nsmall advance by arithmetic (no pointer select: they stay in registers)
      const int big = m > 16 ? 1 : 0, sml = (m >= 2 && m <= 16) ? 1 : 0;
      if (lane == 0 && m >= 2) { uint16_t *q = big ? stk + 2 * sp : small + 2 * nsmall; q[0] = (uint16_t)lo; q[1] = (uint16_t)m; }
      sp += big; nsmall += sml;
    };
    route(0, ns);
    wave_sync();
    while (sp > 0) {
      sp--;
      const int lo = stk[2 * sp], m = stk[2 * sp + 1];
      const uint32_t p = lw[lo + m / 2];
      const int nchunk = (m + 63) >> 6;
      int nI = 0, nJ = 0;
      for (int c = 0; c < nchunk; c++) {
        const int idx = c * 64 + lane;
        const bool ge = idx < m && lw[lo + idx] >= p;
        const unsigned long long mask = __ballot(ge);
        if (ge) tmpI[nI + __popcll(mask & lt)] = (uint16_t)idx;
        nI += __popcll(mask);
      }
      for (int c = nchunk - 1; c >= 0; c--) {
        const int idx = c * 64 + lane;
        const bool le = idx < m && lw[lo + idx] <= p;
        const unsigned long long mask = __ballot(le);
        if (le) tmpJ[nJ + __popcll(mask & gt)] = (uint16_t)idx;
        nJ += __popcll(mask);
      }
      wave_sync();
      const int nmin = nI < nJ ? nI : nJ;
      int K = 0;
      for (int c = 0; c * 64 < nmin; c++) {
        const int k = c * 64 + lane;
        K += __popcll(__ballot(k < nmin && tmpI[k] < tmpJ[k]));
      }
      for (int k = lane; k < K; k += 64) {
        const int a = lo + tmpI[k], b = lo + tmpJ[k];
        const uint32_t tw = lw[a]; lw[a] = lw[b]; lw[b] = tw;
        const uint16_t ts = ls[a]; ls[a] = ls[b]; ls[b] = ts;
      }
      int i = 1 << 20;
      if (K < nI) i = tmpI[K];
      if (K > 0) { const int j = tmpJ[K - 1]; i = j < i ? j : i; }
      wave_sync();
      // larger half first so that the smaller one is taken next (bounded stack)
      if (i >= m - i) { route(lo, i); route(lo + i, m - i); } else { route(lo + i, m - i); route(lo, i); }
      wave_sync();
    }
    for (int t = lane; t < nsmall; t += 64) small_qsort(lw, ls, small[2 * t], small[2 * t + 1], lstk + lane * 8);
    wave_sync();
  }

  // ---- package-merge, level by level ----
  // Level l's list = merge(leaves, packages of level l-1's list).  Each lane merges one chunk of CH
  // consecutive items of the list (merge-path split, then a two-pointer walk), adds them up in
  // pairs into the next level's packages, and keeps what Extract_Bit_Lengths needs from its chunk:
  // the number of leaves before it and one leaf / package flag per item.
  {
    uint32_t *pk0 = (uint32_t *)u;                    // packages merged into the current level (<= 287)
    uint32_t *pk1 = pk0 + 288;                        // packages made from the current level
    uint32_t rec[max_bits + 1];                       // per level: leaves before the lane's chunk | flags << 16 (registers)
    int np = ns >> 1;
    for (int i = lane; i < np; i += 64) pk0[i] = lw[2 * i] + lw[2 * i + 1];
    const int CH = (((2 * ns - 1 + 63) >> 6) + 1) & ~1;  // even, 64 * CH >= the longest list (2 ns - 1 items)
    const int d0 = lane * CH;
    wave_sync();
#pragma unroll
    for (int l = 2; l <= max_bits; l++) {
      const int total = ns + np;
      int i = ns;
      uint32_t flags = 0;
      if (d0 < total) {
        // merge path: i = leaves among the first d0 items (a package precedes a leaf of equal weight)
        int lo = d0 > np ? d0 - np : 0, hi = d0 < ns ? d0 : ns;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (lw[mid] < pk0[d0 - 1 - mid]) lo = mid + 1; else hi = mid;
        }
        i = lo;
        int a = i, b = d0 - i;
        uint32_t va = a < ns ? lw[a] : 0u, vb = b < np ? pk0[b] : 0u, sum = 0;
        const int cnt = total - d0 < CH ? total - d0 : CH;
        for (int t = 0; t < cnt; t++) {
          const bool take_p = b < np && (a >= ns || vb <= va);
          const uint32_t v = take_p ? vb : va;
          if (take_p) { b++; vb = b < np ? pk0[b] : 0u; }
          else { flags |= 1u << t; a++; va = a < ns ? lw[a] : 0u; }
          if (t & 1) pk1[(d0 + t) >> 1] = sum + v; else sum = v;
        }
      }
      rec[l] = (uint32_t)i | (flags << 16);
      wave_sync();
      uint32_t *tp = pk0; pk0 = pk1; pk1 = tp;
      np = total >> 1;
    }
    // Extract_Bit_Lengths (:180-189): leaves among the x items in use at each level (uniform over the wave)
    const float rch = 1.0f / (float)CH;
    int x = 2 * ns - 2;
    int len[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int l = max_bits; l >= 2; l--) {
      const int ln = __builtin_amdgcn_readfirstlane((int)(((float)x + 0.5f) * rch)), t = x - ln * CH;   // x = ln * CH + t exactly (x <= 575)
      int c = ns;
      if (ln < 64) { const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)rec[l], ln); c = (int)(w & 0xFFFFu) + __popc((w >> 16) & ((1u << t) - 1u)); }
#pragma unroll
      for (int q = 0; q < 5; q++) len[q] += c > lane + 64 * q ? 1 : 0;
      x = 2 * (x - c);
    }
    {
      const int c = x < ns ? x : ns;
#pragma unroll
      for (int q = 0; q < 5; q++) len[q] += c > lane + 64 * q ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < 5; q++) { const int r = lane + 64 * q; if (r < ns) bl[ls[r]] = (uint8_t)len[q]; }
    wave_sync();
  }
}

}  // namespace zada
