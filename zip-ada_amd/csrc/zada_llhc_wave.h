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
//    The lists are built here level by level with parallel merges (binary-search ranks), each
//    level keeping one bit per item (leaf / package); Extract_Bit_Lengths (:180-189) becomes a
//    prefix popcount per level.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zada {

constexpr int LLHC_WAVE_SCRATCH = 6656;    // bytes of LDS per instance

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
__device__ inline void llhc_wave(const uint32_t *freq, int n, int max_bits, uint8_t *bl, uint8_t *scratch, int lane) {
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
    auto route = [&](int lo, int m) {                 // uniform
      if (m < 2) return;
      if (m <= 16) { if (lane == 0) { small[2 * nsmall] = (uint16_t)lo; small[2 * nsmall + 1] = (uint16_t)m; } nsmall++; }
      else { if (lane == 0) { stk[2 * sp] = (uint16_t)lo; stk[2 * sp + 1] = (uint16_t)m; } sp++; }
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
  {
    uint32_t *cur = (uint32_t *)u;                    // merged list of the current level (<= 575 items)
    uint32_t *pk = cur + 608;                         // packages made of the previous level (<= 287)
    uint32_t *bits = pk + 304;                        // [max_bits + 1][19] leaf flags per level
    uint16_t *acnt = (uint16_t *)(bits + 16 * 19);
    const uint32_t *prev = lw;
    int plen = ns;
    for (int l = 2; l <= max_bits; l++) {
      const int np = plen >> 1;
      for (int i = lane; i < np; i += 64) pk[i] = prev[2 * i] + prev[2 * i + 1];
      for (int i = lane; i < 19; i += 64) bits[l * 19 + i] = 0;
      wave_sync();
      // leaves: position = r + #(packages <= weight)      (a package precedes a leaf of equal weight)
      for (int r = lane; r < ns; r += 64) {
        const uint32_t v = lw[r];
        int lo = 0, cnt = np;
        while (cnt > 0) { const int half = cnt >> 1; if (pk[lo + half] <= v) { lo += half + 1; cnt -= half + 1; } else cnt = half; }
        const int pos = r + lo;
        cur[pos] = v;
        atomicOr(&bits[l * 19 + (pos >> 5)], 1u << (pos & 31));
      }
      // packages: position = i + #(leaves < weight)
      for (int i = lane; i < np; i += 64) {
        const uint32_t v = pk[i];
        int lo = 0, cnt = ns;
        while (cnt > 0) { const int half = cnt >> 1; if (lw[lo + half] < v) { lo += half + 1; cnt -= half + 1; } else cnt = half; }
        cur[i + lo] = v;
      }
      wave_sync();
      prev = cur;
      plen = ns + np;
    }
    // Extract_Bit_Lengths: number of leaves among the items in use at each level
    int x = 2 * ns - 2;
    for (int l = max_bits; l >= 2; l--) {
      uint32_t c = 0;
      if (lane < 19) {
        const int lo_bit = lane * 32;
        uint32_t wv = bits[l * 19 + lane];
        if (x <= lo_bit) wv = 0;
        else if (x < lo_bit + 32) wv &= (1u << (x - lo_bit)) - 1u;
        c = __popc(wv);
      }
      for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
      if (lane == 0) acnt[l] = (uint16_t)c;
      x = 2 * (x - (int)c);
    }
    if (lane == 0) acnt[1] = (uint16_t)(x < ns ? x : ns);
    wave_sync();
    for (int r = lane; r < ns; r += 64) {
      int len = 0;
      for (int l = 1; l <= max_bits; l++) len += acnt[l] > r ? 1 : 0;
      bl[ls[r]] = (uint8_t)len;
    }
    wave_sync();
  }
}

}  // namespace zada
