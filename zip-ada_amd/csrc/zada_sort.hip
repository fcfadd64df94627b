// zada_sort.hip -- a stable key-value radix sort for gfx950, the one sort primitive of the library that works on whole arrays in HBM:
//   * the BT4 producer of LZMA Level_3 groups the positions of a batch by their hash-2 / hash-3 / hash-4 values (zada_bt4.hip);
//   * the BZip2 rotation sort puts the group lists of its late rounds into text order (zada_bz2.hip, "Late rounds: group lists").
// Keys are 32-bit, sorted by the bits [begin_bit, end_bit); values are 4 or 16 bytes and travel with their keys.
//
// Least significant digit first, at most 9 bits a pass, three launches a pass:
//   k_rs_hist     one workgroup per tile of 4 096 keys: the tile's digit histogram (LDS atomics), written digit-major;
//   (scan)        one exclusive scan over [digit][tile] = where every (digit, tile) group starts in the output;
//   k_rs_scatter  the tile again: a key's place = the group's start + its rank among the tile's keys of the same digit, in index order.
// The rank needs no sort and no atomics: a wave holds 64 consecutive keys per row; the lanes of a row that share a digit find each
// other with one ballot per digit bit, the lowest of them books the group in the wave's own counter table (rows are taken in order, so
// a plain read-then-write by one lane is race free), and the four waves' tables are summed in wave order.  Index order inside equal
// digits is what makes every pass -- and the sort -- stable.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zada_internal.h"

namespace zada {
namespace {

constexpr int RS_THREADS = 256, RS_WAVES = RS_THREADS / 64, RS_ITEMS = 16, RS_TILE = RS_THREADS * RS_ITEMS, RS_MAXRADIX = 512;

__global__ void __launch_bounds__(RS_THREADS) k_rs_hist(const uint32_t *__restrict__ keys, uint32_t n, uint32_t shift, uint32_t bits, uint32_t ntiles, uint32_t *__restrict__ counts) {
  __shared__ uint32_t h[RS_MAXRADIX];
  const uint32_t radix = 1u << bits, mask = radix - 1;
  for (uint32_t i = threadIdx.x; i < radix; i += RS_THREADS) h[i] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int k = 0; k < RS_ITEMS; k++) {
    const uint32_t i = base + k * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < radix; i += RS_THREADS) counts[(size_t)i * ntiles + blockIdx.x] = h[i];
}

template <typename V>
__global__ void __launch_bounds__(RS_THREADS) k_rs_scatter(const uint32_t *__restrict__ keys, const V *__restrict__ vals, uint32_t n, uint32_t shift, uint32_t bits, uint32_t ntiles,
                                                           const uint32_t *__restrict__ starts, uint32_t *__restrict__ keys_out, V *__restrict__ vals_out) {
  __shared__ uint32_t wh[RS_WAVES][RS_MAXRADIX];                    // per wave: keys of each digit seen so far (then: before this wave)
  __shared__ uint32_t gbase[RS_MAXRADIX];                           // where the tile's keys of a digit go in the output, minus where they lie in the sorted tile
  __shared__ uint32_t lbase[RS_MAXRADIX];                           // where they lie in the sorted tile
  __shared__ uint32_t wsum[RS_WAVES];
  __shared__ uint16_t perm[RS_TILE];                                // sorted tile: the element's place in the tile as it was read
  const uint32_t radix = 1u << bits, mask = radix - 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < RS_WAVES * RS_MAXRADIX; i += RS_THREADS) (&wh[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tbase = blockIdx.x * RS_TILE, wbase = tbase + w * (RS_ITEMS * 64);
  const uint64_t lt = (1ull << lane) - 1;
  uint32_t key[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys[i] : 0u;
    const uint32_t d = (key[r] >> shift) & mask;
    uint64_t m = __ballot(valid);
    for (uint32_t b = 0; b < bits; b++) {
      const uint64_t bb = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bb : ~bb;
    }
    const uint32_t before = wh[w][d];
    rank[r] = before + (uint32_t)__popcll(m & lt);
    if (valid && (m & lt) == 0) wh[w][d] = before + (uint32_t)__popcll(m);
  }
  __syncthreads();
  // per digit: the waves' counts become "before this wave"; the tile's count per digit, scanned over the digits = the sorted tile's layout
  {
    constexpr int PER = RS_MAXRADIX / RS_THREADS;                    // digits per thread, consecutive
    uint32_t tot[PER], s = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const uint32_t d = threadIdx.x * PER + q;
      uint32_t t = 0;
      if (d < radix) {
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) { const uint32_t c = wh[k][d]; wh[k][d] = t; t += c; }
      }
      tot[q] = t; s += t;
    }
    uint32_t incl = s;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = incl - s;
    for (int k = 0; k < w; k++) run += wsum[k];
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const uint32_t d = threadIdx.x * PER + q;
      if (d < radix) { lbase[d] = run; gbase[d] = starts[(size_t)d * ntiles + blockIdx.x] - run; }
      run += tot[q];
    }
  }
  __syncthreads();
  // the sorted tile in LDS (who goes where), ...
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t i = wbase + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (key[r] >> shift) & mask;
      perm[lbase[d] + wh[w][d] + rank[r]] = (uint16_t)(i - tbase);
    }
  }
  __syncthreads();
  // ... then written out in sorted order: the keys of a digit go to consecutive places, so neighbouring lanes store side by side (as
  // each element went straight from its register to its place, every store was a sector of its own: 14 GB written for 4 GB of pairs)
  const uint32_t cnt = n - tbase < (uint32_t)RS_TILE ? n - tbase : (uint32_t)RS_TILE;
  for (uint32_t j = threadIdx.x; j < cnt; j += RS_THREADS) {
    const uint32_t src = tbase + perm[j];
    const uint32_t k = keys[src];                                    // (the tile was read a moment ago: L1 / L2)
    const uint32_t dst = gbase[(k >> shift) & mask] + j;
    keys_out[dst] = k;
    vals_out[dst] = vals[src];
  }
}

struct V16 { uint32_t a, b, c, d; };

template <typename V> __global__ void k_rs_copy(const uint32_t *__restrict__ k, const V *__restrict__ v, uint32_t n, uint32_t *__restrict__ ko, V *__restrict__ vo) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ko[i] = k[i]; vo[i] = v[i]; }
}

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// Temporary storage of radix_sort_pairs for n pairs with values of value_bytes (4 or 16): the tile histograms and their scan, and one
// more copy of keys and values for the passes to alternate between.
size_t radix_sort_tmp_bytes(size_t n, size_t value_bytes) {
  const size_t ntiles = (n + RS_TILE - 1) / RS_TILE, cells = ntiles * RS_MAXRADIX;
  return up256(cells * 4) + up256(((cells + 1023) / 1024 + 1) * 4) + 256 + up256(n * 4) + up256(n * value_bytes);
}

// Sorts n (key, value) pairs by the key bits [begin_bit, end_bit), stable; the result is in keys_out / vals_out, the inputs are left
// as they are (keys_in == keys_out and vals_in == vals_out is allowed).  Returns 0 or ZADA_E_HIP.
int radix_sort_pairs(Ctx *c, hipStream_t st, void *tmp, size_t tmp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const void *vals_in, void *vals_out, size_t value_bytes,
                     size_t n, unsigned begin_bit, unsigned end_bit) {
  if (n == 0) return 0;
  if (n >= (1ull << 32) || (value_bytes != 4 && value_bytes != 16) || end_bit > 32 || begin_bit > end_bit || tmp_bytes < radix_sort_tmp_bytes(n, value_bytes)) {
    c->err = "radix_sort_pairs: bad arguments"; return ZADA_E_HIP_;
  }
  const uint32_t N = (uint32_t)n, ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
  const size_t cells = (size_t)ntiles * RS_MAXRADIX;
  uint8_t *t = (uint8_t *)tmp;
  uint32_t *counts = (uint32_t *)t; t += up256(cells * 4);
  uint32_t *sums = (uint32_t *)t; t += up256(((cells + 1023) / 1024 + 1) * 4);
  uint32_t *total = (uint32_t *)t; t += 256;
  uint32_t *keys_alt = (uint32_t *)t; t += up256(n * 4);
  void *vals_alt = t;
  const unsigned nbits = end_bit - begin_bit, passes = (nbits + 8) / 9;
  if (passes == 0) {
    if (keys_in != keys_out || vals_in != vals_out) {
      if (value_bytes == 4) hipLaunchKernelGGL(k_rs_copy<uint32_t>, dim3((N + 255) / 256), dim3(256), 0, st, keys_in, (const uint32_t *)vals_in, N, keys_out, (uint32_t *)vals_out);
      else hipLaunchKernelGGL(k_rs_copy<V16>, dim3((N + 255) / 256), dim3(256), 0, st, keys_in, (const V16 *)vals_in, N, keys_out, (V16 *)vals_out);
    }
    return hip_check(c, hipGetLastError(), "radix sort") ? ZADA_E_HIP_ : 0;
  }
  const uint32_t *ksrc = keys_in; const void *vsrc = vals_in;
  unsigned shift = begin_bit, left = nbits;
  for (unsigned p = 0; p < passes; p++) {
    const unsigned bits = (left + (passes - p) - 1) / (passes - p);          // even digits: 22 bits = 8 + 7 + 7, 17 bits = 9 + 8
    const bool to_out = ((passes - 1 - p) & 1) == 0;                          // the last pass lands in keys_out / vals_out
    uint32_t *kdst = to_out ? keys_out : keys_alt; void *vdst = to_out ? vals_out : vals_alt;
    // (a first pass from in to out with in == out would overwrite its own input: an odd pass count starts in `out`; go through alt first)
    if (p == 0 && to_out && (keys_in == keys_out || vals_in == vals_out)) {
      if (value_bytes == 4) hipLaunchKernelGGL(k_rs_copy<uint32_t>, dim3((N + 255) / 256), dim3(256), 0, st, keys_in, (const uint32_t *)vals_in, N, keys_alt, (uint32_t *)vals_alt);
      else hipLaunchKernelGGL(k_rs_copy<V16>, dim3((N + 255) / 256), dim3(256), 0, st, keys_in, (const V16 *)vals_in, N, keys_alt, (V16 *)vals_alt);
      ksrc = keys_alt; vsrc = vals_alt;
    }
    hipLaunchKernelGGL(k_rs_hist, dim3(ntiles), dim3(RS_THREADS), 0, st, ksrc, N, shift, bits, ntiles, counts);
    exclusive_scan_u32(st, counts, counts, sums, total, (uint32_t)((size_t)ntiles << bits));
    if (value_bytes == 4)
      hipLaunchKernelGGL(k_rs_scatter<uint32_t>, dim3(ntiles), dim3(RS_THREADS), 0, st, ksrc, (const uint32_t *)vsrc, N, shift, bits, ntiles, counts, kdst, (uint32_t *)vdst);
    else
      hipLaunchKernelGGL(k_rs_scatter<V16>, dim3(ntiles), dim3(RS_THREADS), 0, st, ksrc, (const V16 *)vsrc, N, shift, bits, ntiles, counts, kdst, (V16 *)vdst);
    ksrc = kdst; vsrc = vdst;
    shift += bits; left -= bits;
  }
  return hip_check(c, hipGetLastError(), "radix sort") ? ZADA_E_HIP_ : 0;
}

}  // namespace zada
