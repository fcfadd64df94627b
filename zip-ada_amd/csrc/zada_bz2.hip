// zada_bz2.hip -- BZip2 encoder: zip-ada's BZip2.Encoding.Encode (zip_lib/bzip2-encoding.adb:87-1431)
// behind Zip.Compress.BZip2_E (zip_lib/zip-compress-bzip2_e.adb:44-157), on the GPU.
//
// The format makes blocks independent, and the reference adds more independent work per block: four splitting tactics
// (:1214-1345) whose sub-blocks are each a complete Encode_Block (:148-1134).  The unit of work here is therefore the
// SUB-BLOCK (one Encode_Block: a raw range of the input), and a batch of sub-blocks goes through every stage together:
//
//   acquisition (block limits, :1161-1209)  ->  segmentation (data_segmentation.adb:39-105)  ->  per sub-block:
//   RLE_1 + CRC (:167-213)  ->  BWT (:222-300)  ->  MTF + RLE_2 (:320-412)  ->  entropy coders (:418-1010)  ->  bits (:1014-1116)
//   ->  per block: smallest tactic (:1312-1318), bit-shifted concatenation.
//
// Element space: the RLE_1 bytes of the batch's sub-blocks, back to back (sub-block s owns [off[s], off[s] + n[s])).  The
// rotation sort keeps every sub-block's rows inside that same index range, so one segmented radix sort pass serves all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <thread>
#include <vector>
#include "../../include/zada.h"
#include "zada_internal.h"
#include "zada_llhc_wave.h"

namespace zada {

#define BZ_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_check(c, e__, #call); } while (0)

// ---------------------------------------------------------------------------------------------------------------
//  small device helpers
// ---------------------------------------------------------------------------------------------------------------
struct OpSum { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } static constexpr uint32_t identity = 0; };
struct OpMax { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } static constexpr uint32_t identity = 0; };

template <class Op>
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane, Op op) {
  for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off); if (lane >= off) v = op(v, t); }
  return v;
}
// inclusive scan over the threads of a workgroup (blockDim.x a multiple of 64, <= 1024); *total = the reduction
template <class Op>
__device__ __forceinline__ uint32_t wg_scan_incl(uint32_t v, uint32_t *lds17, Op op, uint32_t *total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  const uint32_t incl = wave_scan_incl(v, lane, op);
  __syncthreads();
  if (lane == 63) lds17[w] = incl;
  __syncthreads();
  uint32_t base = Op::identity, tot = Op::identity;
  for (int k = 0; k < nw; k++) { const uint32_t x = lds17[k]; if (k < w) base = op(base, x); tot = op(tot, x); }
  if (total) *total = tot;
  return op(base, incl);
}

// ---------------------------------------------------------------------------------------------------------------
//  generic scans of uint32 arrays (three kernels: tile aggregates, scan of the aggregates, apply)
// ---------------------------------------------------------------------------------------------------------------
constexpr int SC_TILE = 8192;   // 1024 threads x 8

struct FArr { const uint32_t *a; __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return a[i]; } };
struct FArrPad { const uint32_t *a; uint64_t n; __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return i < n ? a[i] : 0u; } };
// a thread's eight consecutive values f(base) .. f(base + 7), the identity beyond n; plain arrays (allocations: 256-byte aligned) as two
// 16-byte loads -- eight 4-byte loads 32 bytes apart from lane to lane ask the memory pipe for eight times as many sectors
template <class Op, class F>
__device__ __forceinline__ void scan_load8(const F &f, uint64_t base, uint64_t n, uint32_t (&x)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) x[k] = base + k < n ? f(base + k) : Op::identity;
}
template <class Op>
__device__ __forceinline__ void scan_load8(const FArr &f, uint64_t base, uint64_t n, uint32_t (&x)[8]) {
  if (base + 8 <= n && ((uintptr_t)(f.a + base) & 15u) == 0) {
    const uint4 a = *(const uint4 *)(f.a + base), b = *(const uint4 *)(f.a + base + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = base + k < n ? f.a[base + k] : Op::identity;
  }
}

template <class Op, class F>
__global__ void __launch_bounds__(1024) k_scan_agg(F f, uint64_t n, uint32_t *__restrict__ agg) {
  __shared__ uint32_t l17[17];
  const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * 8;
  Op op;
  uint32_t v = Op::identity, x[8];
  scan_load8<Op>(f, base, n, x);
#pragma unroll
  for (int k = 0; k < 8; k++) v = op(v, x[k]);
  uint32_t tot;
  wg_scan_incl(v, l17, op, &tot);
  if (threadIdx.x == 0) agg[blockIdx.x] = tot;
}
// exclusive scan of the aggregates in place (one workgroup); *total (may be null) receives the reduction
template <class Op>
__global__ void __launch_bounds__(1024) k_scan_aggs(uint32_t *__restrict__ agg, uint32_t nb, uint32_t *__restrict__ total) {
  __shared__ uint32_t l17[17];
  __shared__ uint32_t carry_s;
  Op op;
  if (threadIdx.x == 0) carry_s = Op::identity;
  __syncthreads();
  for (uint32_t b = 0; b < nb; b += 1024) {
    const uint32_t i = b + threadIdx.x;
    const uint32_t v = i < nb ? agg[i] : Op::identity;
    uint32_t tot;
    const uint32_t incl = wg_scan_incl(v, l17, op, &tot);
    const uint32_t carry = carry_s;
    // exclusive value = carry op (inclusive without own): recompute from the neighbour
    uint32_t excl = __shfl_up(incl, 1);
    if ((threadIdx.x & 63) == 0) {
      excl = Op::identity;
      for (int k = 0; k < (int)(threadIdx.x >> 6); k++) excl = op(excl, l17[k]);
    }
    if (i < nb) agg[i] = op(carry, excl);
    __syncthreads();
    if (threadIdx.x == 0) carry_s = op(carry, tot);
    __syncthreads();
  }
  if (total && threadIdx.x == 0) *total = carry_s;
}
template <class Op, bool INCLUSIVE, class F>
__global__ void __launch_bounds__(1024) k_scan_apply(F f, uint64_t n, const uint32_t *__restrict__ agg, uint32_t *__restrict__ out) {
  __shared__ uint32_t l17[17];
  const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * 8;
  Op op;
  uint32_t x[8], v = Op::identity;
  scan_load8<Op>(f, base, n, x);
#pragma unroll
  for (int k = 0; k < 8; k++) v = op(v, x[k]);
  const uint32_t incl = wg_scan_incl(v, l17, op, nullptr);
  // exclusive prefix of this thread = carry op (everything before this thread in the tile)
  uint32_t before = __shfl_up(incl, 1);
  if ((threadIdx.x & 63) == 0) { before = Op::identity; for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before = op(before, l17[k]); }
  uint32_t run = op(agg[blockIdx.x], before);
  uint32_t y[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (INCLUSIVE) { run = op(run, x[k]); y[k] = run; }
    else { y[k] = run; run = op(run, x[k]); }
  }
  if (base + 8 <= n && ((uintptr_t)(out + base) & 15u) == 0) {
    *(uint4 *)(out + base) = make_uint4(y[0], y[1], y[2], y[3]);
    *(uint4 *)(out + base + 4) = make_uint4(y[4], y[5], y[6], y[7]);
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) if (base + k < n) out[base + k] = y[k];
  }
}
template <class Op, bool INCLUSIVE, class F>
static void scan_launch(hipStream_t st, F f, uint64_t n, uint32_t *agg, uint32_t *out, uint32_t *d_total) {
  if (n == 0) { if (d_total) hipMemsetAsync(d_total, 0, 4, st); return; }
  const uint32_t nb = (uint32_t)((n + SC_TILE - 1) / SC_TILE);
  hipLaunchKernelGGL((k_scan_agg<Op, F>), dim3(nb), dim3(1024), 0, st, f, n, agg);
  hipLaunchKernelGGL((k_scan_aggs<Op>), dim3(1), dim3(1024), 0, st, agg, nb, d_total);
  hipLaunchKernelGGL((k_scan_apply<Op, INCLUSIVE, F>), dim3(nb), dim3(1024), 0, st, f, n, agg, out);
}



// ---------------------------------------------------------------------------------------------------------------
//  sub-block tables (device, structure of arrays)
// ---------------------------------------------------------------------------------------------------------------
struct SubTab {
  const uint64_t *raw_start;   // offset in the input
  const uint32_t *raw_len;
  uint32_t *off;               // element offset (RLE_1 bytes of the sub-blocks, back to back)
  uint32_t *n;                 // RLE_1 size
  uint32_t *inuse;             // [nsb][8]: bytes in use (bit b of word b >> 5)
  uint32_t *crc;               // block CRC (final)
  uint32_t *bwt_index;
  uint32_t nsb;
};

// A tile of a sub-block: `lo` is the offset inside the sub-block (raw space or element space)
struct Tile { uint32_t sb, lo; };
// Workgroups are dealt to the eight XCDs in turn; each XCD has its own L2.  The rotation sort's kernels look up classes and marks
// all over a sub-block, so the tiles of a sub-block should meet in ONE L2: workgroup b of a grid of 8 * ceil(n / 8) takes tile
// (b % 8) * ceil(n / 8) + b / 8, i.e. every XCD walks a contiguous eighth of the tile list.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t ntiles) {
  const uint32_t per = (ntiles + 7) >> 3;
  return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}
static inline uint32_t xcd_grid(uint32_t ntiles) { return ((ntiles + 7) >> 3) << 3; }

// ---------------------------------------------------------------------------------------------------------------
//  RLE_1 (:167-213).  The reference's state machine cuts a run of equal bytes into pieces of at most 259 bytes,
//  restarting at the sub-block's first byte; a piece of r bytes is stored as min(4, r) bytes plus, when r >= 4, the count r - 4.
//  Byte p (offset o in its clipped run) therefore emits itself when o % 259 < 4 and, if it ends its piece, the count.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RT_TILE = 8192;     // raw bytes per tile, 256 threads x 32

// per thread: 32 bytes.  run0 = offset of the thread's first byte in its clipped run (from the workgroup max-scan)
struct RleThread {
  uint32_t emits;
  uint32_t use_lo[8];
};

// Offsets within the sub-block: tile covers [lo, hi).  Returns for the calling thread the number of bytes it emits; if
// `dst` is non-null also writes them at dst[prefix ...] (prefix = exclusive scan of the emit counts, computed inside).
template <bool EMIT>
__device__ __forceinline__ uint32_t rle1_tile(const uint8_t *__restrict__ raw, uint32_t len, uint32_t lo, uint32_t carry_rs1, uint32_t *l17,
                                              uint8_t *__restrict__ dst, uint32_t dst_base, uint8_t *seen /* [256], LDS: byte values that occur */) {
  const int tid = threadIdx.x;
  const uint32_t p0 = lo + (uint32_t)tid * 32;
  uint8_t b[34];   // b[0] = byte before the thread's first, b[1..32] = the thread's bytes, b[33] = byte after
  if (p0 >= 4 && (uint64_t)p0 + 36 <= (uint64_t)len) {
    // (inside the sub-block: ten words instead of 34 byte loads)
    uint32_t w[10];
    __builtin_memcpy(w, raw + p0 - 4, 40);
#pragma unroll
    for (int k = 0; k < 34; k++) b[k] = (uint8_t)(w[(k + 3) >> 2] >> (8 * ((k + 3) & 3)));
  } else {
#pragma unroll
    for (int k = 0; k < 34; k++) { const int64_t p = (int64_t)p0 + k - 1; b[k] = (p >= 0 && p < (int64_t)len) ? raw[p] : 0; }
  }
  // last run start (sub-block offset + 1, 0 = none) among the thread's bytes
  uint32_t last = 0;
#pragma unroll
  for (int k = 0; k < 32; k++) { const uint32_t p = p0 + k; if (p < len && (p == 0 || b[k + 1] != b[k])) last = p + 1; }
  OpMax mx;
  const uint32_t incl = wg_scan_incl(last, l17, mx, nullptr);
  uint32_t before = __shfl_up(incl, 1);
  if ((tid & 63) == 0) { before = 0; for (int k = 0; k < (tid >> 6); k++) before = mx(before, l17[k]); }
  __syncthreads();
  // a thread whose bytes follow no run start inside the tile continues the run the tile begins in: its start comes with the
  // tile (carry_rs1 = last run start in the tiles before + 1, from a scan over the sub-block's tiles)
  uint32_t rs = before ? before - 1 : (carry_rs1 ? carry_rs1 - 1 : 0u);
  uint32_t emits = 0;
  uint8_t outb[40];
#pragma unroll
  for (int k = 0; k < 32; k++) {
    const uint32_t p = p0 + k;
    if (p < len) {
      if (p == 0 || b[k + 1] != b[k]) rs = p;
      const uint32_t r = (p - rs) % 259u;
      const bool piece_end = r == 258u || p + 1 == len || b[k + 2] != b[k + 1];
      if (r < 4) { if (EMIT) outb[emits] = b[k + 1]; emits++; }
      if (piece_end && r + 1 >= 4) { if (EMIT) outb[emits] = (uint8_t)(r + 1 - 4); emits++; }
    }
  }
  if (!EMIT) return emits;
  OpSum sm;
  const uint32_t inc2 = wg_scan_incl(emits, l17, sm, nullptr);
  uint32_t pre = __shfl_up(inc2, 1);
  if ((tid & 63) == 0) { pre = 0; for (int k = 0; k < (tid >> 6); k++) pre += l17[k]; }
  for (uint32_t k = 0; k < emits; k++) {
    dst[dst_base + pre + k] = outb[k];
    seen[outb[k]] = 1;                                 // (plain stores of the same value: as atomics on eight words every byte queued behind its neighbours)
  }
  return emits;
}

// last run start (+ 1; 0 = none) among a tile's bytes
__global__ void __launch_bounds__(256) k_bz_rle_runs(const uint8_t *__restrict__ in, SubTab T, const Tile *__restrict__ tiles, uint32_t *__restrict__ tile_last) {
  __shared__ uint32_t l17[17];
  const Tile t = tiles[blockIdx.x];
  const uint8_t *raw = in + T.raw_start[t.sb];
  const uint32_t len = T.raw_len[t.sb], p0 = t.lo + threadIdx.x * 32;
  uint32_t last = 0;
  uint8_t prev = p0 > 0 && p0 - 1 < len ? raw[p0 - 1] : 0;
  for (uint32_t k = 0; k < 32; k++) {
    const uint32_t p = p0 + k;
    if (p < len) { const uint8_t b = raw[p]; if (p == 0 || b != prev) last = p + 1; prev = b; }
  }
  OpMax mx;
  uint32_t tot;
  wg_scan_incl(last, l17, mx, &tot);
  if (threadIdx.x == 0) tile_last[blockIdx.x] = tot;
}
// exclusive max-scan of per-tile values inside each sub-block
__global__ void __launch_bounds__(64) k_bz_tile_scan_max(const uint32_t *__restrict__ first_tile, uint32_t *__restrict__ tile_val) {
  const uint32_t s = blockIdx.x, t0 = first_tile[s], t1 = first_tile[s + 1];
  const int lane = threadIdx.x;
  uint32_t carry = 0;
  OpMax mx;
  for (uint32_t b = t0; b < t1; b += 64) {
    const uint32_t i = b + lane;
    const uint32_t v = i < t1 ? tile_val[i] : 0;
    const uint32_t incl = wave_scan_incl(v, lane, mx);
    uint32_t excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0;
    if (i < t1) tile_val[i] = mx(carry, excl);
    carry = mx(carry, __shfl(incl, 63));
  }
}
__global__ void __launch_bounds__(256) k_bz_rle_count(const uint8_t *__restrict__ in, SubTab T, const Tile *__restrict__ tiles, const uint32_t *__restrict__ tile_rs,
                                                      uint32_t *__restrict__ tile_cnt) {
  __shared__ uint32_t l17[17];
  const Tile t = tiles[blockIdx.x];
  const uint8_t *raw = in + T.raw_start[t.sb];
  const uint32_t e = rle1_tile<false>(raw, T.raw_len[t.sb], t.lo, tile_rs[blockIdx.x], l17, nullptr, 0, nullptr);
  OpSum sm;
  uint32_t tot;
  wg_scan_incl(e, l17, sm, &tot);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

// exclusive scan of per-tile values inside each sub-block (tiles of a sub-block are consecutive); total -> T.n / other
__global__ void __launch_bounds__(64) k_bz_tile_scan(const uint32_t *__restrict__ first_tile /*[nsb+1]*/, uint32_t *__restrict__ tile_val,
                                                     uint32_t *__restrict__ totals) {
  const uint32_t s = blockIdx.x, t0 = first_tile[s], t1 = first_tile[s + 1];
  const int lane = threadIdx.x;
  uint32_t carry = 0;
  OpSum sm;
  for (uint32_t b = t0; b < t1; b += 64) {
    const uint32_t i = b + lane;
    const uint32_t v = i < t1 ? tile_val[i] : 0;
    const uint32_t incl = wave_scan_incl(v, lane, sm);
    if (i < t1) tile_val[i] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) totals[s] = carry;
}

__global__ void __launch_bounds__(256) k_bz_rle_emit(const uint8_t *__restrict__ in, SubTab T, const Tile *__restrict__ tiles, const uint32_t *__restrict__ tile_rs,
                                                     const uint32_t *__restrict__ tile_off, uint8_t *__restrict__ rle) {
  __shared__ uint32_t l17[17];
  __shared__ uint8_t seen[256];
  const Tile t = tiles[blockIdx.x];
  seen[threadIdx.x] = 0;                               // (256 threads)
  __syncthreads();
  const uint8_t *raw = in + T.raw_start[t.sb];
  rle1_tile<true>(raw, T.raw_len[t.sb], t.lo, tile_rs[blockIdx.x], l17, rle, T.off[t.sb] + tile_off[blockIdx.x], seen);
  __syncthreads();
  {
    const unsigned long long mk = __ballot(seen[threadIdx.x] != 0);               // wave w holds the byte values 64 w .. 64 w + 63
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0 && (uint32_t)mk) atomicOr(&T.inuse[t.sb * 8 + 2 * w], (uint32_t)mk);
    if (lane == 1 && (uint32_t)(mk >> 32)) atomicOr(&T.inuse[t.sb * 8 + 2 * w + 1], (uint32_t)(mk >> 32));
  }
}

// ---------------------------------------------------------------------------------------------------------------
//  BZip2.CRC (bzip2.adb): MSB-first CRC-32, polynomial 0x04C11DB7.  One lane per raw tile, then one lane per sub-block
//  folds the tiles: crc(A ++ B) = crc_reg(A) * x^(8 |B|) + crc_0(B)  in GF(2)[x] / P.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bzcrc_mulx8(uint32_t r) {     // r * x^8 mod P
#pragma unroll
  for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04C11DB7u : r << 1;
  return r;
}
__device__ __forceinline__ uint32_t bzcrc_mul(uint32_t a, uint32_t b) {   // a * b mod P (polynomials, bit 31 = x^31)
  uint32_t r = 0;
  for (int k = 0; k < 32; k++) {
    r = (r & 0x80000000u) ? (r << 1) ^ 0x04C11DB7u : r << 1;      // r *= x
    if (b & (0x80000000u >> k)) r ^= a;                           // + a * (bit of b, from x^31 down)
  }
  return r;
}
__device__ __forceinline__ uint32_t bzcrc_xpow8(uint32_t nbytes) {       // x^(8 nbytes) mod P
  uint32_t result = 1, base = 0x100;                                     // x^8
  while (nbytes) { if (nbytes & 1) result = bzcrc_mul(result, base); base = bzcrc_mul(base, base); nbytes >>= 1; }
  return result;
}
__global__ void __launch_bounds__(64) k_bz_crc_tiles(const uint8_t *__restrict__ in, SubTab T, const Tile *__restrict__ tiles, uint32_t ntiles,
                                                     uint32_t *__restrict__ tile_crc) {
  __shared__ uint32_t tab[256];
  for (int i = threadIdx.x; i < 256; i += 64) tab[i] = bzcrc_mulx8((uint32_t)i << 24);
  __syncthreads();
  const uint32_t ti = blockIdx.x * 64 + threadIdx.x;
  if (ti >= ntiles) return;
  const Tile t = tiles[ti];
  const uint8_t *raw = in + T.raw_start[t.sb] + t.lo;
  const uint32_t len = min((uint32_t)RT_TILE, T.raw_len[t.sb] - t.lo);
  uint32_t r = 0;
  for (uint32_t k = 0; k < len; k++) r = tab[(r >> 24) ^ raw[k]] ^ (r << 8);
  tile_crc[ti] = r;
}
__global__ void k_bz_crc_fold(SubTab T, const uint32_t *__restrict__ first_tile, const uint32_t *__restrict__ tile_crc) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= T.nsb) return;
  const uint32_t t0 = first_tile[s], t1 = first_tile[s + 1], len = T.raw_len[s];
  const uint32_t xt = bzcrc_xpow8(RT_TILE);
  // register after |len| bytes from the all-ones start: 0xFFFFFFFF * x^(8 len) + crc_0(data)
  uint32_t r = 0;
  for (uint32_t t = t0; t < t1; t++) {
    const uint32_t tl = min((uint32_t)RT_TILE, len - (t - t0) * RT_TILE);
    r = bzcrc_mul(r, tl == RT_TILE ? xt : bzcrc_xpow8(tl)) ^ tile_crc[t];
  }
  r ^= bzcrc_mul(0xFFFFFFFFu, bzcrc_xpow8(len));
  T.crc[s] = ~r;
}

// ---------------------------------------------------------------------------------------------------------------
//  BWT (:222-300): rotation sort by prefix doubling.  SA[i] = element (global index) of the i-th smallest rotation of
//  its sub-block, CL[g] = class of element g = the index of the first row of its group of rotations that agree on the
//  first h bytes.  Round: rows shifted back by h are already ordered by their second half, so one stable sort of those by
//  the class of their first half orders them by 2h bytes.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BW_TILE = 8192;    // elements per tile, 1024 threads x 8 (wave w owns 512 consecutive elements)

// stable radix pass over (key, val) pairs, segmented by sub-block: digit = (key >> shift) & 255
// (BITS: width of the digit -- 8 for the first sort's four bytes, 10 for the rounds' classes, which are below 2^20: two passes)
template <int BITS>
__global__ void __launch_bounds__(1024) k_bz_radix_hist(const uint32_t *__restrict__ key, SubTab T, const Tile *__restrict__ tiles,
                                                        const uint32_t *__restrict__ first_tile, const uint8_t *__restrict__ done, int shift,
                                                        uint32_t *__restrict__ H, uint32_t ntiles_x) {
  constexpr uint32_t NB = 1u << BITS;
  __shared__ uint32_t cnt[NB];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t n = T.n[t.sb], base = T.off[t.sb] + t.lo, m = done[t.sb] ? 0u : min((uint32_t)BW_TILE, n - t.lo);
  if (threadIdx.x < NB) cnt[threadIdx.x] = 0;
  __syncthreads();
  {
    uint32_t kx[BW_TILE / 1024];                       // (the eight loads together, then the counts)
#pragma unroll
    for (int q = 0; q < BW_TILE / 1024; q++) { const uint32_t i = (uint32_t)q * 1024u + threadIdx.x; kx[q] = key[base + (i < m ? i : 0u)]; }
#pragma unroll
    for (int q = 0; q < BW_TILE / 1024; q++) if ((uint32_t)q * 1024u + threadIdx.x < m) atomicAdd(&cnt[(kx[q] >> shift) & (NB - 1u)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < NB) {
    const uint32_t t0 = first_tile[t.sb], ts = first_tile[t.sb + 1] - t0;
    H[(uint64_t)t0 * NB + (uint64_t)threadIdx.x * ts + (bx - t0)] = cnt[threadIdx.x];
  }
}
template <int BITS>
__global__ void __launch_bounds__(1024) k_bz_radix_scatter(const uint32_t *__restrict__ key, const uint32_t *__restrict__ val, SubTab T,
                                                           const Tile *__restrict__ tiles, const uint32_t *__restrict__ first_tile,
                                                           const uint8_t *__restrict__ done, int shift,
                                                           const uint32_t *__restrict__ H, uint32_t *__restrict__ key_out, uint32_t *__restrict__ val_out, uint32_t ntiles_x) {
  constexpr uint32_t NB = 1u << BITS;
  __shared__ uint32_t cnt[16 * NB];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  if (done[t.sb]) return;
  const uint32_t n = T.n[t.sb], base = T.off[t.sb] + t.lo, m = min((uint32_t)BW_TILE, n - t.lo);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int i = tid; i < (int)(16 * NB); i += 1024) cnt[i] = 0;
  __syncthreads();
  uint32_t kx[8], vx[8], rk[8];
  uint32_t *mycnt = cnt + w * NB;
  // rank among the wave's elements with the same digit, in element order (LDS atomics of one instruction are served in lane
  // order; see sort_pass in zada_lz.hip and tests/probes/lds_atomic_order.hip)
  // (the sixteen loads first, without conditions -- an element beyond the tile's end reads the tile's first --, then the atomics: under
  // `if (i < m) { load; load; atomic }` every element was a global round trip of its own, eight in a row per lane)
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const uint32_t i = (uint32_t)w * 512 + it * 64 + lane, j = i < m ? i : 0u;
    kx[it] = key[base + j]; vx[it] = val[base + j];
  }
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const uint32_t i = (uint32_t)w * 512 + it * 64 + lane;
    rk[it] = 0;
    if (i < m) rk[it] = atomicAdd(&mycnt[(kx[it] >> shift) & (NB - 1u)], 1u);
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves, plus the tile's base from the scanned histogram
  if ((uint32_t)tid < NB) {
    const uint32_t t0 = first_tile[t.sb], ts = first_tile[t.sb + 1] - t0;
    // the scan runs over all sub-blocks' histograms; a sub-block's own part starts at its first entry
    uint32_t run = T.off[t.sb] + H[(uint64_t)t0 * NB + (uint64_t)tid * ts + (bx - t0)] - H[(uint64_t)t0 * NB];
    for (int k = 0; k < 16; k++) { const uint32_t x = cnt[k * NB + tid]; cnt[k * NB + tid] = run; run += x; }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const uint32_t i = (uint32_t)w * 512 + it * 64 + lane;
    if (i < m) { const uint32_t d = mycnt[(kx[it] >> shift) & (NB - 1u)] + rk[it]; key_out[d] = kx[it]; val_out[d] = vx[it]; }
  }
}

// first keys: four bytes of the rotation starting at the element, big end first
__global__ void k_bz_bwt_init(const uint8_t *__restrict__ rle, SubTab T, const Tile *__restrict__ tiles, uint32_t *__restrict__ key, uint32_t *__restrict__ val, uint32_t ntiles_x) {
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
    const uint32_t l = t.lo + i;
    uint32_t k = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { uint32_t q = l + j; q = q >= n ? q % n : q; k = (k << 8) | rle[off + q]; }
    key[off + l] = k; val[off + l] = off + l;
  }
}

// head values after the first sort: row i starts a group iff its key differs from row i - 1's (or it is the sub-block's first row).
// hv[i] = i + 1 for heads, 0 otherwise: an inclusive max-scan turns it into "my group's first row + 1".
__global__ void __launch_bounds__(1024) k_bz_heads0(const uint32_t *__restrict__ key, SubTab T, const Tile *__restrict__ tiles,
                                                    const uint8_t *__restrict__ done, uint32_t *__restrict__ hv, uint32_t ntiles_x) {
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  if (done[t.sb]) return;
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  uint32_t k0[BW_TILE / 1024], k1[BW_TILE / 1024];   // (loads before stores)
#pragma unroll
  for (int q = 0; q < BW_TILE / 1024; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, l = t.lo + (i < m ? i : 0u), g = off + l;
    k0[q] = key[g]; k1[q] = key[l ? g - 1 : g];
  }
#pragma unroll
  for (int q = 0; q < BW_TILE / 1024; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, l = t.lo + i, g = off + l;
    if (i < m) hv[g] = (l == 0 || k0[q] != k1[q]) ? g + 1 : 0u;
  }
}
// classes from the scanned head values of the first sort; the elements of groups of more than one row are marked (acte)
__global__ void __launch_bounds__(1024) k_bz_set_class(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ hv, const uint32_t *__restrict__ hr,
                                                       SubTab T, const Tile *__restrict__ tiles, uint32_t *__restrict__ cl, uint8_t *__restrict__ acte, uint32_t ntiles_x) {
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  // (a tile is eight rows per thread: all their loads first, then the scattered stores and atomics -- stores count in the same counter as
  // loads, so a loop of load / store / load waits for every scattered store before it goes on)
  constexpr int PER = BW_TILE / 1024;
  uint32_t e8[PER], c8[PER];
  bool act8[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, j = i < m ? i : 0u, l = t.lo + j, g = off + l;
    e8[q] = sa[g]; c8[q] = hr[g] - 1;
    const uint32_t h0 = hv[g], h1 = l + 1 == n ? 1u : hv[g + 1];
    act8[q] = i < m && !(h0 != 0 && h1 != 0);
  }
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x;
    // (the marks are bytes: every element gets its own, a plain store -- as bits they were 535 M scattered atomics and a memset)
    if (i < m) { cl[e8[q]] = c8[q]; acte[e8[q]] = act8[q] ? 1 : 0; }
  }
}
// A doubling round only moves the rows of groups that still have more than one row.  The rows, read in order and shifted
// back by h, are in the order of their second halves; those whose shifted element is still unsorted are filtered out (in
// that order), sorted by the first row of that element's group (stable: three passes over the filtered rows only), and
// written back to the group's rows.
// pass 1: per tile of rows, how many survive the filter; pass 2 (after a scan of the tile counts) writes them in order.  Both
// read the rows and the mark of the shifted element; nothing per row is stored in between.
__device__ __forceinline__ uint32_t bz_shifted_active(const uint32_t *__restrict__ sa, const uint8_t *__restrict__ acte, uint32_t g, uint32_t off, uint32_t n,
                                                       uint32_t h, uint32_t *e_out) {
  uint32_t l = sa[g] - off;
  l = l >= h ? l - h : l + n - h;               // h < n for a sub-block that is not done
  const uint32_t e = off + l;
  *e_out = e;
  return acte[e];
}
// the same for a thread's eight consecutive rows r0 .. r0+7 of the tile (those below m): the rows first, then the marks -- two round trips,
// not sixteen.  Bit k of the result: row r0+k exists and its shifted element is marked; ev[k] = that element.
__device__ __forceinline__ uint32_t bz_shifted_active8(const uint32_t *__restrict__ sa, const uint8_t *__restrict__ acte, uint32_t g0, uint32_t r0, uint32_t m,
                                                        uint32_t off, uint32_t n, uint32_t h, uint32_t (&ev)[8]) {
  uint32_t r[8], a[8];
#pragma unroll
  for (int k = 0; k < 8; k++) r[k] = sa[g0 + (r0 + k < m ? r0 + k : 0u)];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint32_t l = r[k] - off;
    l = l >= h ? l - h : l + n - h;             // h < n for a sub-block that is not done
    ev[k] = off + l;
    a[k] = acte[ev[k]];
  }
  uint32_t fl = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) if (r0 + k < m && a[k]) fl |= 1u << k;
  return fl;
}
__global__ void __launch_bounds__(1024) k_bz_filter_count(const uint32_t *__restrict__ sa, const uint8_t *__restrict__ acte, SubTab T, const Tile *__restrict__ tiles,
                                                          const uint8_t *__restrict__ done, uint32_t h, uint32_t *__restrict__ tile_cnt, uint32_t ntiles_x) {
  __shared__ uint32_t l17[17];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  uint32_t c = 0;
  if (!done[t.sb]) {
    uint32_t ev[8];
    c = (uint32_t)__popc(bz_shifted_active8(sa, acte, off + t.lo, threadIdx.x * 8, m, off, n, h, ev));
  }
  OpSum sm;
  uint32_t tot;
  wg_scan_incl(c, l17, sm, &tot);
  if (threadIdx.x == 0) tile_cnt[bx] = tot;
}
// tscan = exclusive scan of tile_cnt over all tiles (entry ntiles = the total)
__global__ void k_bz_counts(SubTab T, const uint32_t *__restrict__ first_tile, const uint32_t *__restrict__ tscan, uint32_t *__restrict__ coff, uint32_t *__restrict__ cm) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= T.nsb) return;
  const uint32_t a = tscan[first_tile[s]], b = tscan[first_tile[s + 1]];
  coff[s] = a; cm[s] = b - a;
  if (s + 1 == T.nsb) coff[s + 1] = b;
}
__global__ void __launch_bounds__(1024) k_bz_filter_emit(const uint32_t *__restrict__ sa, const uint8_t *__restrict__ acte, const uint32_t *__restrict__ cl, SubTab T,
                                                         const Tile *__restrict__ tiles, const uint8_t *__restrict__ done, uint32_t h,
                                                         const uint32_t *__restrict__ tscan, uint32_t *__restrict__ ckey, uint32_t *__restrict__ cval, uint32_t ntiles_x) {
  __shared__ uint32_t l17[17];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  if (done[t.sb]) return;
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  const uint32_t r0 = threadIdx.x * 8;
  uint32_t ev[8], ck[8];
  const uint32_t fl = bz_shifted_active8(sa, acte, off + t.lo, r0, m, off, n, h, ev), c = (uint32_t)__popc(fl);
#pragma unroll
  for (int k = 0; k < 8; k++) ck[k] = cl[((fl >> k) & 1u) ? ev[k] : off];          // (the survivors' classes, all in flight)
  OpSum sm;
  const uint32_t incl = wg_scan_incl(c, l17, sm, nullptr);
  uint32_t j = tscan[bx] + incl - c;
#pragma unroll
  for (int k = 0; k < 8; k++) if ((fl >> k) & 1u) { ckey[j] = ck[k] - off; cval[j] = ev[k]; j++; }
}
// C = the filtered rows' space (sub-block s owns [coff[s], coff[s] + cm[s])), its tiles in `tiles`.  After the sort the rows of a
// group are together (equal keys).  Three kernels, a thread taking eight consecutive slots, scans inside the tile and a carry
// from a scan over the tiles' aggregates:  (1) per tile, the last slot that starts a run of equal keys;  (2) every slot's run
// start -> its row; the element goes there; slots whose second half differs from the slot before start a new group (hd);
// (3) every slot's group start -> its element's class; elements alone in their group stop being marked.
__global__ void __launch_bounds__(1024) k_bz_rf_agg(const uint32_t *__restrict__ ckey, SubTab C, const Tile *__restrict__ tiles, uint32_t *__restrict__ agg_out, uint32_t ntiles_x) {
  __shared__ uint32_t l17[17];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t cn = C.n[t.sb], coff = C.off[t.sb], m = min((uint32_t)BW_TILE, cn - t.lo);
  const uint32_t r0 = threadIdx.x * 8;
  uint32_t v = 0;
  for (uint32_t k = 0; k < 8; k++) if (r0 + k < m) { const uint32_t l = t.lo + r0 + k, j = coff + l; if (l == 0 || ckey[j] != ckey[j - 1]) v = j + 1; }
  OpMax mx;
  uint32_t tot;
  wg_scan_incl(v, l17, mx, &tot);
  if (threadIdx.x == 0) agg_out[bx] = tot;
}
__global__ void __launch_bounds__(1024) k_bz_place(const uint32_t *__restrict__ ckey, const uint32_t *__restrict__ cval, const uint32_t *__restrict__ carry_rf, SubTab T, SubTab C,
                                                   const Tile *__restrict__ tiles, const uint32_t *__restrict__ cl, uint32_t h, uint32_t *__restrict__ sa,
                                                   uint32_t *__restrict__ hd, uint32_t *__restrict__ agg_out, uint32_t ntiles_x) {
  __shared__ uint32_t l17[17];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t cn = C.n[t.sb], coff = C.off[t.sb], m = min((uint32_t)BW_TILE, cn - t.lo), n = T.n[t.sb], off = T.off[t.sb];
  const uint32_t r0 = threadIdx.x * 8;
  uint32_t key[8], val[8], flags = 0, last = 0;
  auto second = [&](uint32_t e) -> uint32_t { uint32_t l = e - off + h; if (l >= n) l -= n; return cl[off + l]; };
  uint32_t sec_prev = 0;
  // (loads first: the eight slots' keys and elements and the slot in front, then the classes of their second halves -- the loop below
  // stores to scattered rows, and a load behind a store waits for it)
  uint32_t pk0 = 0, pv0 = 0, sec[8];
  {
    const uint32_t jm = coff + t.lo + (r0 < m ? r0 : 0u);
    if (t.lo + r0 > 0 && r0 < m) { pk0 = ckey[jm - 1]; pv0 = cval[jm - 1]; }
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint32_t j = coff + t.lo + (r0 + k < m ? r0 + k : 0u); key[k] = ckey[j]; val[k] = cval[j]; }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) sec[k] = second(r0 + k < m ? val[k] : off);
  if (r0 < m && t.lo + r0 > 0) sec_prev = second(pv0);                              // the slot in front of this thread's first
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (r0 + k < m) {
      const uint32_t l = t.lo + r0 + k, j = coff + l;
      const uint32_t pk = k ? key[k - 1] : pk0;
      if (l == 0 || key[k] != pk) { flags |= 1u << k; last = j + 1; }
    } else { key[k] = 0; val[k] = 0; }
  }
  OpMax mx;
  const uint32_t incl = wg_scan_incl(last, l17, mx, nullptr);
  uint32_t before = __shfl_up(incl, 1);
  if ((threadIdx.x & 63) == 0) { before = 0; for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before = mx(before, l17[k]); }
  uint32_t run = mx(carry_rf[bx], before), hmax = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (r0 + k < m) {
      const uint32_t j = coff + t.lo + r0 + k, e = val[k];
      if ((flags >> k) & 1u) run = j + 1;
      const uint32_t row = off + key[k] + (j - (run - 1));
      sa[row] = e;
      const uint32_t sc = sec[k];
      const uint32_t hv = (((flags >> k) & 1u) || sc != sec_prev) ? row + 1 : 0u;
      hd[j] = hv;
      hmax = mx(hmax, hv);
      sec_prev = sc;
    }
  }
  __syncthreads();
  uint32_t tot;
  wg_scan_incl(hmax, l17, mx, &tot);
  if (threadIdx.x == 0) agg_out[bx] = tot;
}
__global__ void __launch_bounds__(1024) k_bz_newclass(const uint32_t *__restrict__ cval, const uint32_t *__restrict__ hd, const uint32_t *__restrict__ carry_hd, SubTab C,
                                                      const Tile *__restrict__ tiles, uint32_t *__restrict__ cl, uint8_t *__restrict__ acte, uint32_t ntiles_x) {
  __shared__ uint32_t l17[17];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t cn = C.n[t.sb], coff = C.off[t.sb], m = min((uint32_t)BW_TILE, cn - t.lo);
  const uint32_t r0 = threadIdx.x * 8;
  uint32_t hv[9], last = 0;
  for (uint32_t k = 0; k < 9; k++) {
    const uint32_t l = t.lo + r0 + k;
    hv[k] = l < cn ? hd[coff + l] : 1u;                                        // beyond the sub-block: as if a group started there
    if (k < 8 && r0 + k < m && hv[k]) last = hv[k];
  }
  OpMax mx;
  const uint32_t incl = wg_scan_incl(last, l17, mx, nullptr);
  uint32_t before = __shfl_up(incl, 1);
  if ((threadIdx.x & 63) == 0) { before = 0; for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before = mx(before, l17[k]); }
  uint32_t run = mx(carry_hd[bx], before);
  uint32_t ee[8];
#pragma unroll
  for (int k = 0; k < 8; k++) ee[k] = cval[coff + t.lo + (r0 + k < m ? r0 + k : 0u)];   // (loads before the scattered stores)
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (r0 + k < m) {
      const uint32_t e = ee[k];
      if (hv[k]) run = hv[k];
      cl[e] = run - 1;
      if (hv[k] != 0 && hv[k + 1] != 0) acte[e] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
//  Late rounds: group lists.  The rounds above sweep every row of a sub-block (the rows read in order are what gives the second
//  halves' order for free), whatever share of them is still unsorted -- and on data with long repeats most of what is left after a
//  few doublings are groups of a handful of rows that take ten more doublings to come apart.  Once every unsorted group of a
//  sub-block has at most GL_MAX rows the sub-block leaves the sweeps: its groups go on a list (first row, rows, sub-block) and a
//  round sorts each group by the class of its rows' second halves -- a thread per group of up to 8 rows, a wave per larger one --
//  touching nothing but the groups' own rows.  The order among rows that still agree is immaterial (they stay one group; rotations
//  that are equal to the end have equal last bytes, and the original's row is its group's first: :266-276).
//  A round reads the classes of the round before while it makes the new ones: the listed sub-blocks' classes live in TWO arrays that
//  take turns (a round reads one and writes the other, every row of every listed group); a row that comes to stand alone stays on the
//  list for one more round as a group of one, which writes its (final) class into the other array as well.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t GL_SMALL = 8, GL_WAVE = 64, GL_MAX = 8192;
constexpr int GL_NCL = 4;                              // lists by group size
struct GlEntry { uint32_t first, sb_rows; };           // first row | sub-block << 13 | rows - 1
constexpr int GL_SB_BITS = 19;                         // sub-block numbers a list entry can hold (32 - 13 bits): bz_transform leaves the lists off beyond
static_assert(GL_SB_BITS + 13 == 32, "GlEntry::sb_rows: 13 bits of rows - 1, the rest is the sub-block");
__device__ __forceinline__ GlEntry gl_entry(uint32_t first, uint32_t rows, uint32_t sb) { return GlEntry{first, (sb << 13) | (rows - 1u)}; }
__device__ __forceinline__ void gl_unpack(const GlEntry E, uint32_t &first, uint32_t &rows, uint32_t &sb) { first = E.first; rows = (E.sb_rows & 8191u) + 1u; sb = E.sb_rows >> 13; }
// four lists by group size: up to 8 rows (a thread sorts the group), 9 .. 16 (sixteen lanes), 17 .. 64 (a wave), 65 .. 8 192 (a workgroup, in LDS)
constexpr uint32_t GL_MID = 16;
// the thread-per-group list (groups of up to 8 rows) has records of its own: the group's first two rows travel with the entry -- most
// groups are pairs, and in text order (k_bz_gl_keys) fetching a pair's rows from the sorted order would be the one gather left
// Bit 31 of v1 (element indices stay below 2^30): BOTH class arrays hold the group's class -- it was listed when the arrays were equal
// (k_bz_gl_build) or has been through a round without coming apart; such a group has nothing to write while it stays whole.
struct GlSmall { uint32_t first, sb_rows, v0, v1; };
constexpr uint32_t GLS_BOTH = 0x80000000u;
struct GlLists { GlEntry *l[GL_NCL]; GlSmall *s; uint32_t *cnt; uint32_t cap[GL_NCL]; };   // (l[0] is not used: list 0 is s)
static_assert(sizeof(GlSmall) == 16, "the list sort moves 16-byte values (radix_sort_pairs, zada_sort.hip)");   // cnt[0 .. 3]: entries of the lists; cnt[4]: overflow flag
__device__ __forceinline__ int gl_class(uint32_t rows) { return rows <= GL_SMALL ? 0 : rows <= GL_MID ? 1 : rows <= GL_WAVE ? 2 : 3; }

// Text order.  k_bz_gl_build lists the groups in the order of the sorted rotations, where the rows of neighbouring groups lie side by side
// and everything else a round touches -- the classes of the rows' second halves, the classes it writes -- is scattered: four or five 64-byte
// sectors for the 16 bytes a pair needs.  Data with long repeats (what is left for the lists: the same two, three stretches of text, row
// by row) has the opposite order to offer: the groups {a+j, b+j}, j = 0, 1, 2 ... read classes at a+j+h and b+j+h and write them at a+j and
// b+j.  Listed by the position of their first row, neighbouring threads share those sectors and only the group's rows themselves are a
// gather.  The key is that position; one stable radix sort of the (key, entry) pairs when a list has been built (radix_sort_pairs,
// zada_sort.hip); the rounds keep the order workgroup by workgroup.
__global__ void k_bz_gl_keys(const GlSmall *__restrict__ list, uint32_t n, uint32_t *__restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = list[i].v0;
}
// largest unsorted group of every sub-block that is still swept: a row is its group's last when the next row's class differs
__global__ void __launch_bounds__(1024) k_bz_gl_max(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ cl, SubTab T, const Tile *__restrict__ tiles,
                                                    const uint8_t *__restrict__ done, uint32_t *__restrict__ submax, uint32_t ntiles_x) {
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  if (done[t.sb]) return;
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  uint32_t mx = 0;
  // (eight rows per thread: their rows first, then their classes -- two round trips instead of sixteen)
  constexpr int PER = BW_TILE / 1024;
  uint32_t r0[PER], r1[PER], c0[PER], c1[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, j = i < m ? i : 0u, l = t.lo + j, g = off + l;
    r0[q] = sa[g]; r1[q] = sa[l + 1 == n ? g : g + 1];
  }
#pragma unroll
  for (int q = 0; q < PER; q++) { c0[q] = cl[r0[q]]; c1[q] = cl[r1[q]]; }
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, l = t.lo + i, g = off + l;
    if (i < m && (l + 1 == n || c1[q] != c0[q])) mx = max(mx, g - c0[q] + 1);
  }
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 1) atomicMax(&submax[t.sb], mx);
}
__global__ void k_bz_gl_decide(SubTab T, const uint8_t *__restrict__ done, const uint32_t *__restrict__ submax, uint8_t *__restrict__ lmode, uint32_t gl_max) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < T.nsb) lmode[s] = (!done[s] && submax[s] <= gl_max) ? 1 : 0;
}
// Room for this thread's entries (cnt[k] of list k): one atomic per WORKGROUP and list (a counter that every wave hits by itself is
// one address for millions of atomics a round: they queue up at its L2 channel).  Every thread of the workgroup must call it;
// lds: 4 * waves + 4 words.  Returns the thread's first slot of each list (past the capacity: the overflow flag is set).
__device__ __forceinline__ void gl_reserve(GlLists L, const uint32_t (&cnt)[GL_NCL], uint32_t *lds, uint32_t (&slot)[GL_NCL]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t sc[GL_NCL];                                 // inclusive scans over the wave
#pragma unroll
  for (int k = 0; k < GL_NCL; k++) sc[k] = cnt[k];
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < GL_NCL; k++) { const uint32_t a = __shfl_up(sc[k], o); if (lane >= o) sc[k] += a; }
  }
  __syncthreads();                                     // (the call before has read its bases)
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < GL_NCL; k++) lds[GL_NCL * w + k] = sc[k];
  }
  __syncthreads();
  if (threadIdx.x < GL_NCL) {
    const int k = threadIdx.x;
    uint32_t t = 0;
    for (int q = 0; q < nw; q++) { const uint32_t a = lds[GL_NCL * q + k]; lds[GL_NCL * q + k] = t; t += a; }
    lds[GL_NCL * nw + k] = t ? atomicAdd(&L.cnt[k], t) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < GL_NCL; k++) {
    slot[k] = lds[GL_NCL * nw + k] + lds[GL_NCL * w + k] + sc[k] - cnt[k];
    if (cnt[k] && slot[k] + cnt[k] > L.cap[k]) L.cnt[GL_NCL] = 1;
  }
}
// the unsorted groups of the sub-blocks that leave the sweeps, listed by their last rows; the second class array gets the classes
__global__ void __launch_bounds__(1024) k_bz_gl_build(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ cl, uint32_t *__restrict__ cl2, SubTab T,
                                                      const Tile *__restrict__ tiles, const uint8_t *__restrict__ lmode, GlLists L, uint32_t ntiles_x) {
  __shared__ uint32_t lds[GL_NCL * 17];
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  if (!lmode[t.sb]) return;
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  // (the tile's groups are found first, eight rows per thread, and listed after ONE reservation: gl_reserve)
  constexpr int PER = BW_TILE / 1024;
  uint32_t firstA[PER], rowsA[PER];
  uint32_t cnt[GL_NCL] = {0u, 0u, 0u, 0u};
  {
    uint32_t r0[PER], r1[PER], c0[PER], c1[PER], cg[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, j = i < m ? i : 0u, l = t.lo + j, g = off + l;
      r0[q] = sa[g]; r1[q] = sa[l + 1 == n ? g : g + 1]; cg[q] = cl[g];
    }
#pragma unroll
    for (int q = 0; q < PER; q++) { c0[q] = cl[r0[q]]; c1[q] = cl[r1[q]]; }
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, l = t.lo + i, g = off + l;
      uint32_t first = 0, rows = 0;
      if (i < m) {
        cl2[g] = cg[q];                                // (the sub-block's elements are the same index range as its rows: a straight copy)
        if (l + 1 == n || c1[q] != c0[q]) { first = c0[q]; rows = g - c0[q] + 1; }
      }
      firstA[q] = first; rowsA[q] = rows;
      if (rows > 1) cnt[gl_class(rows)]++;
    }
  }
  uint32_t slot[GL_NCL];
  gl_reserve(L, cnt, lds, slot);
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t first = firstA[q], rows = rowsA[q];
    if (rows > 1) {
      const int kc = gl_class(rows);
      if (slot[kc] < L.cap[kc]) {
        const GlEntry e = gl_entry(first, rows, t.sb);
        if (kc == 0) L.s[slot[0]] = GlSmall{e.first, e.sb_rows, sa[first], sa[first + 1] | GLS_BOTH};   // (cl2 := cl above: the arrays agree)
        else L.l[kc][slot[kc]] = e;
      }
      slot[kc]++;
    }
  }
}
__global__ void k_bz_gl_leave(SubTab T, const uint8_t *__restrict__ lmode, uint8_t *__restrict__ done) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < T.nsb && lmode[s]) done[s] = 1;
}
// One round of the small groups (1 .. 8 rows): a THREAD per group -- most groups of the late rounds are pairs, a team of lanes would
// idle -- with the group's rows and keys in registers and a sorting network over them.  clr: the classes of the round before; clw: the
// other array, which gets the class of every row of the group.
__device__ __forceinline__ bool gl_cex(uint32_t &ka, uint32_t &va, uint32_t &kb, uint32_t &vb) {
  const bool sw = kb < ka || (kb == ka && vb < va);
  const uint32_t k0 = sw ? kb : ka, v0 = sw ? vb : va, k1 = sw ? ka : kb, v1 = sw ? va : vb;
  ka = k0; va = v0; kb = k1; vb = v1;
  return sw;
}
#ifndef ZADA_GLS_THREADS
#define ZADA_GLS_THREADS 1024
#endif
#ifndef ZADA_GLS_PER
#define ZADA_GLS_PER 2
#endif
// One reservation on the lists' counters per workgroup and round: the atomics of all workgroups go to one address and are served one
// after the other (≈ 13 ns each: with 256 entries per workgroup the rounds' 430 000 reservations WERE the round, 5.5 ms).  So few, large
// workgroups (1 024 threads), every thread with GLS_PER consecutive entries whose results wait in registers for the one reservation.
constexpr int GLS_THREADS = ZADA_GLS_THREADS, GLS_PER = ZADA_GLS_PER;
__global__ void __launch_bounds__(GLS_THREADS) k_bz_gl_sort_small(const GlSmall *__restrict__ list, const uint32_t *__restrict__ cnt_p, uint32_t h, uint32_t *__restrict__ sa,
                                                          const uint32_t *__restrict__ clr, uint32_t *__restrict__ clw, SubTab T, GlLists next) {
  __shared__ uint32_t lds[GL_NCL * (GLS_THREADS / 64 + 1)];
  const uint32_t count = *cnt_p, g0 = (blockIdx.x * (uint32_t)GLS_THREADS + threadIdx.x) * (uint32_t)GLS_PER;
  uint32_t firstA[GLS_PER], packA[GLS_PER], endsA[GLS_PER], vA[GLS_PER][9];
  bool wholeA[GLS_PER];
  GlSmall EA[GLS_PER];
#pragma unroll
  for (int q = 0; q < GLS_PER; q++) { EA[q] = GlSmall{0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu}; if (g0 + q < count) EA[q] = list[g0 + q]; }
  uint32_t total = 0;
#pragma unroll
  for (int q = 0; q < GLS_PER; q++) {
    const GlSmall E = EA[q];
    uint32_t first = 0, rows = 0, sb = 0;
    if (g0 + q < count) gl_unpack(GlEntry{E.first, E.sb_rows}, first, rows, sb);
    uint32_t k[8], v[9];
#pragma unroll
    for (int j = 0; j < 8; j++) { k[j] = 0xFFFFFFFFu; v[j] = 0xFFFFFFFFu; }
    v[8] = 0xFFFFFFFFu;
    uint32_t ends = 0;                                 // groups for the next round: bit j set = a group ends behind row j
    bool whole = false;
    if (rows == 1) clw[E.v0] = first;                  // a row that came to stand alone in the round before: its class, in the other array as well
    else if (rows > 1) {
      const uint32_t n = T.n[sb], off = T.off[sb];
      const bool both = (E.v1 & GLS_BOTH) != 0;
      v[0] = E.v0; v[1] = E.v1 & ~GLS_BOTH;
#pragma unroll
      for (int j = 2; j < 8; j++) if ((uint32_t)j < rows) v[j] = sa[first + j];
      if (h >= n) {                                    // the rotations of the group are equal: nothing is left to tell them apart, the group goes off the lists
#pragma unroll
        for (int j = 0; j < 8; j++) if ((uint32_t)j < rows) clw[v[j]] = first;
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) if ((uint32_t)j < rows) { uint32_t l = v[j] - off + h; if (l >= n) l -= n; k[j] = clr[off + l]; }
        bool moved = false;                            // (rows that stay where they are are not written back: in text order that write is a gather)
        if (rows == 2) moved = gl_cex(k[0], v[0], k[1], v[1]);
        else {
          // 19 compare-exchanges sort eight (the empty places hold the largest key and stay behind)
#define CX(a, b) moved |= gl_cex(k[a], v[a], k[b], v[b])
          CX(0, 1); CX(2, 3); CX(4, 5); CX(6, 7); CX(0, 2); CX(1, 3); CX(4, 6); CX(5, 7); CX(1, 2); CX(5, 6); CX(0, 4); CX(3, 7); CX(1, 5); CX(2, 6); CX(1, 4); CX(3, 6); CX(2, 4); CX(3, 5); CX(3, 4);
#undef CX
        }
        // rows in order; a row starts a new group where its key differs from the row before
#pragma unroll
        for (int j = 1; j < 8; j++) if ((uint32_t)j < rows && k[j] != k[j - 1]) ends |= 1u << (j - 1);
        whole = ends == 0;                             // the group has not come apart: its class stays `first`
        uint32_t start = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if ((uint32_t)j < rows) {
            if (j > 0 && ((ends >> (j - 1)) & 1u)) start = (uint32_t)j;
            if (moved) sa[first + j] = v[j];
            if (!(whole && both)) clw[v[j]] = first + start;
          }
        }
        ends |= 1u << (rows - 1);
      }
    }
    firstA[q] = first; packA[q] = E.sb_rows; endsA[q] = ends; wholeA[q] = whole;
#pragma unroll
    for (int j = 0; j < 9; j++) vA[q][j] = v[j];
    total += (uint32_t)__popc(ends);
  }
  const uint32_t cnt[GL_NCL] = {total, 0u, 0u, 0u};
  uint32_t slot[GL_NCL];
  gl_reserve(next, cnt, lds, slot);
  uint32_t is = slot[0];
#pragma unroll
  for (int q = 0; q < GLS_PER; q++) {
    // a group starts at row 0 and behind every group's end (static row numbers: the rows stay in registers)
    const uint32_t ends = endsA[q], rows = (packA[q] & 8191u) + 1u, sb = packA[q] >> 13;
    const uint32_t starts = ends ? ((ends << 1) | 1u) & ((1u << rows) - 1u) : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if ((starts >> j) & 1u) {
        const uint32_t e = (uint32_t)__builtin_ctz(ends >> j) + (uint32_t)j;
        const GlEntry ge = gl_entry(firstA[q] + (uint32_t)j, e + 1u - (uint32_t)j, sb);
        // (a group that is still whole has its class in both arrays after this round, whether it was written just now or before)
        if (is < next.cap[0]) next.s[is] = GlSmall{ge.first, ge.sb_rows, vA[q][j], (vA[q][j + 1] & ~GLS_BOTH) | (wholeA[q] ? GLS_BOTH : 0u)};
        is++;
      }
    }
  }
}
// one round of the larger groups: TW lanes per group (16 for 9 .. 16 rows, 64 for 17 .. 64), a bitonic network over cross-lane reads
template <int TW>
__global__ void __launch_bounds__(GLS_THREADS) k_bz_gl_sort_team(const GlEntry *__restrict__ list, const uint32_t *__restrict__ cnt_p, uint32_t h, uint32_t *__restrict__ sa,
                                                         const uint32_t *__restrict__ clr, uint32_t *__restrict__ clw, SubTab T, GlLists next) {
  __shared__ uint32_t lds[GL_NCL * (GLS_THREADS / 64 + 1)];
  const uint32_t count = *cnt_p;
  const uint32_t team = (blockIdx.x * (uint32_t)GLS_THREADS + threadIdx.x) / TW;
  const int lane = threadIdx.x & 63, tl = lane & (TW - 1), tbase = lane - tl;
  uint32_t first = 0, rows = 0, sb = 0;
  if (team < count) gl_unpack(list[team], first, rows, sb);
  uint32_t n = 1, off = 0;
  if (rows) { n = T.n[sb]; off = T.off[sb]; }
  const bool live = rows != 0 && h < n;
  const bool mine = (uint32_t)tl < rows;
  uint32_t k = 0xFFFFFFFFu, v = 0xFFFFFFFFu;
  if (mine) {
    v = sa[first + tl];
    if (live) { uint32_t l = v - off + h; if (l >= n) l -= n; k = clr[off + l]; }
    else clw[v] = first;                               // (equal rotations: the group goes off the lists, its class in both arrays)
  }
#pragma unroll
  for (int size = 2; size <= TW; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const uint32_t ok = __shfl_xor(k, stride), ov = __shfl_xor(v, stride);
      const bool keep_min = ((tl & stride) == 0) == ((tl & size) == 0);
      const bool other_less = ok < k || (ok == k && ov < v);
      const bool other_more = ok > k || (ok == k && ov > v);
      if (keep_min ? other_less : other_more) { k = ok; v = ov; }
    }
  }
  // lanes 0 .. rows-1 of the team hold the group's rows in order; a row starts a new group where its key differs from the row before
  const uint32_t kp = __shfl_up(k, 1);
  const bool head = live && mine && (tl == 0 || kp != k);
  unsigned long long hm = __ballot(head);
  if (TW < 64) hm = (hm >> tbase) & ((1ull << (TW & 63)) - 1ull);      // the team's own heads, bit j = lane j of the team
  uint32_t rows_new = 0;
  if (live && mine) {
    const unsigned long long upto = hm & (tl == 63 ? ~0ull : ((2ull << tl) - 1ull));
    const uint32_t start = 63u - (uint32_t)__builtin_clzll(upto);
    const unsigned long long above = tl == 63 ? 0ull : hm & ~((2ull << tl) - 1ull);
    const uint32_t nxt = above ? (uint32_t)__builtin_ctzll(above) : rows;
    sa[first + tl] = v;
    clw[v] = first + start;
    if (head) rows_new = nxt - (uint32_t)tl;
  }
  const int kc = gl_class(rows_new);
  const uint32_t cnt[GL_NCL] = {rows_new >= 1 && kc == 0 ? 1u : 0u, kc == 1 ? 1u : 0u, kc == 2 ? 1u : 0u, 0u};
  uint32_t slot[GL_NCL];
  gl_reserve(next, cnt, lds, slot);
  const uint32_t v1 = __shfl_down(v, 1);               // (the row behind: the second of a group that this lane's row starts)
  if (rows_new >= 1 && slot[kc] < next.cap[kc]) {
    const GlEntry e = gl_entry(first + (uint32_t)tl, rows_new, sb);
    if (kc == 0) next.s[slot[0]] = GlSmall{e.first, e.sb_rows, v, v1};
    else next.l[kc][slot[kc]] = e;
  }
}

// one round of the large groups (65 .. 8 192 rows): a workgroup per group, keys and rows in LDS, a bitonic network over them
// (two instances over the same list: groups of up to 1 024 rows with 8 KB of LDS, many to a CU, and the few larger ones with 64 KB; a
// workgroup whose entry belongs to the other instance leaves at once)
template <uint32_t LO, uint32_t CAP>
__global__ void __launch_bounds__(256) k_bz_gl_sort_wg(const GlEntry *__restrict__ list, const uint32_t *__restrict__ cnt_p, uint32_t h, uint32_t *__restrict__ sa,
                                                       const uint32_t *__restrict__ clr, uint32_t *__restrict__ clw, SubTab T, GlLists next) {
  __shared__ uint32_t K[CAP], V[CAP];
  __shared__ uint32_t lds[GL_NCL * 5];
  __shared__ uint32_t l17[17];
  if (blockIdx.x >= *cnt_p) return;
  uint32_t first, rows, sb;
  gl_unpack(list[blockIdx.x], first, rows, sb);
  if (rows <= LO || rows > CAP) return;
  const uint32_t n = T.n[sb], off = T.off[sb];
  const int tid = threadIdx.x;
  if (h >= n) {                                        // (equal rotations: the group goes off the lists, its class in both arrays)
    for (uint32_t i = tid; i < rows; i += 256) clw[sa[first + i]] = first;
    return;
  }
  uint32_t P = 128;
  while (P < rows) P <<= 1;
  for (uint32_t i = tid; i < P; i += 256) {
    uint32_t k = 0xFFFFFFFFu, v = 0xFFFFFFFFu;
    if (i < rows) { v = sa[first + i]; uint32_t l = v - off + h; if (l >= n) l -= n; k = clr[off + l]; }
    K[i] = k; V[i] = v;
  }
  __syncthreads();
  for (uint32_t size = 2; size <= P; size <<= 1)
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t idx = tid; idx < P / 2; idx += 256) {
        const uint32_t a = 2 * idx - (idx & (stride - 1)), b = a + stride;
        const uint32_t ka = K[a], kb = K[b], va = V[a], vb = V[b];
        const bool b_less = kb < ka || (kb == ka && vb < va);
        if (((a & size) == 0) == b_less) { K[a] = kb; K[b] = ka; V[a] = vb; V[b] = va; }   // ascending stretches keep the smaller one in front, descending ones the larger
      }
      __syncthreads();
    }
  // the rows in order; a thread takes a stretch of them: where its stretch's first group starts comes from a max-scan over the threads
  const uint32_t per = (rows + 255) / 256, lo = min((uint32_t)tid * per, rows), hi = min(lo + per, rows);
  uint32_t last_head = 0;                              // (index + 1 of the last row of the stretch that starts a group, 0: none)
  for (uint32_t i = lo; i < hi; i++) if (i == 0 || K[i] != K[i - 1]) last_head = i + 1;
  OpMax mx;
  const uint32_t incl = wg_scan_incl(last_head, l17, mx, nullptr);
  uint32_t before = __shfl_up(incl, 1);
  if ((tid & 63) == 0) { before = 0; for (int q = 0; q < (tid >> 6); q++) before = mx(before, l17[q]); }
  uint32_t start = before ? before - 1 : 0u;           // the group the stretch's first row belongs to starts here (row 0 starts one)
  uint32_t cnt[GL_NCL] = {0u, 0u, 0u, 0u};
  {
    uint32_t s0 = start;
    for (uint32_t i = lo; i < hi; i++) {
      if (i > 0 && K[i] != K[i - 1]) s0 = i;
      if (i + 1 == rows || K[i + 1] != K[i]) cnt[gl_class(i + 1 - s0)]++;        // a group ends behind row i: listed by the thread that holds its last row
    }
  }
  uint32_t slot[GL_NCL];
  gl_reserve(next, cnt, lds, slot);
  for (uint32_t i = lo; i < hi; i++) {
    if (i > 0 && K[i] != K[i - 1]) start = i;
    const uint32_t v = V[i];
    sa[first + i] = v;
    clw[v] = first + start;
    if (i + 1 == rows || K[i + 1] != K[i]) {
      const uint32_t rn = i + 1 - start;
      const int kc = gl_class(rn);
      if (slot[kc] < next.cap[kc]) {
        const GlEntry e = gl_entry(first + start, rn, sb);
        if (kc == 0) next.s[slot[0]] = GlSmall{e.first, e.sb_rows, V[start], rn > 1 ? V[start + 1] : 0u};
        else next.l[kc][slot[kc]] = e;
      }
      slot[kc]++;
    }
  }
}

// rows ordered by `prefix` bytes: a sub-block whose rotations are that short is done (equal rotations stay in one group)
__global__ void k_bz_done(SubTab T, uint32_t prefix, uint8_t *__restrict__ done) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < T.nsb && prefix >= T.n[s]) done[s] = 1;
}
// last column (:266-276): the byte in front of each row's rotation; the original message is the first row of its group
__global__ void __launch_bounds__(1024) k_bz_bwt_out(const uint8_t *__restrict__ rle, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ cl, SubTab T,
                                                     const Tile *__restrict__ tiles, uint8_t *__restrict__ bwt, uint32_t ntiles_x) {
  const uint32_t bx = xcd_tile(ntiles_x);
  if (bx >= ntiles_x) return;
  const Tile t = tiles[bx];
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  constexpr int PER = BW_TILE / 1024;                  // (rows, then bytes, then stores)
  uint32_t r[PER];
  uint8_t b[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) { const uint32_t i = (uint32_t)q * 1024u + threadIdx.x; r[q] = sa[off + t.lo + (i < m ? i : 0u)] - off; }
#pragma unroll
  for (int q = 0; q < PER; q++) b[q] = rle[off + (r[q] == 0 ? n - 1 : r[q] - 1)];
#pragma unroll
  for (int q = 0; q < PER; q++) { const uint32_t i = (uint32_t)q * 1024u + threadIdx.x; if (i < m) bwt[off + t.lo + i] = b[q]; }
  if (t.lo == 0 && threadIdx.x == 0) T.bwt_index[t.sb] = cl[off] - off;
}


// ---------------------------------------------------------------------------------------------------------------
//  MTF + RLE_2 (:320-412).  The move-to-front list in front of a chunk of the last column depends on what came before only
//  through the order of the last occurrences: pass 1 finds each chunk's distinct symbols by recency, pass 2 (one wave per
//  sub-block, chunk after chunk) turns them into the list in front of every chunk, pass 3 runs the plain algorithm inside
//  the chunks, all at once.  Zero runs (:363-379) are independent of the list: a zero is a byte equal to its predecessor.
// ---------------------------------------------------------------------------------------------------------------
constexpr int MTF_CHUNK = 512;
constexpr int MTF_PER_TILE = BW_TILE / MTF_CHUNK;   // 16

// byte -> sequence number among the bytes in use (Prepare_Mapping :322-337); nsym[s] = normal_symbols_in_use
__global__ void __launch_bounds__(64) k_bz_seqmap(SubTab T, uint8_t *__restrict__ seq /*[nsb][256]*/, uint32_t *__restrict__ nsym) {
  const uint32_t s = blockIdx.x;
  const int lane = threadIdx.x;
  for (int k = 0; k < 4; k++) {
    const int b = lane * 4 + k;
    uint32_t cnt = 0;
    for (int w = 0; w < 8; w++) {
      const uint32_t bits = T.inuse[s * 8 + w];
      if (w < (b >> 5)) cnt += __popc(bits);
      else if (w == (b >> 5)) cnt += __popc(bits & ((1u << (b & 31)) - 1u));
    }
    seq[s * 256 + b] = (uint8_t)cnt;
  }
  if (lane == 0) { uint32_t tot = 0; for (int w = 0; w < 8; w++) tot += __popc(T.inuse[s * 8 + w]); nsym[s] = tot; }
}

// pass 1: per chunk, its distinct symbols most recent first (rec[chunk][..cnt]) and their set (bm[chunk][8])
__global__ void __launch_bounds__(64) k_bz_mtf_recency(const uint8_t *__restrict__ bwt, SubTab T, const Tile *__restrict__ tiles, uint32_t ntiles,
                                                       const uint8_t *__restrict__ seq, uint8_t *__restrict__ rec, uint32_t *__restrict__ bm,
                                                       uint32_t *__restrict__ cnt) {
  __shared__ uint32_t seen[8 * 64];
  const uint32_t slot = blockIdx.x * 64 + threadIdx.x, ti = slot / MTF_PER_TILE, j = slot % MTF_PER_TILE;
  const int lane = threadIdx.x;
  if (ti >= ntiles) return;
  const Tile t = tiles[ti];
  const uint32_t n = T.n[t.sb], lo = t.lo + j * MTF_CHUNK;
  if (lo >= n) { cnt[slot] = 0; return; }
  const uint32_t m = min((uint32_t)MTF_CHUNK, n - lo);
  const uint8_t *src = bwt + T.off[t.sb] + lo, *sq = seq + t.sb * 256;
  for (int w = 0; w < 8; w++) seen[w * 64 + lane] = 0;
  uint32_t c = 0, acc = 0;
  uint32_t *out = (uint32_t *)(rec + (uint64_t)slot * 256);
  // (the chunk sixteen bytes per load, the distinct symbols four per store: a lane's chunk is 512 bytes of its own, and as one byte load and one
  // byte store per step every step was a memory round trip that also waited for the store before it.  The last piece may read a few bytes
  // past the chunk: they lie inside the buffer and are not looked at.)
  for (int i0 = (int)((m - 1) & ~15u); i0 >= 0; i0 -= 16) {
    uint32_t in[4];
    __builtin_memcpy(in, src + i0, 16);
#pragma unroll
    for (int q = 15; q >= 0; q--) {
      if ((uint32_t)(i0 + q) < m) {
        const uint32_t y = sq[(in[q >> 2] >> (8 * (q & 3))) & 0xFFu];
        const uint32_t w = seen[(y >> 5) * 64 + lane];
        if (!((w >> (y & 31)) & 1u)) {
          seen[(y >> 5) * 64 + lane] = w | (1u << (y & 31));
          acc |= y << (8 * (c & 3u));
          if ((++c & 3u) == 0) { out[(c >> 2) - 1] = acc; acc = 0; }
        }
      }
    }
  }
  if (c & 3u) out[c >> 2] = acc;
  for (int w = 0; w < 8; w++) bm[(uint64_t)slot * 8 + w] = seen[w * 64 + lane];
  cnt[slot] = c;
}

// pass 2: one wave per sub-block; lists[slot][256] = the list in front of the chunk
__global__ void __launch_bounds__(64) k_bz_mtf_lists(SubTab T, const uint32_t *__restrict__ first_tile, const uint8_t *__restrict__ rec,
                                                     const uint32_t *__restrict__ bm, const uint32_t *__restrict__ cnt, uint8_t *__restrict__ lists) {
  __shared__ __align__(16) uint8_t L[2][256];
  __shared__ uint32_t B[8];
  const uint32_t s = blockIdx.x, n = T.n[s];
  const int lane = threadIdx.x;
  const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
  const uint64_t slot0 = (uint64_t)first_tile[s] * MTF_PER_TILE;
  for (int k = 0; k < 4; k++) L[0][k * 64 + lane] = (uint8_t)(k * 64 + lane);
  int cur = 0;
  wave_sync();
  for (uint32_t ch = 0; ch < nchunks; ch++) {
    const uint64_t slot = slot0 + ch;
    ((uint32_t *)(lists + slot * 256))[lane] = ((const uint32_t *)L[cur])[lane];
    const uint32_t r = cnt[slot];
    if (lane < 8) B[lane] = bm[slot * 8 + lane];
    wave_sync();
    uint32_t kept = r;
    for (int k = 0; k < 4; k++) {
      const uint32_t y = L[cur][k * 64 + lane];
      const bool keep = !((B[y >> 5] >> (y & 31)) & 1u);
      const unsigned long long mask = __ballot(keep);
      if (keep) L[cur ^ 1][kept + __popcll(mask & ((1ull << lane) - 1ull))] = (uint8_t)y;
      kept += __popcll(mask);
    }
    for (uint32_t i = lane; i < r; i += 64) L[cur ^ 1][i] = rec[slot * 256 + i];
    cur ^= 1;
    wave_sync();
  }
}

// pass 3: the move-to-front indices of a chunk.  A lane's list is 64 words (+ 1 of padding, so that lanes at the same
// position hit different banks); the search compares four entries per step and the shift moves four per step.
__global__ void __launch_bounds__(64) k_bz_mtf_apply(const uint8_t *__restrict__ bwt, SubTab T, const Tile *__restrict__ tiles, uint32_t ntiles,
                                                     const uint8_t *__restrict__ seq, const uint8_t *__restrict__ lists, uint8_t *__restrict__ idx_out) {
  __shared__ uint32_t L[65 * 64];
  const uint32_t slot = blockIdx.x * 64 + threadIdx.x, ti = slot / MTF_PER_TILE, j = slot % MTF_PER_TILE;
  const int lane = threadIdx.x;
  if (ti >= ntiles) return;
  const Tile t = tiles[ti];
  const uint32_t n = T.n[t.sb], lo = t.lo + j * MTF_CHUNK;
  if (lo >= n) return;
  const uint32_t m = min((uint32_t)MTF_CHUNK, n - lo);
  const uint8_t *src = bwt + T.off[t.sb] + lo, *sq = seq + t.sb * 256;
  const uint32_t *l0 = (const uint32_t *)(lists + (uint64_t)slot * 256);
  uint8_t *dst = idx_out + T.off[t.sb] + lo;
  uint32_t *my = L + lane * 65;                                   // entry p = byte p & 3 of word p >> 2
  for (int i = 0; i < 64; i++) my[i] = l0[i];
  // (sixteen bytes per load and per store: see k_bz_mtf_recency)
  for (uint32_t i0 = 0; i0 < m; i0 += 16) {
    uint32_t in[4], outw[4] = {0u, 0u, 0u, 0u};
    __builtin_memcpy(in, src + i0, 16);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      if (i0 + (uint32_t)q < m) {
        const uint32_t y = sq[(in[q >> 2] >> (8 * (q & 3))) & 0xFFu], yy = y * 0x01010101u;
        uint32_t wi = 0, wv = my[0], x = wv ^ yy;
        while (!((x - 0x01010101u) & ~x & 0x80808080u)) { wi++; wv = my[wi]; x = wv ^ yy; }       // no zero byte: y is not among these four
        const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;      // lowest set bit marks the first zero byte (no borrow can reach below it)
        const uint32_t b = (uint32_t)(__ffs((int)z) - 1) >> 3;        // its byte
        const uint32_t idx = wi * 4 + b;
        if (idx) {
          // entries 0 .. idx - 1 move up by one, y goes to the front: whole words below wi, part of word wi
          const uint32_t keep = b == 3 ? 0u : (wv & (0xFFFFFFFFu << (8 * (b + 1))));
          uint32_t carry = y;
          for (uint32_t k = 0; k < wi; k++) { const uint32_t w = my[k]; my[k] = (w << 8) | carry; carry = w >> 24; }
          const uint32_t lowmask = (1u << (8 * b)) - 1u;               // the bytes of word wi in front of y
          my[wi] = keep | ((((wv & lowmask) << 8) | carry) & ((b == 3) ? 0xFFFFFFFFu : ((1u << (8 * (b + 1))) - 1u)));
        }
        outw[q >> 2] |= idx << (8 * (q & 3));
      }
    }
    if (i0 + 16 <= m) __builtin_memcpy(dst + i0, outw, 16);
    else for (uint32_t q = 0; i0 + q < m; q++) dst[i0 + q] = (uint8_t)(outw[q >> 2] >> (8 * (q & 3)));
  }
}

// zero runs: position after the last non-zero index in front of g (the run's first element)
__global__ void __launch_bounds__(1024) k_bz_rle2_val(const uint8_t *__restrict__ idx, SubTab T, const Tile *__restrict__ tiles, uint32_t *__restrict__ v) {
  const Tile t = tiles[blockIdx.x];
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  uint8_t x[BW_TILE / 1024];                           // (loads before stores)
#pragma unroll
  for (int q = 0; q < BW_TILE / 1024; q++) { const uint32_t i = (uint32_t)q * 1024u + threadIdx.x; x[q] = idx[off + t.lo + (i < m ? i : 0u)]; }
#pragma unroll
  for (int q = 0; q < BW_TILE / 1024; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, l = t.lo + i, g = off + l;
    if (i < m) v[g] = x[q] != 0 ? g + 1 : (l == 0 ? g : 0u);
  }
}
// symbols produced at element g: 1 for a non-zero index, the run's binary digits at the end of a zero run (:363-379)
__global__ void __launch_bounds__(1024) k_bz_rle2_cnt(const uint8_t *__restrict__ idx, const uint32_t *__restrict__ rs, SubTab T, const Tile *__restrict__ tiles,
                                                      uint32_t *__restrict__ cnt) {
  const Tile t = tiles[blockIdx.x];
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  constexpr int PER = BW_TILE / 1024;                  // (loads before stores)
  uint8_t x0[PER], x1[PER];
  uint32_t r[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, l = t.lo + (i < m ? i : 0u), g = off + l;
    x0[q] = idx[g]; x1[q] = l + 1 == n ? 1 : idx[g + 1]; r[q] = rs[g];
  }
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, g = off + t.lo + i;
    uint32_t c = 1;
    if (x0[q] == 0) {
      c = 0;
      if (x1[q] != 0) { const uint32_t run = g - r[q] + 1; c = 31 - __clz(run + 1); }
    }
    if (i < m) cnt[g] = c;
  }
}
// symbol space: sub-block s owns [soff[s], soff[s] + mtf_n[s]); the sub-blocks follow each other with at most two symbols between them
__global__ void k_bz_sym_layout(SubTab T, const uint32_t *__restrict__ P, const uint32_t *__restrict__ nsym, uint32_t *__restrict__ soff, uint32_t *__restrict__ mtf_n,
                                uint16_t *__restrict__ sym) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= T.nsb) return;
  const uint32_t a = P[T.off[s]], b = P[T.off[s] + T.n[s]];
  const uint32_t o = a + 2 * s + (a & 1u);             // even, so that a group's fifty symbols can be read as 25 words
  soff[s] = o;
  mtf_n[s] = b - a + 1;
  sym[o + (b - a)] = (uint16_t)(nsym[s] + 1);          // EOB = last_symbol_in_use (:331-335)
}
__global__ void __launch_bounds__(1024) k_bz_rle2_emit(const uint8_t *__restrict__ idx, const uint32_t *__restrict__ rs, const uint32_t *__restrict__ P, SubTab T,
                                                       const Tile *__restrict__ tiles, const uint32_t *__restrict__ soff, uint16_t *__restrict__ sym) {
  const Tile t = tiles[blockIdx.x];
  const uint32_t n = T.n[t.sb], off = T.off[t.sb], m = min((uint32_t)BW_TILE, n - t.lo);
  const uint32_t base = soff[t.sb] - P[off];
  constexpr int PER = BW_TILE / 1024;                  // (loads before stores)
  uint8_t x0[PER], x1[PER];
  uint32_t r[PER], pp[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, l = t.lo + (i < m ? i : 0u), g = off + l;
    x0[q] = idx[g]; x1[q] = l + 1 == n ? 1 : idx[g + 1]; r[q] = rs[g]; pp[q] = P[g];
  }
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const uint32_t i = (uint32_t)q * 1024u + threadIdx.x, g = off + t.lo + i;
    if (i < m) {
      const uint32_t x = x0[q];
      if (x != 0) sym[base + pp[q]] = (uint16_t)(x + 1);
      else if (x1[q] != 0) {
        uint32_t rc = g - r[q] + 2, o = base + pp[q];
        do { sym[o++] = (uint16_t)(rc & 1u); rc >>= 1; } while (rc >= 2);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
//  Entropy coders (:418-1010).  One workgroup per sub-block runs the reference's brute-force search as it stands: for every
//  (max code length, sample width, number of coders) on the list, start from the ranking of the groups of 50 symbols
//  (:555-635), then up to ten rounds of {code lengths per cluster, every group to its cheapest coder} (:781-811), and keep
//  the cheapest total (:926-950).  The order of equal keys in the ranking is the one GNAT's heap sort leaves (DESIGN.md §9,
//  exactness notes): k_bz_rank replays that sort, one lane per ranking, in LDS.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BZ_GROUP = 50;
constexpr int BZ_MAX_SEL = 18002;          // 1 + 900 005 / 50
constexpr int BZ_LSTRIDE = 260;            // row stride of the code length tables

struct EntTab {
  const uint16_t *sym; const uint32_t *soff, *mtf_n, *nsym, *sel_off;
  uint16_t *rank_idx;          // [2][selcap]: 1-based group numbers in ranking order, per sample width
  uint32_t selcap;
  unsigned long long *gcost;   // [selcap]: bits of a group under each of the six coders, ten bits apart
  unsigned long long *gcbest;  // ... of the best candidate so far
  uint8_t *sel;                // [selcap]: the coder of each group (1 ..)
  uint8_t *lens;               // [nsb][6][260]
  uint32_t *res;               // [nsb][8]: coders, max code length, sample width, groups, data bits, selector bits, tree bits, block bits
  uint32_t *deflist;           // [selcap]: the groups that change coder in a round
  const uint32_t *order;       // sub-blocks, largest first: workgroup b takes sub-block order[b] (the long ones must not start last)
  // the long sub-blocks' search is cut into its four independent chains (max code length x sample width): a workgroup each, results in slots of their own
  uint8_t *csel; unsigned long long *cgcbest; uint8_t *clens; uint32_t *cres;   // [4][selcap], [4][selcap], [nsb][4][6][260], [nsb][4][8]
  unsigned long long *dbg;     // [nsb][8] clock counts (profiling aid): histogram, code lengths, costs, chain, passes, rounds, constructs, total
  int option;                  // 0 / 1 / 2 = block_100k / 400k / 900k
};

__global__ void __launch_bounds__(64) k_bz_rank(EntTab E, uint32_t first) {
  extern __shared__ __align__(8) uint32_t P[];                 // key << 16 | group number, 1-based: a node's two sons are one 8-byte read
  const uint32_t s = E.order[first + blockIdx.x];
  const int lane = threadIdx.x, w = blockIdx.y;                // one ranking (sample width) per workgroup
  const int width = E.option == 2 ? 3 + w : 4;
  const uint32_t m = E.mtf_n[s], ns = 1 + (m - 1) / BZ_GROUP;
  const uint16_t *sym = E.sym + E.soff[s];
  const int eob = (int)E.nsym[s] + 1;
  const int last_sampled = eob - 1 < width - 1 ? eob - 1 : width - 1;          // :595
  // keys (:586-600): how many of a group's symbols are among the sampled ones.  The wave reads the symbols as words, neighbouring lanes
  // neighbouring words (a lane per group was fifty 2-byte loads 100 bytes apart per group), and adds a word's count to its group's entry.
  for (uint32_t g = lane; g < ns; g += 64) P[g + 1] = g + 1;
  wave_sync();
  {
    const uint32_t *sw = (const uint32_t *)sym;                   // (a sub-block's symbols start at an even offset: k_bz_sym_layout)
    const uint32_t nw = (m + 1) / 2;
    for (uint32_t w0 = 0; w0 < nw; w0 += 64 * 8) {
      uint32_t x[8];
#pragma unroll
      for (int q = 0; q < 8; q++) { const uint32_t w = w0 + (uint32_t)q * 64 + lane; x[q] = sw[w < nw ? w : 0u]; }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t w = w0 + (uint32_t)q * 64 + lane;
        if (w < nw) {
          const uint32_t a = (int)(x[q] & 0xFFFFu) <= last_sampled ? 1u : 0u, b = (2 * w + 1 < m && (int)(x[q] >> 16) <= last_sampled) ? 1u : 0u;
          if (a + b) atomicAdd(&P[w / (BZ_GROUP / 2) + 1], (a + b) << 16);        // (a word lies inside one group: fifty is even)
        }
      }
    }
  }
  wave_sync();
  {
    // GNAT's heap sort (a-cgcaso.adb: the hole sinks to a leaf along the larger sons -- the left one on a tie --, then the saved
    // element rises from there), replayed by the WAVE: the sort is one chain of dependent reads, ~14 levels deep per sift with
    // one lane at work; here 63 lanes read the son pairs of the six levels under the hole at once, the path through them is a
    // scalar walk over two ballots, the lanes on the path move their chosen son up with one write, and the rise reads all the
    // fathers at once: three round trips per sift instead of fourteen and more.  Same moves, same array after every sift.
    int Max = (int)ns;
    const int rel = lane + 1, dl = 31 - __clz(rel), jl = rel - (1 << dl);      // this lane's node of the subtree under the hole
    auto sift = [&](const int S, const uint32_t t) {
      int C = S;
      for (;;) {
        const int node = (C << dl) + jl, son = 2 * node;
        const bool has = lane < 63 && son <= Max;
        uint2 two = make_uint2(0u, 0u);
        if (has) two = *(const uint2 *)&P[son];
        const bool right = has && son < Max && (two.x >> 16) < (two.y >> 16);
        const unsigned long long hm = __ballot(has), rm = __ballot(right);
        int r = 1;
        unsigned long long pm = 0;
        for (int k = 0; k < 6; k++) {
          if (!((hm >> (r - 1)) & 1ull)) break;
          pm |= 1ull << (r - 1);
          r = 2 * r + (int)((rm >> (r - 1)) & 1ull);
        }
        if ((pm >> lane) & 1ull) P[node] = right ? two.y : two.x;
        const int dr = 31 - __clz(r);
        C = (C << dr) + (r - (1 << dr));
        if (dr < 6) break;
      }
      // the rise: the fathers of C up to S, all at once
      const int nf = __clz(S) - __clz(C);
      uint32_t f = 0;
      if (lane < nf) f = P[C >> (lane + 1)];
      const unsigned long long um = __ballot(lane < nf && (f >> 16) < (t >> 16));
      const int u = (int)__builtin_ctzll(~um);
      if (lane < u) P[C >> lane] = f;
      if (lane == 0) P[C >> u] = t;
    };
    for (int J = Max / 2; J >= 1; J--) { const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)P[J]); sift(J, t); }
    while (Max > 1) {
      const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)P[Max]);
      const uint32_t top = P[1];
      if (lane == 0) P[Max] = top;
      Max--;
      sift(1, t);
    }
  }
  wave_sync();
  uint16_t *out = E.rank_idx + (size_t)w * E.selcap + E.sel_off[s];
  for (uint32_t i = lane; i < ns; i += 64) out[i] = (uint16_t)(P[i + 1] & 0xFFFFu);
}

constexpr int EN_THREADS = 512;
#ifndef ZADA_EN_SMALL_THREADS
#define ZADA_EN_SMALL_THREADS 256
#endif
constexpr int EN_SMALL_THREADS = ZADA_EN_SMALL_THREADS;
constexpr uint32_t EN_SMALL_SEL = 2040;     // groups a small workgroup takes

// Move-to-front order of the (at most six) coders, and the effect of a stretch of groups on it: the coders chosen in the
// stretch, most recent first.  Both are lists of nibbles (entry j in bits 4j .. 4j+3) with their length in bits 28 .. 30; the
// stretch B after the stretch (or state) A leaves B's list followed by what A's list holds besides.
__device__ __forceinline__ uint32_t mtf_compose(uint32_t A, uint32_t B) {
  const uint32_t kA = A >> 28, kB = B >> 28;
  uint32_t inB = 0;
  for (uint32_t j = 0; j < kB; j++) inB |= 1u << ((B >> (4 * j)) & 15u);
  uint32_t list = B & 0x0FFFFFFFu, k = kB;
  for (uint32_t j = 0; j < kA; j++) {
    const uint32_t x = (A >> (4 * j)) & 15u;
    if (!((inB >> x) & 1u)) { list |= x << (4 * k); k++; }
  }
  return list | (k << 28);
}

// THREADS lanes per sub-block, room for MAXSEL groups: the many short sub-blocks (the segments of the splitting tactics) take a
// smaller workgroup with less LDS, so that twice as many share a CU -- the search is a chain of short dependent phases, the
// more sub-blocks in flight the better.  `first`: offset in the (largest-first) order of the sub-blocks.
template <int THREADS, int MAXSEL, bool SPLIT>
// (second bound: waves per SIMD that the LDS footprint lets a CU hold -- 3 / 5 / 8 workgroups of 8 / 4 / 2 waves)
__global__ void __launch_bounds__(THREADS, THREADS >= 512 ? 6 : THREADS >= 256 ? 5 : 4) k_bz_entropy(EntTab E, uint32_t nsb, uint32_t first) {
  constexpr int NW = THREADS / 64, NLL = NW < 6 ? NW : 6;                // waves, and how many of them make code lengths at a time
  __shared__ uint32_t sel4[(MAXSEL + 7) / 8 + 2];  // the coder of each group, four bits each (LDS decides how many workgroups a CU holds)
  __shared__ uint32_t freq[6 * BZ_LSTRIDE];          // symbol counts per cluster
  __shared__ uint8_t lens[6 * BZ_LSTRIDE];
  __shared__ unsigned long long lens6[BZ_LSTRIDE];   // the six coders' lengths of a symbol, ten bits apart
  __shared__ __align__(16) uint8_t scratch[NLL * LLHC_WAVE_SCRATCH];
  __shared__ uint32_t wtot[NW];
  __shared__ uint32_t red[16];
  __shared__ uint32_t dirty;                         // clusters whose counts changed since their code lengths were made
  const uint32_t s = E.order[first + blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t m = E.mtf_n[s], A = E.nsym[s] + 2, ns = 1 + (m - 1) / BZ_GROUP;
  const uint16_t *sym = E.sym + E.soff[s];
  const uint32_t so = E.sel_off[s];
  // SPLIT: this workgroup runs ONE of the four chains (max code length, sample width) = blockIdx.y of the search, in scratch and result
  // slots of its own.  The chains do not depend on each other: what a construct hands the next one is `low`, and the first construct of a
  // chain (six coders) is on every list (:900-925), so it runs whatever `low` says.  k_bz_pick keeps the best chain, the first in the
  // reference's order among equals (:926-950 compares with "<").
  const uint32_t chn = SPLIT ? blockIdx.y : 0u;
  unsigned long long *gc = (unsigned long long *)E.gcost + (size_t)chn * E.selcap + so;
  uint32_t *deflist = E.deflist + (size_t)chn * E.selcap + so;
  uint8_t *sel_out = SPLIT ? E.csel + (size_t)chn * E.selcap + so : E.sel + so;
  unsigned long long *gcbest_out = SPLIT ? E.cgcbest + (size_t)chn * E.selcap + so : E.gcbest + so;
  uint8_t *lens_out = SPLIT ? E.clens + ((size_t)s * 4 + chn) * 6 * BZ_LSTRIDE : E.lens + (size_t)s * 6 * BZ_LSTRIDE;
  uint32_t *res_out = SPLIT ? E.cres + ((size_t)s * 4 + chn) * 8 : E.res + (size_t)s * 8;
  const uint32_t G = (((ns + THREADS - 1) / THREADS) + 7) & ~7u;             // groups per thread in the chain: whole words of sel4
  auto sel_get = [&](uint32_t g) -> uint32_t { return (sel4[g >> 3] >> (4 * (g & 7))) & 15u; };
  const uint32_t g0 = min((uint32_t)tid * G, ns), g1 = min(g0 + G, ns);
  unsigned long long t_hist = 0, t_llhc = 0, t_cost = 0, t_chain = 0, n_pass = 0, n_round = 0;
  const unsigned long long t_begin = wall_clock64();

  // counts of one group's symbols go to (sign > 0) or leave (sign < 0) a cluster; the four most frequent symbols (the two
  // run digits and the first two move-to-front ranks) are gathered in registers first: they would queue up at the LDS
  // a group's fifty symbols as 25 words, all loads in flight together (the symbol space starts at an even offset and is padded)
  auto load_group = [&](uint32_t g, uint32_t (&v)[25]) {
    const uint32_t *p = (const uint32_t *)(sym + (size_t)g * BZ_GROUP);
#pragma unroll
    for (int k = 0; k < 25; k++) v[k] = p[k];
  };
  // counts of one group's symbols leave cluster `from` and go to cluster `to` (either may be none: 6); the four most frequent
  // symbols (the two run digits and the first two move-to-front ranks) are gathered in registers first: they would queue up at the LDS
  auto count_group = [&](uint32_t g, uint32_t from, uint32_t to) {
    const uint32_t cnt = min((uint32_t)BZ_GROUP, m - g * BZ_GROUP);
    uint32_t v[25];
    load_group(g, v);
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t *ff = freq + (from < 6 ? from : 0) * BZ_LSTRIDE, *ft = freq + (to < 6 ? to : 0) * BZ_LSTRIDE;
#pragma unroll
    for (int k = 0; k < 50; k++) {
      if ((uint32_t)k < cnt) {
        const uint32_t y = (v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        if (y == 0) c0++; else if (y == 1) c1++; else if (y == 2) c2++; else if (y == 3) c3++;
        else { if (from < 6) atomicSub(&ff[y], 1u); if (to < 6) atomicAdd(&ft[y], 1u); }
      }
    }
    if (from < 6) { if (c0) atomicSub(&ff[0], c0); if (c1) atomicSub(&ff[1], c1); if (c2) atomicSub(&ff[2], c2); if (c3) atomicSub(&ff[3], c3); }
    if (to < 6) { if (c0) atomicAdd(&ft[0], c0); if (c1) atomicAdd(&ft[1], c1); if (c2) atomicAdd(&ft[2], c2); if (c3) atomicAdd(&ft[3], c3); }
  };
  auto histogram = [&]() {                                                            // :637-655
    const unsigned long long ta = wall_clock64();
    for (int i = tid; i < 6 * BZ_LSTRIDE; i += THREADS) freq[i] = 0;
    __syncthreads();
    for (uint32_t g = tid; g < ns; g += THREADS) count_group(g, 6, sel_get(g) - 1u);
    if (tid == 0) dirty = 63u;
    __syncthreads();
    t_hist += wall_clock64() - ta;
  };
  auto define_descriptors = [&](int ec, int ml) {                                   // :656-659, :497-517
    const unsigned long long tb = wall_clock64();
    // the counts as the code length procedure reads them (after Avoid_Zeros) sit in that procedure's own scratch, in the part it
    // only uses once the counts have been read
    const uint32_t todo = dirty;                                                      // same counts, same code lengths: only the clusters that changed
    if (w < NLL) {
      for (int k = w; k < ec; k += NLL) {
        if (!((todo >> k) & 1u)) continue;
        uint8_t *sc = scratch + w * LLHC_WAVE_SCRATCH;
        uint32_t *f = (uint32_t *)(sc + 1728);
        for (uint32_t a = lane; a < (uint32_t)BZ_LSTRIDE; a += 64) f[a] = freq[k * BZ_LSTRIDE + a];
        wave_sync();
        int zeroes = 0;
        for (uint32_t base = 0; base < A; base += 64) { const uint32_t a = base + lane; zeroes += __popcll(__ballot(a < A && f[a] == 0)); }
        if (zeroes > 0) {                                                               // Avoid_Zeros :436-460
          for (uint32_t a = lane; a < A; a += 64) { const uint32_t v = f[a]; f[a] = zeroes <= 100 ? (v < 1 ? 1u : v) : (v == 0 ? 1u : v * 2); }
        }
        wave_sync();
        uint8_t *bl = lens + k * BZ_LSTRIDE;
        if (ml == 15) llhc_wave<15>(f, (int)A, bl, sc, lane);
        else if (ml == 16) llhc_wave<16>(f, (int)A, bl, sc, lane);
        else llhc_wave<17>(f, (int)A, bl, sc, lane);
        wave_sync();
      }
    }
    __syncthreads();
    if (tid == 0) dirty = 0;
    for (uint32_t y = tid; y < A; y += THREADS) {
      unsigned long long v = 0;
      for (int cl = 0; cl < ec; cl++) v |= (unsigned long long)lens[cl * BZ_LSTRIDE + y] << (10 * cl);
      lens6[y] = v;
    }
    __syncthreads();
    t_llhc += wall_clock64() - tb;
  };
  auto compute_costs = [&]() {                                                        // the bit_count of :735-739, all groups at once
    const unsigned long long ta = wall_clock64();
    for (uint32_t g = tid; g < ns; g += THREADS) {
      const uint32_t cnt = min((uint32_t)BZ_GROUP, m - g * BZ_GROUP);
      uint32_t v[25];
      load_group(g, v);
      unsigned long long acc = 0;
#pragma unroll
      for (int k = 0; k < 50; k++) if ((uint32_t)k < cnt) acc += lens6[(v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
      gc[g] = acc;
    }
    __syncthreads();
    t_cost += wall_clock64() - ta;
  };
  // Simulate_Entropy_Coding_Variants_and_Reclassify (:664-752).  The choice of a group depends on the groups before it
  // only through the move-to-front order of the coders.  Every thread runs its stretch of groups from a guessed order; a
  // scan over the stretches' effects gives every stretch the order it really starts from; stretches that guessed wrong run
  // again, until nothing moves (the choices rarely depend on the order, so this takes two or three turns).
  auto chain = [&](int ec, uint32_t &defectors, uint32_t &selbits) {
    const unsigned long long ta = wall_clock64();
    const uint32_t ident = 0x654321u | (6u << 28);
    uint32_t used = ident & 0x0FFFFFFFu, outv = 0, def = 0, selc = 0, chosen = 0;
    unsigned long long nw0 = 0, nw1 = 0;                                               // the stretch's new coders, three bits each
    auto run = [&]() {
      uint32_t perm = used;
      def = 0; selc = 0; chosen = 0; nw0 = 0; nw1 = 0;
      // the groups' costs eight at a time: the loads are in flight together (one by one, each was a round trip to L2 in a serial loop)
      for (uint32_t gb = g0; gb < g1; gb += 8) {
        unsigned long long cps[8];
#pragma unroll
        for (int q = 0; q < 8; q++) cps[q] = gb + q < g1 ? gc[gb + q] : 0ull;
        const uint32_t olds = sel4[gb >> 3];                                           // (g0 and the stretch length are multiples of eight)
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t g = gb + q;
          if (g < g1) {
            const unsigned long long cp = cps[q];
            const uint32_t old = (olds >> (4 * q)) & 15u;
            uint32_t bestc = 0xFFFFFFFFu, bestcl = old, bestpos = 1;
            for (int j = 0; j < ec; j++) {
              const uint32_t cl = (perm >> (4 * j)) & 15u;
              const uint32_t cost = (uint32_t)((cp >> (10 * (cl - 1))) & 1023u) + (uint32_t)j + 1;
              if (cost < bestc || (cost == bestc && cl < bestcl)) { bestc = cost; bestcl = cl; bestpos = (uint32_t)j + 1; }
            }
            if (bestcl != old) def++;
            selc += bestpos;
            chosen |= 1u << bestcl;
            const uint32_t lowm = (1u << (4 * (bestpos - 1))) - 1u, upto = (1u << (4 * bestpos)) - 1u;
            perm = (perm & ~upto) | ((perm & lowm) << 4) | bestcl;
            { const uint32_t qq = g - g0; if (qq < 21) nw0 |= (unsigned long long)bestcl << (3 * qq); else nw1 |= (unsigned long long)bestcl << (3 * (qq - 21)); }
          }
        }
      }
      const uint32_t k = __popc(chosen);
      outv = (perm & ((1u << (4 * k)) - 1u)) | (k << 28);                              // the stretch's effect
    };
    bool need = true;
    for (;;) {
      if (need) run();
      n_pass++;
      // exclusive scan of the effects over the threads, from the initial order
      uint32_t incl = outv;
      for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl = mtf_compose(t, incl); }
      if (lane == 63) wtot[w] = incl;
      __syncthreads();
      uint32_t before = ident;
      for (int k = 0; k < w; k++) before = mtf_compose(before, wtot[k]);
      const uint32_t prev = __shfl_up(incl, 1);
      if (lane) before = mtf_compose(before, prev);
      const uint32_t in = before & 0x0FFFFFFFu;
      need = in != used;
      used = in;
      if (!__syncthreads_or(need ? 1 : 0)) break;
    }
    if (tid < 3) red[tid] = 0;
    __syncthreads();
    if (def) atomicAdd(&red[0], def);
    if (selc) atomicAdd(&red[1], selc);
    for (uint32_t gw = g0; gw < g1; gw += 8) {                                          // the groups that change party: listed, ...
      uint32_t word = 0;
      for (uint32_t g = gw; g < gw + 8 && g < g1; g++) {
        const uint32_t q = g - g0, old = sel_get(g);
        const uint32_t nw = (uint32_t)((q < 21 ? nw0 >> (3 * q) : nw1 >> (3 * (q - 21))) & 7u);
        if (nw != old) deflist[atomicAdd(&red[2], 1u)] = g | ((old - 1) << 16) | ((nw - 1) << 20);
        word |= nw << (4 * (g & 7));
      }
      sel4[gw >> 3] = word;
    }
    __syncthreads();
    const uint32_t ndef = red[2];
    for (uint32_t i = tid; i < ndef; i += THREADS) {                                 // ... they take their counts along, all threads sharing the work
      const uint32_t v = deflist[i];
      count_group(v & 0xFFFFu, (v >> 16) & 15u, (v >> 20) & 15u);
      atomicOr(&dirty, (1u << ((v >> 16) & 15u)) | (1u << ((v >> 20) & 15u)));
    }
    __syncthreads();
    defectors = red[0]; selbits = red[1];
    __syncthreads();
    t_chain += wall_clock64() - ta;
  };
  auto cluster_statistics = [&](int ec) -> bool {                                     // :756-779
    if (tid < 8) red[8 + tid] = 0;
    __syncthreads();
    for (uint32_t g = tid; g < ns; g += THREADS) atomicAdd(&red[8 + sel_get(g)], 1u);
    __syncthreads();
    const uint32_t uniform_usage = ns / (uint32_t)ec;
    bool low = false;
    for (int cidx = 1; cidx <= ec; cidx++) if (red[8 + cidx] < uniform_usage / 2) low = true;
    __syncthreads();
    return low;
  };
  struct Cost { uint32_t data, selb, tree; };
  auto construct = [&](int ec, int ml, int widx, bool &low) -> Cost {                 // :781-811 and :813-890
    {                                                                                 // Initial_Clustering_by_Rank :574-591
      const uint32_t attr = ec == 2 ? 0x12u : ec == 3 ? 0x213u : ec == 4 ? 0x3124u : ec == 5 ? 0x42135u : 0x531246u;
      const uint16_t *rk = E.rank_idx + (size_t)widx * E.selcap + so;
      for (uint32_t i = tid; i < (ns + 7) / 8 + 1; i += THREADS) sel4[i] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < ns; i += THREADS) {
        uint32_t a32 = 1;
        while ((uint32_t)(a32 * ns / (uint32_t)ec) < i + 1) a32++;
        const uint32_t g = rk[i] - 1u;
        atomicOr(&sel4[g >> 3], ((attr >> (4 * (a32 - 1))) & 15u) << (4 * (g & 7)));
      }
      __syncthreads();
    }
    histogram();
    uint32_t defectors = 0, selbits = 0;
    for (int it = 1; it <= 10; it++) {
      n_round++;
      define_descriptors(ec, ml);
      compute_costs();
      chain(ec, defectors, selbits);
      if (defectors == 0) break;
    }
    if (defectors > 0) { define_descriptors(ec, ml); compute_costs(); }
    low = cluster_statistics(ec);
    // Compute_Total_Entropy_Cost: data, selectors (their move-to-front indices were summed by the last chain), code lengths
    if (tid < 2) red[tid] = 0;
    __syncthreads();
    uint32_t d = 0;
    for (uint32_t g = tid; g < ns; g += THREADS) d += (uint32_t)((gc[g] >> (10 * (sel_get(g) - 1))) & 1023u);
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o);
    if (lane == 0 && d) atomicAdd(&red[0], d);
    uint32_t tb = 0;
    for (uint32_t q = tid; q < (uint32_t)ec * A; q += THREADS) {
      const uint32_t cl = q / A, i = q % A;
      const int cur = lens[cl * BZ_LSTRIDE + i], prev = lens[cl * BZ_LSTRIDE + (i ? i - 1 : 0)];
      tb += 1u + 2u * (uint32_t)(cur > prev ? cur - prev : prev - cur) + (i == 0 ? 5u : 0u);
    }
    for (int o = 32; o > 0; o >>= 1) tb += __shfl_down(tb, o);
    if (lane == 0 && tb) atomicAdd(&red[1], tb);
    __syncthreads();
    Cost r{red[0], selbits, red[1]};
    __syncthreads();
    return r;
  };

  int mcl[2], nmcl, cc[4], ncc, nsw;
  if (E.option == 2) {                                                                // :900-925
    mcl[0] = 15; mcl[1] = 17; nmcl = 2; nsw = 2;
    if (m <= 5000) { cc[0] = 2; cc[1] = 3; cc[2] = 6; ncc = 3; }
    else if (m <= 10000) { cc[0] = 3; cc[1] = 4; cc[2] = 6; ncc = 3; }
    else { cc[0] = 3; cc[1] = 4; cc[2] = 5; cc[3] = 6; ncc = 4; }
  } else { mcl[0] = 16; nmcl = 1; nsw = 1; cc[0] = 4; cc[1] = 6; ncc = 2; }
  bool low = false;
  uint32_t best_cost = 0x7FFFFFFFu;
  int best_ec = 2, best_ml = mcl[0], best_w = 0;
  if (SPLIT && tid == 0) { res_out[4] = 0x7FFFFFFFu; res_out[5] = 0; res_out[6] = 0; }      // (a chain that constructs nothing: never the best)
  for (int a = SPLIT ? (int)(chn >> 1) : 0; a < (SPLIT ? (int)(chn >> 1) + 1 : nmcl); a++)
    for (int b = SPLIT ? (int)(chn & 1u) : 0; b < (SPLIT ? (int)(chn & 1u) + 1 : nsw); b++)
      for (int ec = 6; ec >= 2; ec--) {
        bool listed = false;
        for (int q = 0; q < ncc; q++) listed |= cc[q] == ec;
        if (!(low || listed)) continue;
        const Cost k = construct(ec, mcl[a], b, low);
        const uint32_t cost = k.data + k.selb + k.tree;
        if (cost < best_cost) {      // :926-950; the reference constructs the winner once more at the end (:952-960): same input, same result, so it is kept here
          best_cost = cost; best_ec = ec; best_ml = mcl[a]; best_w = b;
          for (uint32_t g = tid; g < ns; g += THREADS) { sel_out[g] = (uint8_t)sel_get(g); gcbest_out[g] = gc[g]; }
          for (int i = tid; i < 6 * BZ_LSTRIDE; i += THREADS) lens_out[i] = (i / BZ_LSTRIDE < ec && (uint32_t)(i % BZ_LSTRIDE) < A) ? lens[i] : 0;
          if (tid == 0) {
            uint32_t *r = res_out;
            r[0] = (uint32_t)ec; r[1] = (uint32_t)mcl[a]; r[2] = (uint32_t)(E.option == 2 ? 3 + b : 4); r[3] = ns;
            r[4] = k.data; r[5] = k.selb; r[6] = k.tree; r[7] = 0;
          }
          __syncthreads();
        }
      }
  (void)best_ec; (void)best_ml; (void)best_w;
  if (tid == 0 && chn == 0) {
    unsigned long long *d = E.dbg + (size_t)s * 8;
    d[0] = t_hist; d[1] = t_llhc; d[2] = t_cost; d[3] = t_chain; d[4] = n_pass; d[5] = n_round; d[6] = t_begin; d[7] = wall_clock64() - t_begin;
  }
}

// the best of a long sub-block's four chains (k_bz_entropy<.., true>) becomes its result: the first in the reference's order among equals
__global__ void __launch_bounds__(256) k_bz_pick(EntTab E) {
  const uint32_t s = E.order[blockIdx.x];
  const uint32_t so = E.sel_off[s];
  const uint32_t *cr = E.cres + (size_t)s * 4 * 8;
  uint32_t best = 0;
  unsigned long long bc = ~0ull;
  for (uint32_t q = 0; q < 4; q++) {
    const unsigned long long cost = (unsigned long long)cr[q * 8 + 4] + cr[q * 8 + 5] + cr[q * 8 + 6];
    if (cost < bc) { bc = cost; best = q; }
  }
  const uint32_t ns = cr[best * 8 + 3];
  const uint8_t *cs = E.csel + (size_t)best * E.selcap + so;
  const unsigned long long *cg = E.cgcbest + (size_t)best * E.selcap + so;
  for (uint32_t g = threadIdx.x; g < ns; g += 256) { E.sel[so + g] = cs[g]; E.gcbest[so + g] = cg[g]; }
  const uint8_t *cl = E.clens + ((size_t)s * 4 + best) * 6 * BZ_LSTRIDE;
  for (int i = threadIdx.x; i < 6 * BZ_LSTRIDE; i += 256) E.lens[(size_t)s * 6 * BZ_LSTRIDE + i] = cl[i];
  if (threadIdx.x < 8) E.res[(size_t)s * 8 + threadIdx.x] = cr[best * 8 + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------
//  Output (:1014-1116; Put_Bits is most significant bit first, bzip2-buffers.adb:21-33).  Bit strings are built in 32-bit
//  words whose bit 31 comes first; the final copy swaps them into bytes.
// ---------------------------------------------------------------------------------------------------------------
struct BitW {      // appends to a zero-initialised word array with atomicOr, so that neighbours may share a word
  uint32_t *w; uint64_t acc; uint32_t fill; uint64_t word;
  __device__ __forceinline__ void start(uint32_t *base, uint64_t bitpos) { w = base; word = bitpos >> 5; fill = (uint32_t)(bitpos & 31u); acc = 0; }
  __device__ __forceinline__ void put(uint32_t code, uint32_t len) {        // len <= 32
    acc = (acc << len) | code; fill += len;
    if (fill >= 32) { fill -= 32; const uint32_t x = (uint32_t)(acc >> fill); if (x) atomicOr(&w[word], x); word++; acc &= (1ull << fill) - 1ull; }
  }
  __device__ __forceinline__ void flush() { if (fill) { const uint32_t x = (uint32_t)(acc << (32 - fill)); if (x) atomicOr(&w[word], x); } }
};

constexpr uint32_t BZ_FIXED_HEAD_BITS = 48 + 32 + 1 + 24;

// bits of every part of a block; res[7] = the block's size
__global__ void k_bz_block_bits(SubTab T, uint32_t *__restrict__ res) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= T.nsb) return;
  uint32_t used16 = 0;
  for (int i = 0; i < 16; i++) used16 += ((T.inuse[s * 8 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu) ? 1u : 0u;
  uint32_t *r = res + (size_t)s * 8;
  r[7] = BZ_FIXED_HEAD_BITS + 16 + 16 * used16 + 3 + 15 + r[5] + r[6] + r[4];
}

// block header, mapping table, selectors, code lengths: one wave per sub-block, lane 0 writes
__global__ void __launch_bounds__(64) k_bz_emit_head(SubTab T, EntTab E, const uint32_t *__restrict__ woff, uint32_t *__restrict__ words) {
  const uint32_t s = blockIdx.x;
  if (threadIdx.x != 0) return;
  const uint32_t *r = E.res + (size_t)s * 8;
  const uint32_t ec = r[0], ns = r[3], A = E.nsym[s] + 2;
  BitW o; o.start(words + woff[s], 0);
  o.put(0x314159u, 24); o.put(0x265359u, 24);                       // "1AY&SY"
  o.put(T.crc[s], 32);
  o.put(0, 1);
  o.put(T.bwt_index[s], 24);
  uint32_t used16 = 0;
  for (int i = 0; i < 16; i++) if ((T.inuse[s * 8 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu) used16 |= 1u << i;
  for (int i = 0; i < 16; i++) o.put((used16 >> i) & 1u, 1);
  for (int i = 0; i < 16; i++) if ((used16 >> i) & 1u) {
    const uint32_t bits = (T.inuse[s * 8 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
    for (int j = 0; j < 16; j++) o.put((bits >> j) & 1u, 1);
  }
  o.put(ec, 3);
  o.put(ns, 15);
  {                                                                   // Put_Selectors :1052-1079
    uint32_t perm = 0x654321u;
    const uint8_t *sel = E.sel + E.sel_off[s];
    for (uint32_t g = 0; g < ns; g++) {
      const uint32_t v = sel[g];
      uint32_t pos = 1;
      while (((perm >> (4 * (pos - 1))) & 15u) != v) pos++;
      const uint32_t lowm = (1u << (4 * (pos - 1))) - 1u, upto = (1u << (4 * pos)) - 1u;
      perm = (perm & ~upto) | ((perm & lowm) << 4) | v;
      o.put(((1u << (pos - 1)) - 1u) << 1, pos);                      // pos - 1 ones, then a zero
    }
  }
  const uint8_t *lens = E.lens + (size_t)s * 6 * BZ_LSTRIDE;
  for (uint32_t cdr = 0; cdr < ec; cdr++) {                           // Put_Huffman_Bit_Lengths :1081-1105
    int cur = lens[cdr * BZ_LSTRIDE];
    o.put((uint32_t)cur, 5);
    for (uint32_t i = 0; i < A; i++) {
      const int nw = lens[cdr * BZ_LSTRIDE + i];
      while (cur != nw) { if (cur < nw) { cur++; o.put(2, 2); } else { cur--; o.put(3, 2); } }
      o.put(0, 1);
    }
  }
  o.flush();
}

// Entropy_Output :1115-1138: one workgroup per sub-block; the groups' bit offsets by a scan, every thread writes its groups
__global__ void __launch_bounds__(EN_THREADS) k_bz_emit_data(SubTab T, EntTab E, const uint32_t *__restrict__ woff, uint32_t *__restrict__ words) {
  __shared__ uint32_t code[6 * BZ_LSTRIDE];
  __shared__ uint8_t lens[6 * BZ_LSTRIDE];
  __shared__ uint32_t l17[17];
  const uint32_t s = blockIdx.x;
  const int tid = threadIdx.x;
  const uint32_t *r = E.res + (size_t)s * 8;
  const uint32_t ec = r[0], ml = r[1], ns = r[3], A = E.nsym[s] + 2, m = E.mtf_n[s];
  const uint16_t *sym = E.sym + E.soff[s];
  const uint8_t *sel = E.sel + E.sel_off[s];
  const unsigned long long *gc = E.gcbest + E.sel_off[s];
  for (int i = tid; i < 6 * BZ_LSTRIDE; i += EN_THREADS) lens[i] = E.lens[(size_t)s * 6 * BZ_LSTRIDE + i];
  __syncthreads();
  if ((uint32_t)tid < ec) {                                           // Prepare_Codes (huffman-encoding.adb:45-80), bit order kept
    uint32_t bl_count[24], next_code[24];
    for (int i = 0; i < 24; i++) { bl_count[i] = 0; next_code[i] = 0; }
    for (uint32_t i = 0; i < A; i++) bl_count[lens[tid * BZ_LSTRIDE + i]]++;
    uint32_t cd = 0;
    for (uint32_t b = 1; b <= ml; b++) { cd = (cd + bl_count[b - 1]) * 2; next_code[b] = cd; }
    for (uint32_t i = 0; i < A; i++) { const uint32_t bl = lens[tid * BZ_LSTRIDE + i]; code[tid * BZ_LSTRIDE + i] = bl ? next_code[bl]++ : 0u; }
  }
  __syncthreads();
  const uint32_t G = (ns + EN_THREADS - 1) / EN_THREADS;
  const uint32_t g0 = min((uint32_t)tid * G, ns), g1 = min(g0 + G, ns);
  uint32_t mine = 0;
  for (uint32_t g = g0; g < g1; g++) mine += (uint32_t)((gc[g] >> (10 * (sel[g] - 1))) & 1023u);
  OpSum sm;
  const uint32_t incl = wg_scan_incl(mine, l17, sm, nullptr);
  const uint32_t head_bits = r[7] - r[4];
  BitW o; o.start(words + woff[s], (uint64_t)head_bits + incl - mine);
  for (uint32_t g = g0; g < g1; g++) {
    const uint32_t cl = sel[g] - 1, cnt = min((uint32_t)BZ_GROUP, m - g * BZ_GROUP);
    for (uint32_t k = 0; k < cnt; k++) { const uint32_t y = sym[g * BZ_GROUP + k]; o.put(code[cl * BZ_LSTRIDE + y], lens[cl * BZ_LSTRIDE + y]); }
  }
  o.flush();
}

// A job copies `bits` bits from src (bit 0 = bit 31 of src word 0) to the destination at bit `dpos`
struct CopyJob { uint64_t src_word; uint64_t dpos; uint64_t bits; uint64_t first_dst_word; };
__global__ void __launch_bounds__(256) k_bz_assemble(const CopyJob *__restrict__ jobs, const uint64_t *__restrict__ job_first /* running count of destination words */,
                                                     uint32_t njobs, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) {
  const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;      // index in the concatenation of the jobs' destination word ranges
  if (q >= job_first[njobs]) return;
  uint32_t lo = 0, hi = njobs;                                        // job with job_first[j] <= q < job_first[j + 1]
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (job_first[mid] <= q) lo = mid; else hi = mid; }
  const CopyJob J = jobs[lo];
  const uint64_t dw = J.first_dst_word + (q - job_first[lo]);        // destination word
  // destination bits [dw * 32, dw * 32 + 32) intersected with [dpos, dpos + bits)
  const int64_t a = (int64_t)(dw * 32) - (int64_t)J.dpos;            // source bit of the word's first bit (may be negative)
  uint32_t out = 0;
  const uint32_t *sp = src + J.src_word;
  const int64_t nb = (int64_t)J.bits;
  // word made of source bits a .. a + 31
  const int64_t k = a >= 0 ? a >> 5 : -((-a + 31) >> 5);
  const int sh = (int)(a - k * 32);                                   // 0 .. 31
  const int64_t nwords = (nb + 31) >> 5;
  const uint32_t w0 = (k >= 0 && k < nwords) ? sp[k] : 0u, w1 = (k + 1 >= 0 && k + 1 < nwords) ? sp[k + 1] : 0u;
  out = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
  // mask off bits outside [0, nb)
  if (a < 0) { const int z = (int)(-a); out = z >= 32 ? 0u : (out & (0xFFFFFFFFu >> z)); }
  if (a + 32 > nb) { const int64_t keep = nb - a; out = keep <= 0 ? 0u : (out & ~((keep >= 32) ? 0u : (0xFFFFFFFFu >> keep))); }
  if (out) atomicOr(&dst[dw], out);
}
__global__ void k_bz_words_to_bytes(const uint32_t *__restrict__ words, uint64_t nwords, uint32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) out[i] = __builtin_bswap32(words[i]);
}

// ---------------------------------------------------------------------------------------------------------------
//  Data acquisition (:1161-1209): a block ends one byte after the piece that takes the simulated RLE_1 size to capacity - 5,
//  after ten capacities of raw bytes, or with the stream.  With E[q] = RLE_1 bytes of the pieces that end before q (pieces
//  phased from the stream's runs) the end is a search in E, except inside the block's first run, whose pieces restart at
//  the block's first byte.  One wave walks the blocks; the searches are 64-ary.
// ---------------------------------------------------------------------------------------------------------------
struct FRunStart1 {     // p + 1 where a run of equal bytes starts, else 0; its inclusive max-scan is RS1[p] = start of p's run + 1
  const uint8_t *in;
  __device__ __forceinline__ uint32_t operator()(uint64_t p) const { return (p == 0 || in[p] != in[p - 1]) ? (uint32_t)p + 1u : 0u; }
};
struct FPieceEnd {      // RLE_1 bytes of the piece that ends at p (0 if none ends there)
  const uint8_t *in; const uint32_t *rs1; uint64_t n;
  __device__ __forceinline__ uint32_t operator()(uint64_t p) const {
    if (p >= n) return 0u;
    const uint32_t r = (uint32_t)((p - (rs1[p] - 1u)) % 259u);
    const bool end = r == 258u || p + 1 == n || in[p + 1] != in[p];
    return end ? (r + 1 < 4 ? r + 1 : 5u) : 0u;
  }
};
// first index in [lo, hi) with arr[index] >= target, else hi (arr non-decreasing); the whole wave calls it
__device__ __forceinline__ uint64_t wave_lower_bound(const uint32_t *__restrict__ arr, uint64_t lo, uint64_t hi, uint64_t target, int lane) {
  while (hi - lo > 64) {
    const uint64_t step = (hi - lo + 63) / 64, idx = lo + (uint64_t)lane * step;
    const bool pred = idx >= hi || (uint64_t)arr[idx] >= target;
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) { lo = lo + 63 * step + 1; continue; }
    const int f = __ffsll((long long)mask) - 1;
    if (f == 0) return lo;
    const uint64_t nlo = lo + (uint64_t)(f - 1) * step + 1, nhi = lo + (uint64_t)f * step;
    lo = nlo; hi = nhi < hi ? nhi : hi;
  }
  const uint64_t idx = lo + lane;
  const bool pred = idx >= hi || (uint64_t)arr[idx] >= target;
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return hi;                        // a full last stretch of 64 with no hit
  const uint64_t r = lo + (uint64_t)(__ffsll((long long)mask) - 1);
  return r < hi ? r : hi;
}
__device__ __forceinline__ uint32_t rle1_size_of_run(uint64_t L) {      // a run of L equal bytes from a block's first byte on
  const uint64_t k = (L - 1) / 259, rem = L - 259 * k;
  return (uint32_t)(5 * k + (rem < 4 ? rem : 5));
}
__global__ void __launch_bounds__(64) k_bz_acquire(const uint32_t *__restrict__ rs1, const uint32_t *__restrict__ E, uint64_t n, int64_t size_hint,
                                                   int32_t block_capacity, float f_lo, float f_hi, uint64_t *__restrict__ bstart, uint32_t *__restrict__ blen,
                                                   uint32_t cap_blocks, uint32_t *__restrict__ count) {
  const int lane = threadIdx.x;
  uint64_t pos = 0;
  uint32_t nb = 0;
  do {
    const int64_t stream_rest = size_hint < 0 ? -1 : size_hint - (int64_t)pos;                 // :1411-1423
    const float fr = (float)stream_rest;
    const int32_t cap = (fr >= f_lo && fr <= f_hi) ? (int32_t)(stream_rest / 2) : block_capacity;
    const uint64_t raw_last = 10ull * (uint64_t)cap;
    const uint64_t limit = n - pos < raw_last ? n - pos : raw_last;
    uint64_t consumed = limit;
    if (limit > 0) {
      const int64_t t = (int64_t)cap - 5;                // bytes are read while the simulated size stays below t
      const uint64_t s = pos, b = wave_lower_bound(rs1, s + 1, s + limit, (uint64_t)rs1[s] + 1, lane);   // where the block's first run ends
      const uint64_t L0 = b - s;
      const uint64_t i1 = 259ull * (uint64_t)((t + 4) / 5) + 1;    // first count at which whole pieces of the first run alone reach t
      if (i1 <= L0) consumed = i1 < limit ? i1 : limit;
      else if (b < s + limit) {
        const int64_t tq = t - (int64_t)rle1_size_of_run(L0) + (int64_t)E[b];
        const uint64_t q = tq <= (int64_t)E[b] ? b : wave_lower_bound(E, b, s + limit, (uint64_t)tq, lane);
        consumed = q < s + limit ? q - s + 1 : limit;
      }
    }
    if (lane == 0 && nb < cap_blocks) { bstart[nb] = pos; blen[nb] = (uint32_t)consumed; }
    nb++;
    pos += consumed;
  } while (pos < n);
  if (lane == 0) *count = nb;
}

// ---------------------------------------------------------------------------------------------------------------
//  Data_Segmentation.Segment_by_Entropy (data_segmentation.adb:39-105) for the tactics segmented_1 / segmented_2
//  (:1255-1263: thresholds 0.6 / 0.4, distances 4000 / 8000, window 16000).  The running entropy is a chain of `digits 15`
//  additions, so it is walked in order, one lane per block; both tactics read the same chain.  etab[f] = -(p Log p) for
//  p = f / window, tabulated by the host with the C library's log, the function GNAT's Log maps to.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SEG_WINDOW = 16000;
__device__ __forceinline__ double lane_value(double x, int j) {      // x of lane j (j uniform)
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), j), hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
  return __hiloint2double(hi, lo);
}
// One wave per block.  64 steps at a time: the lanes work out, together, the byte counts every step sees (the counts at the
// start of the stretch plus the bytes that came and went in the steps before it: a broadcast loop over the lanes) and fetch
// the four table values of their step; then the additions run in order, every lane following.
__global__ void __launch_bounds__(64) k_bz_segment(const uint8_t *__restrict__ in, const uint64_t *__restrict__ bstart, const uint32_t *__restrict__ blen,
                                                   uint32_t nblk, const double *__restrict__ etab, double thr1, double thr2,
                                                   const uint32_t *__restrict__ seg_off /*[2 nblk + 1]*/, uint32_t *__restrict__ seg, uint32_t *__restrict__ seg_cnt) {
  __shared__ uint32_t freq[256];
  const int lane = threadIdx.x;
  const uint32_t blk = blockIdx.x;
  for (int b = lane; b < 256; b += 64) freq[b] = 0;
  const uint8_t *buf = in + bstart[blk];
  const uint32_t len = blen[blk];
  uint32_t *s1 = seg + seg_off[2 * blk], *s2 = seg + seg_off[2 * blk + 1];
  uint32_t n1 = 0, n2 = 0;
  wave_sync();
  if (len > SEG_WINDOW + 4000) {
    const bool t2 = len > SEG_WINDOW + 8000;
    for (uint32_t i = lane; i < SEG_WINDOW; i += 64) atomicAdd(&freq[buf[i]], 1u);       // steps 1 .. window: the window fills up
    wave_sync();
    double entropy = 0.0;
    {
      double v[4];
      uint32_t fb[4];
      for (int q = 0; q < 4; q++) { fb[q] = freq[q * 64 + lane]; v[q] = etab[fb[q]]; }
      for (int q = 0; q < 4; q++)
        for (int j = 0; j < 64; j++) {
          const double x = lane_value(v[q], j);
          if (__builtin_amdgcn_readlane((int)fb[q], j)) entropy = entropy + x;
        }
    }
    double mark1 = entropy, mark2 = entropy;
    uint32_t im1 = 1, im2 = 1;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t i0 = SEG_WINDOW + 1; i0 <= len; i0 += 64) {
      const uint32_t cnt = min(64u, len - i0 + 1), i = i0 + lane;
      const bool valid = (uint32_t)lane < cnt;
      const uint32_t xin = valid ? buf[i - 1] : 0u, xout = valid ? buf[i - SEG_WINDOW - 1] : 0u;
      const uint32_t f0in = freq[xin], f0out = freq[xout];
      wave_sync();
      // how often the bytes of this lane's step came in / went out in the steps before it of this stretch: the lanes holding a given
      // byte are the AND, over its eight bits, of the ballots of that bit (or their complements)
      unsigned long long m_ii = ~0ull, m_oi = ~0ull, m_io = ~0ull, m_oo = ~0ull;
#pragma unroll
      for (int bt = 0; bt < 8; bt++) {
        const unsigned long long bi = __ballot((xin >> bt) & 1u), bo = __ballot((xout >> bt) & 1u);
        const bool mi = (xin >> bt) & 1u, mo = (xout >> bt) & 1u;
        m_ii &= mi ? bi : ~bi; m_oi &= mi ? bo : ~bo;
        m_io &= mo ? bi : ~bi; m_oo &= mo ? bo : ~bo;
      }
      const unsigned long long before = lt & (cnt == 64 ? ~0ull : (1ull << cnt) - 1ull);
      const uint32_t ii = (uint32_t)__popcll(m_ii & before), oi = (uint32_t)__popcll(m_oi & before);
      const uint32_t io = (uint32_t)__popcll(m_io & before), oo = (uint32_t)__popcll(m_oo & before);
      if (valid) { atomicAdd(&freq[xin], 1u); atomicSub(&freq[xout], 1u); }
      const uint32_t f = f0in + ii - oi + 1, g = f0out + io + (xin == xout ? 1u : 0u) - oo;
      double A = 0.0, Bv = 0.0, C = 0.0, D = 0.0;
      if (valid) { A = etab[f - 1]; Bv = etab[f]; C = etab[g]; D = etab[g - 1]; }       // etab[0] = 0.0
      wave_sync();
      // the chain of additions, in the reference's order; lane j keeps the value after step j.  The tests against the marks follow for
      // all steps at once: a mark moves at most once in a stretch (64 steps are fewer than the 4 000 / 8 000 between two marks)
      double e_mine = 0.0;
      for (int j = 0; j < (int)cnt; j++) {
        entropy = entropy - lane_value(A, j);
        entropy = entropy + lane_value(Bv, j);
        entropy = entropy - lane_value(C, j);
        entropy = entropy + lane_value(D, j);           // 0.0 where the reference adds nothing (the byte has left the window): the same value
        if (lane == j) e_mine = entropy;
      }
      const uint32_t sp = i0 + (uint32_t)lane - SEG_WINDOW;
      const unsigned long long h1 = __ballot(valid && fabs(e_mine - mark1) > thr1 && sp > im1 && sp - im1 > 4000u);
      if (h1) {
        const int j = __builtin_ctzll(h1);
        if (lane == 0) s1[n1] = i0 + (uint32_t)j - SEG_WINDOW;
        n1++; im1 = i0 + (uint32_t)j - SEG_WINDOW; mark1 = lane_value(e_mine, j);
      }
      if (t2) {
        const unsigned long long h2 = __ballot(valid && fabs(e_mine - mark2) > thr2 && sp > im2 && sp - im2 > 8000u);
        if (h2) {
          const int j = __builtin_ctzll(h2);
          if (lane == 0) s2[n2] = i0 + (uint32_t)j - SEG_WINDOW;
          n2++; im2 = i0 + (uint32_t)j - SEG_WINDOW; mark2 = lane_value(e_mine, j);
        }
      }
    }
  }
  if (lane == 0) {
    if (len > 0) { s1[n1++] = len; s2[n2++] = len; }
    seg_cnt[2 * blk] = n1; seg_cnt[2 * blk + 1] = n2;
  }
}

// ---------------------------------------------------------------------------------------------------------------
//  host side
// ---------------------------------------------------------------------------------------------------------------
struct DBuf {
  void *p = nullptr; size_t cap = 0;
  template <class X> X *as() const { return (X *)p; }
};
static int dbuf_ensure(Ctx *c, DBuf &b, size_t bytes) {
  if (bytes <= b.cap && b.p) return 0;
  if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }
  const size_t want = bytes + bytes / 8 + 4096;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { (void)hipGetLastError(); hip_check(c, e, "hipMalloc (bzip2 workspace)"); b.p = nullptr; return ZADA_E_NOMEM; }
  b.cap = want;
  return 0;
}

struct Bz2State {
  // sub-block tables
  DBuf raw_start, raw_len, off, n, inuse, crc, bwt_index, done, scal;
  // tiles
  DBuf rtiles, rtile_first, rtile_val, rtile_crc, rtile_rs, etiles, etile_first;
  // element space
  DBuf rle, bwt, keyA, keyB, valA, valB, cl, hv, hr, H, agg, cv0, cv1, acte, coff, cm, ctiles, ctile_first;
  DBuf gl_s[2], gl_m[2], gl_l[2], gl_w[2], gl_nc, gl_submax, gl_lmode, gl_cnt, gl_tmp, gl_k0, gl_k1, gl_v, gl_st;   // group lists of the late rounds (k_bz_gl_*)
  uint64_t gl_rows = 0;             // groups the lists' rounds of the last batch sorted (profiling aid)
  size_t gl_st_bytes = 0;            // temporary storage of the list sort
  std::vector<uint32_t> h_cm, h_cfirst;
  std::vector<uint64_t> m_hist;     // rows the doubling rounds of the last batch had to sort (profiling aid)
  std::vector<Tile> h_ct;
  // MTF / symbol space
  DBuf seq, nsym, rec, recbm, reccnt, lists, sym, soff, mtf_n;
  // entropy coders and output
  DBuf sel_off, rank_idx, gcost, gcbest, sel, lens, res, woff, words, jobs, job_first, outw, dbg, deflist, order, csel, cgcbest, clens, cres;
  // stream level
  DBuf rs1, epre, bstart, blen, etab, seg_off, seg, seg_cnt, extra;
  bool etab_ready = false;
  Bz2State *slot1 = nullptr;        // the second batch in flight (bz_blocks_encode pipelines the stages of consecutive batches)
  bool no_pipeline = false;         // the device ran out of memory with two batches in flight: one at a time from then on
  hipStream_t st_pipe = nullptr;    // ... and the stream its entropy stage runs on
  hipStream_t st_small = nullptr; hipEvent_t ev_small = nullptr, ev_rank = nullptr;     // the entropy search of the short sub-blocks runs next to the long ones'
  std::vector<uint64_t> trace;      // per block: raw start, raw length, tactic kept, its number of sub-blocks
  // the call in flight: every unique piece's bits are kept until the tactics are chosen
  struct KeptSub { uint64_t bits, woff; uint32_t crc, buf; };
  struct Piece { uint64_t start; uint32_t len, kept; };
  struct BlkPlan { uint64_t start = 0; uint32_t len = 0; std::vector<uint32_t> tac[4]; std::vector<Piece> pieces; };
  std::vector<KeptSub> kept; std::vector<BlkPlan> plans; std::vector<DBuf> kept_bufs;
  uint64_t min_bits_sum = 0;
  std::vector<uint64_t> rg_bstart; std::vector<uint32_t> rg_blen; int rg_option = 2; const uint8_t *rg_in = nullptr;
  std::vector<uint32_t> h_res, h_woff, h_crc;
  uint64_t nwords = 0;
  uint32_t selcap = 0;
  bool rank_attr = false;
  std::vector<DBuf *> all() {
    return {&raw_start, &raw_len, &off, &n, &inuse, &crc, &bwt_index, &done, &scal, &rtiles, &rtile_first, &rtile_val, &rtile_crc, &rtile_rs,
            &etiles, &etile_first, &rle, &bwt, &keyA, &keyB, &valA, &valB, &cl, &hv, &hr, &H, &agg, &cv0, &cv1, &acte, &coff, &cm, &ctiles, &ctile_first, &seq, &nsym, &rec, &recbm, &reccnt, &lists,
            &sym, &soff, &mtf_n, &sel_off, &rank_idx, &gcost, &sel, &lens, &res, &woff, &words, &jobs, &job_first, &outw,
            &rs1, &epre, &bstart, &blen, &etab, &seg_off, &seg, &seg_cnt, &extra, &dbg, &deflist, &order, &gcbest, &csel, &cgcbest, &clens, &cres,
            &gl_s[0], &gl_s[1], &gl_m[0], &gl_m[1], &gl_l[0], &gl_l[1], &gl_w[0], &gl_w[1], &gl_nc, &gl_submax, &gl_lmode, &gl_cnt, &gl_tmp, &gl_k0, &gl_k1, &gl_v, &gl_st};
  }
  // host mirrors of the batch in flight
  std::vector<uint64_t> h_raw_start;
  std::vector<uint32_t> h_raw_len, h_n, h_off;
  uint32_t nsb = 0, ntot = 0, n_etiles = 0;
  int bwt_rounds = 0;
};

static Bz2State *bz_state(Ctx *c) {
  if (!c->bz) c->bz = new Bz2State();
  return (Bz2State *)c->bz;
}
static void bz_free_state(Bz2State *B);
void bz2_destroy(Ctx *c) {
  if (!c->bz) return;
  bz_free_state((Bz2State *)c->bz);
  c->bz = nullptr;
}
static void bz_free_state(Bz2State *B) {
  if (B->slot1) bz_free_state(B->slot1);
  if (B->st_pipe) { hipStreamSynchronize(B->st_pipe); hipStreamDestroy(B->st_pipe); }
  for (DBuf *b : B->all()) if (b->p) hipFree(b->p);
  for (DBuf &b : B->kept_bufs) if (b.p) hipFree(b.p);
  if (B->st_small) { hipStreamSynchronize(B->st_small); hipStreamDestroy(B->st_small); }
  if (B->ev_small) hipEventDestroy(B->ev_small);
  if (B->ev_rank) hipEventDestroy(B->ev_rank);
  delete B;
}

static void build_tiles(const std::vector<uint32_t> &len, uint32_t tile, std::vector<Tile> &tiles, std::vector<uint32_t> &first) {
  tiles.clear(); first.clear();
  for (uint32_t s = 0; s < len.size(); s++) {
    first.push_back((uint32_t)tiles.size());
    for (uint32_t lo = 0; lo < len[s]; lo += tile) tiles.push_back({s, lo});
  }
  first.push_back((uint32_t)tiles.size());
}

static SubTab subtab(Bz2State *B) {
  SubTab T;
  T.raw_start = B->raw_start.as<uint64_t>(); T.raw_len = B->raw_len.as<uint32_t>(); T.off = B->off.as<uint32_t>(); T.n = B->n.as<uint32_t>();
  T.inuse = B->inuse.as<uint32_t>(); T.crc = B->crc.as<uint32_t>(); T.bwt_index = B->bwt_index.as<uint32_t>(); T.nsb = B->nsb;
  return T;
}

// RLE_1, CRC and BWT of a batch of sub-blocks of d_in.  Leaves rle / bwt / tables in the state.
// (B: the state that holds the batch -- the context's own, or its second slot when batches are pipelined; st: the stream of this
// stage; marks: phase timing marks, main thread only)
static int bz_transform(Ctx *c, Bz2State *B, hipStream_t st, bool marks, const uint8_t *d_in, const std::vector<uint64_t> &starts, const std::vector<uint32_t> &lens) {
  const uint32_t nsb = (uint32_t)starts.size();
  B->nsb = nsb; B->h_raw_start = starts; B->h_raw_len = lens;
  int rc;
  if ((rc = dbuf_ensure(c, B->raw_start, 8ull * nsb)) || (rc = dbuf_ensure(c, B->raw_len, 4ull * nsb)) || (rc = dbuf_ensure(c, B->off, 4ull * (nsb + 1))) ||
      (rc = dbuf_ensure(c, B->n, 4ull * nsb)) || (rc = dbuf_ensure(c, B->inuse, 32ull * nsb)) || (rc = dbuf_ensure(c, B->crc, 4ull * nsb)) ||
      (rc = dbuf_ensure(c, B->bwt_index, 4ull * nsb)) || (rc = dbuf_ensure(c, B->done, nsb)) ||
      (rc = dbuf_ensure(c, B->scal, 64))) return rc;
  BZ_HIP(hipMemcpyAsync(B->raw_start.p, starts.data(), 8ull * nsb, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(B->raw_len.p, lens.data(), 4ull * nsb, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemsetAsync(B->inuse.p, 0, 32ull * nsb, st));
  BZ_HIP(hipMemsetAsync(B->done.p, 0, nsb, st));
  BZ_HIP(hipMemsetAsync(B->bwt_index.p, 0, 4ull * nsb, st));
  // raw tiles
  std::vector<Tile> rt; std::vector<uint32_t> rfirst;
  build_tiles(lens, RT_TILE, rt, rfirst);
  const uint32_t nrt = (uint32_t)rt.size();
  if ((rc = dbuf_ensure(c, B->rtiles, sizeof(Tile) * (size_t)nrt)) || (rc = dbuf_ensure(c, B->rtile_first, 4ull * (nsb + 1))) ||
      (rc = dbuf_ensure(c, B->rtile_val, 4ull * nrt)) || (rc = dbuf_ensure(c, B->rtile_crc, 4ull * nrt)) || (rc = dbuf_ensure(c, B->rtile_rs, 4ull * nrt))) return rc;
  BZ_HIP(hipMemcpyAsync(B->rtiles.p, rt.data(), sizeof(Tile) * (size_t)nrt, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(B->rtile_first.p, rfirst.data(), 4ull * (nsb + 1), hipMemcpyHostToDevice, st));
  SubTab T = subtab(B);
  if (nrt) {
    hipLaunchKernelGGL(k_bz_rle_runs, dim3(nrt), dim3(256), 0, st, d_in, T, B->rtiles.as<Tile>(), B->rtile_rs.as<uint32_t>());
    hipLaunchKernelGGL(k_bz_tile_scan_max, dim3(nsb), dim3(64), 0, st, B->rtile_first.as<uint32_t>(), B->rtile_rs.as<uint32_t>());
    hipLaunchKernelGGL(k_bz_rle_count, dim3(nrt), dim3(256), 0, st, d_in, T, B->rtiles.as<Tile>(), B->rtile_rs.as<uint32_t>(), B->rtile_val.as<uint32_t>());
  }
  hipLaunchKernelGGL(k_bz_tile_scan, dim3(nsb), dim3(64), 0, st, B->rtile_first.as<uint32_t>(), B->rtile_val.as<uint32_t>(), B->n.as<uint32_t>());
  B->h_n.resize(nsb);
  BZ_HIP(hipMemcpyAsync(B->h_n.data(), B->n.p, 4ull * nsb, hipMemcpyDeviceToHost, st));
  BZ_HIP(hipStreamSynchronize(st));
  B->h_off.resize(nsb + 1);
  uint64_t tot = 0;
  for (uint32_t s = 0; s < nsb; s++) { B->h_off[s] = (uint32_t)tot; tot += B->h_n[s]; }
  if (tot >= (1ull << 31)) { c->err = "bzip2: batch too large"; return ZADA_E_TOO_LARGE; }
  B->h_off[nsb] = (uint32_t)tot; B->ntot = (uint32_t)tot;
  BZ_HIP(hipMemcpyAsync(B->off.p, B->h_off.data(), 4ull * (nsb + 1), hipMemcpyHostToDevice, st));
  const size_t ne = (size_t)tot + 16;
  if ((rc = dbuf_ensure(c, B->rle, ne)) || (rc = dbuf_ensure(c, B->bwt, ne))) return rc;
  if (nrt) {
    hipLaunchKernelGGL(k_bz_rle_emit, dim3(nrt), dim3(256), 0, st, d_in, T, B->rtiles.as<Tile>(), B->rtile_rs.as<uint32_t>(), B->rtile_val.as<uint32_t>(), B->rle.as<uint8_t>());
    hipLaunchKernelGGL(k_bz_crc_tiles, dim3((nrt + 63) / 64), dim3(64), 0, st, d_in, T, B->rtiles.as<Tile>(), nrt, B->rtile_crc.as<uint32_t>());
  }
  hipLaunchKernelGGL(k_bz_crc_fold, dim3((nsb + 63) / 64), dim3(64), 0, st, T, B->rtile_first.as<uint32_t>(), B->rtile_crc.as<uint32_t>());
  if (marks) c->tmark("bz:rle1");
  // element tiles
  std::vector<Tile> et; std::vector<uint32_t> efirst;
  build_tiles(B->h_n, BW_TILE, et, efirst);
  const uint32_t net = (uint32_t)et.size();
  B->n_etiles = net;
  if ((rc = dbuf_ensure(c, B->etiles, sizeof(Tile) * (size_t)net)) || (rc = dbuf_ensure(c, B->etile_first, 4ull * (nsb + 1)))) return rc;
  BZ_HIP(hipMemcpyAsync(B->etiles.p, et.data(), sizeof(Tile) * (size_t)net, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(B->etile_first.p, efirst.data(), 4ull * (nsb + 1), hipMemcpyHostToDevice, st));
  if ((rc = dbuf_ensure(c, B->keyA, 4 * ne)) || (rc = dbuf_ensure(c, B->keyB, 4 * ne)) || (rc = dbuf_ensure(c, B->valA, 4 * ne)) ||
      (rc = dbuf_ensure(c, B->valB, 4 * ne)) || (rc = dbuf_ensure(c, B->cl, 4 * ne)) || (rc = dbuf_ensure(c, B->hv, 4 * ne)) ||
      (rc = dbuf_ensure(c, B->hr, 4 * ne)) || (rc = dbuf_ensure(c, B->H, 4096ull * (net + nsb) + 4096)) || (rc = dbuf_ensure(c, B->gl_tmp, 4 * ne)) || (rc = dbuf_ensure(c, B->cv0, 4 * ne)) ||
      (rc = dbuf_ensure(c, B->cv1, 4 * ne)) || (rc = dbuf_ensure(c, B->acte, ne + 64)) || (rc = dbuf_ensure(c, B->coff, 4ull * (nsb + 2))) ||
      (rc = dbuf_ensure(c, B->cm, 4ull * (nsb + 2))) || (rc = dbuf_ensure(c, B->ctile_first, 4ull * (nsb + 2))) ||
      (rc = dbuf_ensure(c, B->agg, 4ull * ((1024ull * net + ne) / SC_TILE + 16)))) return rc;
  BZ_HIP(hipStreamSynchronize(st));   // rt / et vectors go out of scope
  B->bwt_rounds = 0;
  if (net == 0) return 0;
  const Tile *ET = B->etiles.as<Tile>();
  const uint32_t *EF = B->etile_first.as<uint32_t>();
  uint8_t *done = B->done.as<uint8_t>();
  uint32_t *keyA = B->keyA.as<uint32_t>(), *keyB = B->keyB.as<uint32_t>(), *valA = B->valA.as<uint32_t>(), *valB = B->valB.as<uint32_t>();
  uint32_t *cv0 = B->cv0.as<uint32_t>(), *cv1 = B->cv1.as<uint32_t>();
  uint8_t *acte = B->acte.as<uint8_t>();                 // one byte per element: it belongs to a group of more than one row
  uint32_t *H = B->H.as<uint32_t>(), *agg = B->agg.as<uint32_t>(), *hv = B->hv.as<uint32_t>(), *hr = B->hr.as<uint32_t>(), *cl = B->cl.as<uint32_t>();
  auto radix = [&](const SubTab &S, const Tile *tl, const uint32_t *tf, uint32_t nt, const uint32_t *ki, const uint32_t *vi, uint32_t *ko, uint32_t *vo, int shift) {
    hipLaunchKernelGGL(k_bz_radix_hist<8>, dim3(xcd_grid(nt)), dim3(1024), 0, st, ki, S, tl, tf, done, shift, H, nt);
    scan_launch<OpSum, false>(st, FArr{H}, 256ull * nt, agg, H, nullptr);
    hipLaunchKernelGGL(k_bz_radix_scatter<8>, dim3(xcd_grid(nt)), dim3(1024), 0, st, ki, vi, S, tl, tf, done, shift, H, ko, vo, nt);
  };
  auto radix10 = [&](const SubTab &S, const Tile *tl, const uint32_t *tf, uint32_t nt, const uint32_t *ki, const uint32_t *vi, uint32_t *ko, uint32_t *vo, int shift) {
    hipLaunchKernelGGL(k_bz_radix_hist<10>, dim3(xcd_grid(nt)), dim3(1024), 0, st, ki, S, tl, tf, done, shift, H, nt);
    scan_launch<OpSum, false>(st, FArr{H}, 1024ull * nt, agg, H, nullptr);
    hipLaunchKernelGGL(k_bz_radix_scatter<10>, dim3(xcd_grid(nt)), dim3(1024), 0, st, ki, vi, S, tl, tf, done, shift, H, ko, vo, nt);
  };
  B->m_hist.clear();
  B->m_hist.push_back(tot);
  // first sort: four bytes
  hipLaunchKernelGGL(k_bz_bwt_init, dim3(xcd_grid(net)), dim3(1024), 0, st, B->rle.as<uint8_t>(), T, ET, keyA, valA, net);
  radix(T, ET, EF, net, keyA, valA, keyB, valB, 0); radix(T, ET, EF, net, keyB, valB, keyA, valA, 8);
  radix(T, ET, EF, net, keyA, valA, keyB, valB, 16); radix(T, ET, EF, net, keyB, valB, keyA, valA, 24);
  hipLaunchKernelGGL(k_bz_heads0, dim3(xcd_grid(net)), dim3(1024), 0, st, keyA, T, ET, done, hv, net);
  scan_launch<OpMax, true>(st, FArr{hv}, tot, agg, hr, nullptr);
  hipLaunchKernelGGL(k_bz_set_class, dim3(xcd_grid(net)), dim3(1024), 0, st, valA, hv, hr, T, ET, cl, acte, net);
  hipLaunchKernelGGL(k_bz_done, dim3((nsb + 255) / 256), dim3(256), 0, st, T, 4u, done);
  SubTab C = T;
  C.off = B->coff.as<uint32_t>(); C.n = B->cm.as<uint32_t>();
  std::vector<uint32_t> &h_cm = B->h_cm;
  std::vector<Tile> &ct = B->h_ct; std::vector<uint32_t> &cfirst = B->h_cfirst;
  // group lists of the late rounds (k_bz_gl_*): two generations, filled by one round and sorted by the next
  // (a list entry packs its sub-block into GL_SB_BITS bits -- gl_entry: a batch of more sub-blocks than that keeps sweeping)
  const bool use_lists = c->knob_bz_lists != 0 && nsb <= (1u << GL_SB_BITS);
  // from which prefix length on sub-blocks may leave the sweeps ("bz_lists"; < 0: by the batch's longest sub-block -- a batch with the blocks of a stream
  // after the round for 16 bytes, the sub-blocks of small entries after the round for 8.  Round 6, on silesia_mix_v2: a stream of 256 MiB 437 against 457 ms
  // with 16 instead of 8 -- most rows are still in groups after eight bytes, distinct content rather than copies, and sorting that many groups one by one costs more
  // than one more sweep; 10 000 entries of 16 KiB 0.39 against 0.40 s, 2 000 of 256 KiB 0.80 against 0.81 s with 8)
  uint32_t longest_sb = 0;
  for (uint32_t v : B->h_n) longest_sb = v > longest_sb ? v : longest_sb;
  const uint32_t lists_from = c->knob_bz_lists > 0 ? (uint32_t)c->knob_bz_lists : (longest_sb >= 300000u ? 16u : 8u);      // (the blocks of a BZip2_2 / _3 stream: 400 k / 900 k)
  GlLists GL[2];
  uint32_t *nc = nullptr, *submax = nullptr, *glcnt = nullptr;
  uint8_t *lmode = nullptr;
  if (use_lists) {
    const uint32_t cap_s = (uint32_t)(tot + 64), cap_m = (uint32_t)(tot / (GL_SMALL + 1) + 64), cap_l = (uint32_t)(tot / (GL_MID + 1) + 64), cap_w = (uint32_t)(tot / (GL_WAVE + 1) + 64);   // (groups of one included: a round lists a row at most once)
    if ((rc = dbuf_ensure(c, B->gl_s[0], sizeof(GlSmall) * (size_t)cap_s)) || (rc = dbuf_ensure(c, B->gl_s[1], sizeof(GlSmall) * (size_t)cap_s)) ||
        (rc = dbuf_ensure(c, B->gl_m[0], sizeof(GlEntry) * (size_t)cap_m)) || (rc = dbuf_ensure(c, B->gl_m[1], sizeof(GlEntry) * (size_t)cap_m)) ||
        (rc = dbuf_ensure(c, B->gl_l[0], sizeof(GlEntry) * (size_t)cap_l)) || (rc = dbuf_ensure(c, B->gl_l[1], sizeof(GlEntry) * (size_t)cap_l)) ||
        (rc = dbuf_ensure(c, B->gl_w[0], sizeof(GlEntry) * (size_t)cap_w)) || (rc = dbuf_ensure(c, B->gl_w[1], sizeof(GlEntry) * (size_t)cap_w)) ||
        (rc = dbuf_ensure(c, B->gl_nc, 4 * ne)) || (rc = dbuf_ensure(c, B->gl_submax, 4ull * nsb + 64)) || (rc = dbuf_ensure(c, B->gl_lmode, nsb + 64)) ||
        (rc = dbuf_ensure(c, B->gl_cnt, 64))) return rc;
    if (c->knob_bz_text_order) {
      const size_t tb = radix_sort_tmp_bytes(cap_s, sizeof(GlSmall));
      if ((rc = dbuf_ensure(c, B->gl_k0, 4ull * cap_s)) || (rc = dbuf_ensure(c, B->gl_k1, 4ull * cap_s)) || (rc = dbuf_ensure(c, B->gl_v, sizeof(GlSmall) * (size_t)cap_s)) ||
          (rc = dbuf_ensure(c, B->gl_st, tb + 256))) return rc;
      B->gl_st_bytes = tb;
    }
    glcnt = B->gl_cnt.as<uint32_t>();
    for (int k = 0; k < 2; k++) GL[k] = GlLists{{nullptr, B->gl_m[k].as<GlEntry>(), B->gl_l[k].as<GlEntry>(), B->gl_w[k].as<GlEntry>()}, B->gl_s[k].as<GlSmall>(), glcnt + 8 * k, {cap_s, cap_m, cap_l, cap_w}};
    nc = B->gl_nc.as<uint32_t>(); submax = B->gl_submax.as<uint32_t>(); lmode = B->gl_lmode.as<uint8_t>();
    BZ_HIP(hipMemsetAsync(glcnt, 0, 64, st));
  }
  uint32_t gl_n[GL_NCL] = {0, 0, 0, 0};                             // entries of the lists of the generation to sort in this round
  int gcur = 0;
  bool swept = true;                                                // sub-blocks are still being swept
  B->gl_rows = 0;
  for (uint32_t h = 4;; h *= 2) {
    uint32_t nct = 0;
    if (swept) {
      hipLaunchKernelGGL(k_bz_filter_count, dim3(xcd_grid(net)), dim3(1024), 0, st, valA, acte, T, ET, done, h, hv, net);
      scan_launch<OpSum, false>(st, FArrPad{hv, net}, (uint64_t)net + 1, agg, hr, nullptr);          // hr[t] = filtered rows before tile t
      hipLaunchKernelGGL(k_bz_counts, dim3((nsb + 255) / 256), dim3(256), 0, st, T, EF, hr, C.off, C.n);
      h_cm.resize(nsb);
      BZ_HIP(hipMemcpyAsync(h_cm.data(), C.n, 4ull * nsb, hipMemcpyDeviceToHost, st));
      BZ_HIP(hipStreamSynchronize(st));
      build_tiles(h_cm, BW_TILE, ct, cfirst);
      nct = (uint32_t)ct.size();
      if (nct == 0) swept = false;
    }
    if (!swept && gl_n[0] + gl_n[1] + gl_n[2] + gl_n[3] == 0) break;
    B->bwt_rounds++;
    if (swept) {
      if ((rc = dbuf_ensure(c, B->ctiles, sizeof(Tile) * (size_t)nct))) return rc;
      BZ_HIP(hipMemcpyAsync(B->ctiles.p, ct.data(), sizeof(Tile) * (size_t)nct, hipMemcpyHostToDevice, st));
      BZ_HIP(hipMemcpyAsync(B->ctile_first.p, cfirst.data(), 4ull * (nsb + 1), hipMemcpyHostToDevice, st));
      const Tile *CT = B->ctiles.as<Tile>();
      const uint32_t *CF = B->ctile_first.as<uint32_t>();
      uint64_t M = 0;
      for (uint32_t s2 = 0; s2 < nsb; s2++) M += h_cm[s2];
      B->m_hist.push_back(M);
      hipLaunchKernelGGL(k_bz_filter_emit, dim3(xcd_grid(net)), dim3(1024), 0, st, valA, acte, cl, T, ET, done, h, hr, keyA, cv0, net);
      // (first rows are below 2^20 -- the block capacity is 900 000 --: two passes of ten bits, by way of valB / gl_tmp into keyB / cv1)
      radix10(C, CT, CF, nct, keyA, cv0, valB, B->gl_tmp.as<uint32_t>(), 0); radix10(C, CT, CF, nct, valB, B->gl_tmp.as<uint32_t>(), keyB, cv1, 10);
      hipLaunchKernelGGL(k_bz_rf_agg, dim3(xcd_grid(nct)), dim3(1024), 0, st, keyB, C, CT, hv, nct);
      scan_launch<OpMax, false>(st, FArr{hv}, nct, agg, hr, nullptr);                                   // hr[t] = last run start before tile t
      hipLaunchKernelGGL(k_bz_place, dim3(xcd_grid(nct)), dim3(1024), 0, st, keyB, cv1, hr, T, C, CT, cl, h, valA, keyA, hv, nct);
      scan_launch<OpMax, false>(st, FArr{hv}, nct, agg, hr, nullptr);                                   // hr[t] = last group start before tile t
      hipLaunchKernelGGL(k_bz_newclass, dim3(xcd_grid(nct)), dim3(1024), 0, st, cv1, keyA, hr, C, CT, cl, acte, nct);
      hipLaunchKernelGGL(k_bz_done, dim3((nsb + 255) / 256), dim3(256), 0, st, T, 2 * h, done);
    } else B->m_hist.push_back(0);
    if (use_lists) {
      const GlLists cur = GL[gcur], nxt = GL[gcur ^ 1];
      // the listed groups, sorted by 2h bytes now; what is left of them goes on the other generation's lists
      // (the listed sub-blocks' classes: read in one array, written in the other; the swept ones keep to cl)
      const uint32_t *clr = gcur ? nc : cl;
      uint32_t *clw = gcur ? cl : nc;
      if (gl_n[0]) hipLaunchKernelGGL(k_bz_gl_sort_small, dim3((gl_n[0] + GLS_THREADS * GLS_PER - 1) / (GLS_THREADS * GLS_PER)), dim3(GLS_THREADS), 0, st, cur.s, cur.cnt + 0, h, valA, clr, clw, T, nxt);
      if (gl_n[1]) hipLaunchKernelGGL((k_bz_gl_sort_team<(int)GL_MID>), dim3((uint32_t)(((uint64_t)gl_n[1] * GL_MID + GLS_THREADS - 1) / GLS_THREADS)), dim3(GLS_THREADS), 0, st, cur.l[1], cur.cnt + 1, h, valA, clr, clw, T, nxt);
      if (gl_n[2]) hipLaunchKernelGGL((k_bz_gl_sort_team<(int)GL_WAVE>), dim3((uint32_t)(((uint64_t)gl_n[2] * GL_WAVE + GLS_THREADS - 1) / GLS_THREADS)), dim3(GLS_THREADS), 0, st, cur.l[2], cur.cnt + 2, h, valA, clr, clw, T, nxt);
      if (gl_n[3]) {
        hipLaunchKernelGGL((k_bz_gl_sort_wg<0u, 1024u>), dim3(gl_n[3]), dim3(256), 0, st, cur.l[3], cur.cnt + 3, h, valA, clr, clw, T, nxt);
        hipLaunchKernelGGL((k_bz_gl_sort_wg<1024u, GL_MAX>), dim3(gl_n[3]), dim3(256), 0, st, cur.l[3], cur.cnt + 3, h, valA, clr, clw, T, nxt);
      }
      // sub-blocks whose unsorted groups have all become small leave the sweeps: their groups (classes of 2h bytes) join the lists
      if (swept && 2 * h >= lists_from) {
        BZ_HIP(hipMemsetAsync(submax, 0, 4ull * nsb, st));
        hipLaunchKernelGGL(k_bz_gl_max, dim3(xcd_grid(net)), dim3(1024), 0, st, valA, cl, T, ET, done, submax, net);
        hipLaunchKernelGGL(k_bz_gl_decide, dim3((nsb + 255) / 256), dim3(256), 0, st, T, done, submax, lmode, (uint32_t)(c->knob_bz_list_rows > 0 && c->knob_bz_list_rows <= (int)GL_MAX ? c->knob_bz_list_rows : (int)GL_MAX));
        hipLaunchKernelGGL(k_bz_gl_build, dim3(xcd_grid(net)), dim3(1024), 0, st, valA, cl, nc, T, ET, lmode, nxt, net);
        hipLaunchKernelGGL(k_bz_gl_leave, dim3((nsb + 255) / 256), dim3(256), 0, st, T, lmode, done);
      }
      uint32_t hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      BZ_HIP(hipMemcpyAsync(hc, nxt.cnt, 32, hipMemcpyDeviceToHost, st));
      BZ_HIP(hipMemsetAsync(cur.cnt, 0, 32, st));                  // (the generation just sorted is the next one to be filled)
      BZ_HIP(hipStreamSynchronize(st));
      if (hc[GL_NCL] || hc[0] > nxt.cap[0] || hc[1] > nxt.cap[1] || hc[2] > nxt.cap[2] || hc[3] > nxt.cap[3]) { c->err = "bzip2: group list overflow"; return ZADA_E_HIP; }
      B->gl_rows += (uint64_t)gl_n[0] + gl_n[1] + gl_n[2] + gl_n[3];
      for (int k = 0; k < GL_NCL; k++) gl_n[k] = hc[k];
      // a list that has just been built (or added to): into text order (see k_bz_gl_keys; the key is an element index below 2^30, and the
      // order inside 256 positions does not matter: they share their sectors anyway)
      if (c->knob_bz_text_order && swept && 2 * h >= lists_from && hc[0] > 1) {
        hipLaunchKernelGGL(k_bz_gl_keys, dim3((hc[0] + 255) / 256), dim3(256), 0, st, nxt.s, hc[0], B->gl_k0.as<uint32_t>());
        if (radix_sort_pairs(c, st, B->gl_st.p, B->gl_st_bytes, B->gl_k0.as<uint32_t>(), B->gl_k1.as<uint32_t>(), nxt.s, B->gl_v.p, sizeof(GlSmall), (size_t)hc[0], 8, 30)) return ZADA_E_HIP;
        BZ_HIP(hipMemcpyAsync(nxt.s, B->gl_v.p, sizeof(GlSmall) * (size_t)hc[0], hipMemcpyDeviceToDevice, st));
      }
      gcur ^= 1;
    }
  }
  hipLaunchKernelGGL(k_bz_bwt_out, dim3(xcd_grid(net)), dim3(1024), 0, st, B->rle.as<uint8_t>(), valA, cl, T, ET, B->bwt.as<uint8_t>(), net);
  if (marks) c->tmark("bz:bwt");
  BZ_HIP(hipGetLastError());
  return 0;
}


// MTF + RLE_2 of the batch that bz_transform has left in the state: symbols (u16) in B->sym, tables soff / mtf_n / nsym / seq
static int bz_mtf(Ctx *c, Bz2State *B, hipStream_t st, bool marks) {
  const uint32_t nsb = B->nsb, net = B->n_etiles, tot = B->ntot;
  const uint64_t slots = (uint64_t)net * MTF_PER_TILE;
  int rc;
  if ((rc = dbuf_ensure(c, B->seq, 256ull * nsb)) || (rc = dbuf_ensure(c, B->nsym, 4ull * nsb)) || (rc = dbuf_ensure(c, B->rec, 256 * slots + 256)) ||
      (rc = dbuf_ensure(c, B->recbm, 32 * slots + 32)) || (rc = dbuf_ensure(c, B->reccnt, 4 * slots + 4)) || (rc = dbuf_ensure(c, B->lists, 256 * slots + 256)) ||
      (rc = dbuf_ensure(c, B->sym, 2ull * ((uint64_t)tot + 2ull * nsb + 64))) || (rc = dbuf_ensure(c, B->soff, 4ull * (nsb + 1))) ||
      (rc = dbuf_ensure(c, B->mtf_n, 4ull * nsb))) return rc;
  SubTab T = subtab(B);
  const Tile *ET = B->etiles.as<Tile>();
  uint8_t *idx = B->rle.as<uint8_t>();      // the RLE_1 bytes are no longer needed: their place takes the move-to-front indices
  uint32_t *hv = B->hv.as<uint32_t>(), *hr = B->hr.as<uint32_t>(), *cnt = B->keyA.as<uint32_t>(), *P = B->keyB.as<uint32_t>(), *agg = B->agg.as<uint32_t>();
  hipLaunchKernelGGL(k_bz_seqmap, dim3(nsb), dim3(64), 0, st, T, B->seq.as<uint8_t>(), B->nsym.as<uint32_t>());
  if (net) {
    const uint32_t nb = (uint32_t)((slots + 63) / 64);
    hipLaunchKernelGGL(k_bz_mtf_recency, dim3(nb), dim3(64), 0, st, B->bwt.as<uint8_t>(), T, ET, net, B->seq.as<uint8_t>(), B->rec.as<uint8_t>(),
                       B->recbm.as<uint32_t>(), B->reccnt.as<uint32_t>());
    hipLaunchKernelGGL(k_bz_mtf_lists, dim3(nsb), dim3(64), 0, st, T, B->etile_first.as<uint32_t>(), B->rec.as<uint8_t>(), B->recbm.as<uint32_t>(),
                       B->reccnt.as<uint32_t>(), B->lists.as<uint8_t>());
    hipLaunchKernelGGL(k_bz_mtf_apply, dim3(nb), dim3(64), 0, st, B->bwt.as<uint8_t>(), T, ET, net, B->seq.as<uint8_t>(), B->lists.as<uint8_t>(), idx);
    hipLaunchKernelGGL(k_bz_rle2_val, dim3(net), dim3(1024), 0, st, idx, T, ET, hv);
    scan_launch<OpMax, true>(st, FArr{hv}, tot, agg, hr, nullptr);
    hipLaunchKernelGGL(k_bz_rle2_cnt, dim3(net), dim3(1024), 0, st, idx, hr, T, ET, cnt);
  }
  scan_launch<OpSum, false>(st, FArrPad{cnt, tot}, (uint64_t)tot + 1, agg, P, nullptr);
  hipLaunchKernelGGL(k_bz_sym_layout, dim3((nsb + 255) / 256), dim3(256), 0, st, T, P, B->nsym.as<uint32_t>(), B->soff.as<uint32_t>(), B->mtf_n.as<uint32_t>(),
                     B->sym.as<uint16_t>());
  if (net) hipLaunchKernelGGL(k_bz_rle2_emit, dim3(net), dim3(1024), 0, st, idx, hr, P, T, ET, B->soff.as<uint32_t>(), B->sym.as<uint16_t>());
  if (marks) c->tmark("bz:mtf");
  BZ_HIP(hipGetLastError());
  return 0;
}

// entropy coders + bit strings of the batch: words at B->words, sub-block s at word h_woff[s], h_res[8 s + 7] bits
static int bz_entropy_emit(Ctx *c, Bz2State *B, hipStream_t st, bool marks, int option) {
  const uint32_t nsb = B->nsb;
  int rc;
  std::vector<uint32_t> so(nsb + 1);
  for (uint32_t s = 0; s <= nsb; s++) so[s] = B->h_off[s] / BZ_GROUP + 2 * s;
  const uint32_t selcap = so[nsb] + 8;
  B->selcap = selcap;
  if ((rc = dbuf_ensure(c, B->sel_off, 4ull * (nsb + 1))) || (rc = dbuf_ensure(c, B->rank_idx, 4ull * selcap)) || (rc = dbuf_ensure(c, B->gcost, 4 * 8ull * selcap)) || (rc = dbuf_ensure(c, B->gcbest, 8ull * selcap)) ||
      (rc = dbuf_ensure(c, B->csel, 4ull * selcap)) || (rc = dbuf_ensure(c, B->cgcbest, 4 * 8ull * selcap)) || (rc = dbuf_ensure(c, B->clens, 4ull * 6 * BZ_LSTRIDE * nsb)) || (rc = dbuf_ensure(c, B->cres, 4 * 32ull * nsb)) ||
      (rc = dbuf_ensure(c, B->sel, selcap)) || (rc = dbuf_ensure(c, B->lens, 6ull * BZ_LSTRIDE * nsb)) || (rc = dbuf_ensure(c, B->res, 32ull * nsb)) ||
      (rc = dbuf_ensure(c, B->woff, 4ull * (nsb + 1))) || (rc = dbuf_ensure(c, B->dbg, 64ull * nsb)) || (rc = dbuf_ensure(c, B->deflist, 4 * 4ull * selcap))) return rc;
  BZ_HIP(hipMemcpyAsync(B->sel_off.p, so.data(), 4ull * (nsb + 1), hipMemcpyHostToDevice, st));
  std::vector<uint32_t> order(nsb);
  for (uint32_t s = 0; s < nsb; s++) order[s] = s;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return B->h_n[a] > B->h_n[b]; });
  if ((rc = dbuf_ensure(c, B->order, 4ull * nsb))) return rc;
  BZ_HIP(hipMemcpyAsync(B->order.p, order.data(), 4ull * nsb, hipMemcpyHostToDevice, st));
  EntTab E;
  E.sym = B->sym.as<uint16_t>(); E.soff = B->soff.as<uint32_t>(); E.mtf_n = B->mtf_n.as<uint32_t>(); E.nsym = B->nsym.as<uint32_t>();
  E.sel_off = B->sel_off.as<uint32_t>(); E.rank_idx = B->rank_idx.as<uint16_t>(); E.selcap = selcap; E.gcost = B->gcost.as<unsigned long long>(); E.gcbest = B->gcbest.as<unsigned long long>();
  E.sel = B->sel.as<uint8_t>(); E.lens = B->lens.as<uint8_t>(); E.res = B->res.as<uint32_t>(); E.option = option; E.dbg = B->dbg.as<unsigned long long>(); E.deflist = B->deflist.as<uint32_t>(); E.order = B->order.as<uint32_t>();
  E.csel = B->csel.as<uint8_t>(); E.cgcbest = B->cgcbest.as<unsigned long long>(); E.clens = B->clens.as<uint8_t>(); E.cres = B->cres.as<uint32_t>();
  SubTab T = subtab(B);
  // the rankings are replayed by single lanes: the small ones (most of them) take little LDS, so that many share a CU
  constexpr uint32_t RANK_SMALL_NS = 2040;
  const size_t rank_lds_big = 4 * (size_t)(BZ_MAX_SEL + 6), rank_lds_small = 4 * (size_t)(RANK_SMALL_NS + 8);
  if (!B->rank_attr) {
    BZ_HIP(hipFuncSetAttribute((const void *)k_bz_rank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_lds_big));
    B->rank_attr = true;
  }
  uint32_t nbig = 0;
  while (nbig < nsb && 1 + B->h_n[order[nbig]] / BZ_GROUP > RANK_SMALL_NS) nbig++;       // order: largest first
  const uint32_t nwid = option == 2 ? 2 : 1;
  // The long sub-blocks (more than 2 040 groups) and the short ones go their own ways from here, each its rankings and then its
  // entropy search, on two streams: the long ones' rankings are a few hundred single waves of pure latency (a lane replays a heap
  // sort of up to 18 000 keys), their search workgroups run for tens of milliseconds -- what they leave idle is taken by the short
  // ones (256-thread workgroups, 30 KB of LDS: five per CU).
  static_assert(RANK_SMALL_NS == EN_SMALL_SEL, "one split of the sub-blocks for the rankings and the search");
  const uint32_t nlarge = c->knob_bz_small_wg == 0 ? nsb : nbig;
  const bool both = nbig > 0 && nsb > nbig && c->knob_bz_small_wg != 0;
  if (both) {
    if (!B->st_small) {
      // (lowest priority: the short sub-blocks fill what the long ones leave, not the other way round)
      int least = 0, greatest = 0;
      BZ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      BZ_HIP(hipStreamCreateWithPriority(&B->st_small, hipStreamNonBlocking, least));
      BZ_HIP(hipEventCreateWithFlags(&B->ev_small, hipEventDisableTiming));
    }
    BZ_HIP(hipEventRecord(B->ev_small, st));                                          // (the symbols are in place)
    BZ_HIP(hipStreamWaitEvent(B->st_small, B->ev_small, 0));
  }
  hipStream_t sts = both ? B->st_small : st;
  // (the long ones in two launches: a quarter of a block has 4 500 groups and should not hold the 72 KB of LDS a whole block's ranking needs)
  constexpr uint32_t RANK_MID_NS = 6000;
  uint32_t nfull = 0;
  while (nfull < nbig && 1 + B->h_n[order[nfull]] / BZ_GROUP > RANK_MID_NS) nfull++;
  if (nfull) hipLaunchKernelGGL(k_bz_rank, dim3(nfull, nwid), dim3(64), rank_lds_big, st, E, 0u);
  if (nbig > nfull) hipLaunchKernelGGL(k_bz_rank, dim3(nbig - nfull, nwid), dim3(64), 4 * (size_t)(RANK_MID_NS + 8), st, E, nfull);
  if (nsb > nbig) hipLaunchKernelGGL(k_bz_rank, dim3(nsb - nbig, nwid), dim3(64), rank_lds_small, sts, E, nbig);
  if (both) {
    // (the short sub-blocks' search waits for the long ones' rankings: its workgroups would fill every CU's LDS, and the rankings'
    // single waves -- 72 KB of LDS each -- would trickle in behind them: 69 ms instead of 20)
    if (!B->ev_rank) BZ_HIP(hipEventCreateWithFlags(&B->ev_rank, hipEventDisableTiming));
    BZ_HIP(hipEventRecord(B->ev_rank, st));
    BZ_HIP(hipStreamWaitEvent(B->st_small, B->ev_rank, 0));
  }
  if (marks) c->tmark("bz:rank");
  // (the long sub-blocks' search as four workgroups, one per chain, when there are four: a 900 k block's search takes 55 ms as one workgroup,
  // and the launch is not over before the last of them is)
  const bool split = option == 2 && c->knob_bz_split != 0 && nlarge > 0;
  if (split) {
    hipLaunchKernelGGL((k_bz_entropy<EN_THREADS, BZ_MAX_SEL, true>), dim3(nlarge, 4), dim3(EN_THREADS), 0, st, E, nsb, 0u);
    hipLaunchKernelGGL(k_bz_pick, dim3(nlarge), dim3(256), 0, st, E);
  } else if (nlarge) hipLaunchKernelGGL((k_bz_entropy<EN_THREADS, BZ_MAX_SEL, false>), dim3(nlarge), dim3(EN_THREADS), 0, st, E, nsb, 0u);
  if (nsb > nlarge) hipLaunchKernelGGL((k_bz_entropy<EN_SMALL_THREADS, (int)EN_SMALL_SEL, false>), dim3(nsb - nlarge), dim3(EN_SMALL_THREADS), 0, sts, E, nsb, nlarge);
  if (both) { BZ_HIP(hipEventRecord(B->ev_small, B->st_small)); BZ_HIP(hipStreamWaitEvent(st, B->ev_small, 0)); }
  hipLaunchKernelGGL(k_bz_block_bits, dim3((nsb + 255) / 256), dim3(256), 0, st, T, B->res.as<uint32_t>());
  if (marks) c->tmark("bz:entropy");
  B->h_res.resize(8ull * nsb); B->h_crc.resize(nsb);
  BZ_HIP(hipMemcpyAsync(B->h_res.data(), B->res.p, 32ull * nsb, hipMemcpyDeviceToHost, st));
  BZ_HIP(hipMemcpyAsync(B->h_crc.data(), B->crc.p, 4ull * nsb, hipMemcpyDeviceToHost, st));
  BZ_HIP(hipStreamSynchronize(st));
  B->h_woff.resize(nsb + 1);
  uint64_t w = 0;
  for (uint32_t s = 0; s < nsb; s++) { B->h_woff[s] = (uint32_t)w; w += (B->h_res[8ull * s + 7] + 31) / 32 + 1; }
  if (w >= (1ull << 32)) { c->err = "bzip2: batch output too large"; return ZADA_E_TOO_LARGE; }
  B->h_woff[nsb] = (uint32_t)w; B->nwords = w;
  if ((rc = dbuf_ensure(c, B->words, 4 * (w + 16)))) return rc;
  BZ_HIP(hipMemcpyAsync(B->woff.p, B->h_woff.data(), 4ull * (nsb + 1), hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemsetAsync(B->words.p, 0, 4 * (w + 16), st));
  hipLaunchKernelGGL(k_bz_emit_head, dim3(nsb), dim3(64), 0, st, T, E, B->woff.as<uint32_t>(), B->words.as<uint32_t>());
  hipLaunchKernelGGL(k_bz_emit_data, dim3(nsb), dim3(EN_THREADS), 0, st, T, E, B->woff.as<uint32_t>(), B->words.as<uint32_t>());
  if (marks) c->tmark("bz:emit");
  BZ_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
//  The stream level.  Encoding and choosing are separate: every unique sub-block of every block is encoded first (its bits
//  are kept on the device), then the tactics are chosen along the stream (the choice of a block depends on the bit phase
//  the blocks before it leave, :1312-1318), then the chosen pieces are copied to their places.  That order is what lets
//  several GPUs take block ranges of one stream (zada_bz2_range_*): sizes are exchanged, the choice is replayed everywhere.
// ---------------------------------------------------------------------------------------------------------------
using KeptSub = Bz2State::KeptSub;
using BlkPlan = Bz2State::BlkPlan;
static void bz_reset_call(Ctx *c) {
  Bz2State *B = bz_state(c);
  for (DBuf &b : B->kept_bufs) if (b.p) hipFree(b.p);
  B->kept_bufs.clear(); B->kept.clear(); B->plans.clear();       // (the trace of the last call stays)
  B->min_bits_sum = 0;
}

// Block limits (:1161-1209, :1406-1423) of the stretch [pos0, pos0 + span_len) of the stream; d_in is indexed by stream position.
// ends_stream: the stretch ends where the stream ends; otherwise a block that reaches the stretch's end was cut by it and is
// left out.  Blocks that start at or after own_end are left out as well.  *next = where the first block left out starts.
static int bz_span_blocks(Ctx *c, int option, const uint8_t *d_in, uint64_t pos0, uint64_t span_len, bool ends_stream, int64_t size_hint, uint64_t own_end,
                          std::vector<uint64_t> &bstart, std::vector<uint32_t> &blen, uint64_t *next) {
  Bz2State *B = bz_state(c);
  hipStream_t st = c->stream;
  int rc;
  const int level = option == 0 ? 1 : option == 1 ? 4 : 9;
  const int32_t block_capacity = 100000 * level;
  const float fc = (float)block_capacity, f_lo = fc * 1.05f, f_hi = fc * 1.30f;        // :1406-1414
  const uint32_t cap_blocks = (uint32_t)(span_len / ((uint64_t)block_capacity * 3 / 8) + 8);
  if ((rc = dbuf_ensure(c, B->rs1, 4 * (span_len + 16))) || (rc = dbuf_ensure(c, B->epre, 4 * (span_len + 16))) || (rc = dbuf_ensure(c, B->agg, 4 * ((span_len + 16) / SC_TILE + 16))) ||
      (rc = dbuf_ensure(c, B->bstart, 8ull * cap_blocks)) || (rc = dbuf_ensure(c, B->blen, 4ull * cap_blocks)) || (rc = dbuf_ensure(c, B->scal, 64))) return rc;
  uint32_t *rs1 = B->rs1.as<uint32_t>(), *E = B->epre.as<uint32_t>(), *agg = B->agg.as<uint32_t>();
  scan_launch<OpMax, true>(st, FRunStart1{d_in + pos0}, span_len, agg, rs1, nullptr);
  scan_launch<OpSum, false>(st, FPieceEnd{d_in + pos0, rs1, span_len}, span_len + 1, agg, E, nullptr);
  uint32_t *d_count = B->scal.as<uint32_t>() + 8;
  hipLaunchKernelGGL(k_bz_acquire, dim3(1), dim3(64), 0, st, rs1, E, span_len, size_hint < 0 ? size_hint : size_hint - (int64_t)pos0, block_capacity, f_lo, f_hi,
                     B->bstart.as<uint64_t>(), B->blen.as<uint32_t>(), cap_blocks, d_count);
  c->tmark("bz:acquire");
  uint32_t nblk = 0;
  BZ_HIP(hipMemcpyAsync(&nblk, d_count, 4, hipMemcpyDeviceToHost, st));
  BZ_HIP(hipStreamSynchronize(st));
  if (nblk > cap_blocks) { c->err = "bzip2: block table overflow"; return ZADA_E_HIP; }
  bstart.resize(nblk); blen.resize(nblk);
  BZ_HIP(hipMemcpy(bstart.data(), B->bstart.p, 8ull * nblk, hipMemcpyDeviceToHost));
  BZ_HIP(hipMemcpy(blen.data(), B->blen.p, 4ull * nblk, hipMemcpyDeviceToHost));
  if (!ends_stream) while (nblk > 0 && bstart[nblk - 1] + blen[nblk - 1] >= span_len) nblk--;
  for (uint32_t k = 0; k < nblk; k++) bstart[k] += pos0;                              // stream positions from here on
  uint32_t keep = nblk;
  while (keep > 0 && bstart[keep - 1] >= own_end) keep--;
  *next = keep < nblk ? bstart[keep] : (nblk ? bstart[nblk - 1] + blen[nblk - 1] : pos0);
  bstart.resize(keep); blen.resize(keep);
  return 0;
}

// Segmentation and every stage of Encode_Block for the given blocks: appends to B->plans / B->kept.
static int bz_blocks_encode_once(Ctx *c, int option, const uint8_t *d_in, const std::vector<uint64_t> &bstart, const std::vector<uint32_t> &blen,
                                 zada_feedback_fn fb, void *user, double prog0, double prog1, bool pipelined, uint64_t batch_elems) {
  Bz2State *B = bz_state(c);
  hipStream_t st = c->stream, st2 = c->stream2;
  int rc;
  const uint32_t nblk = (uint32_t)bstart.size();
  if (nblk == 0) return 0;
  if ((rc = dbuf_ensure(c, B->bstart, 8ull * nblk)) || (rc = dbuf_ensure(c, B->blen, 4ull * nblk))) return rc;
  BZ_HIP(hipMemcpy(B->bstart.p, bstart.data(), 8ull * nblk, hipMemcpyHostToDevice));
  BZ_HIP(hipMemcpy(B->blen.p, blen.data(), 4ull * nblk, hipMemcpyHostToDevice));
  // ---- segmentation (block_900k only): one wave per block and a serial chain of additions, i.e. next to no load for the GPU
  //      but 0.16 s of latency -- it runs on the second stream while the pieces that do not depend on it are encoded ----
  std::vector<uint32_t> seg_off(2ull * nblk + 1, 0), seg, seg_cnt(2ull * nblk, 0);
  uint64_t seg_total = 0;
  if (option == 2) {
    if (!B->etab_ready) {
      std::vector<double> et(SEG_WINDOW + 2);
      const double inv = 1.0 / (double)SEG_WINDOW;
      et[0] = 0.0;
      for (int f = 1; f <= SEG_WINDOW + 1; f++) { const double pr = (double)f * inv; et[f] = -(pr * log(pr)); }
      if ((rc = dbuf_ensure(c, B->etab, 8ull * et.size()))) return rc;
      BZ_HIP(hipMemcpy(B->etab.p, et.data(), 8ull * et.size(), hipMemcpyHostToDevice));
      B->etab_ready = true;
    }
    uint64_t so = 0;
    for (uint32_t k = 0; k < nblk; k++) { seg_off[2 * k] = (uint32_t)so; so += blen[k] / 4000 + 2; seg_off[2 * k + 1] = (uint32_t)so; so += blen[k] / 8000 + 2; }
    seg_off[2ull * nblk] = (uint32_t)so;
    seg_total = so;
    if ((rc = dbuf_ensure(c, B->seg_off, 4ull * (2ull * nblk + 1))) || (rc = dbuf_ensure(c, B->seg, 4 * so + 16)) || (rc = dbuf_ensure(c, B->seg_cnt, 8ull * nblk + 16))) return rc;
    BZ_HIP(hipMemcpy(B->seg_off.p, seg_off.data(), 4ull * (2ull * nblk + 1), hipMemcpyHostToDevice));
    BZ_HIP(hipStreamSynchronize(st));                                    // the input is in place
    hipLaunchKernelGGL(k_bz_segment, dim3(nblk), dim3(64), 0, st2, d_in, B->bstart.as<uint64_t>(), B->blen.as<uint32_t>(), nblk, B->etab.as<double>(),
                       (double)0.6f, (double)0.4f, B->seg_off.as<uint32_t>(), B->seg.as<uint32_t>(), B->seg_cnt.as<uint32_t>());
  }
  const size_t plan_base = B->plans.size();
  constexpr uint32_t KEPT = 0x80000000u;                                   // piece number that already is a number in B->kept
  // Two batches in flight ("bz_pipeline", default on): the second one in a state of its own (B->slot1).
  if (pipelined) {
    if (!B->slot1) B->slot1 = new Bz2State();
    if (!B->st_pipe) {
      // (round 5, knob "bz_pipe_prio": the kernel timeline (tests/prof_trace_bz2.sh) shows the next batch's transform kernels -- 1 024-thread
      // workgroups -- waiting for the whole of the entropy search of the batch before, whose 512-thread workgroups hold every CU for tens of
      // milliseconds: one launch of k_bz_filter_emit takes 78 ms beside a 79 ms k_bz_entropy.  With the worker's stream at the LOWEST priority a slot
      // that comes free should go to the transforms first -- measured 467 / 460 ms against 453 without: no gain, the default stays 0)
      int least = 0, greatest = 0;
      BZ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      BZ_HIP(hipStreamCreateWithPriority(&B->st_pipe, hipStreamNonBlocking, c->knob_bz_pipe_prio ? least : 0));
    }
  }
  // (err: the worker's error text -- hip_check writes it there through tls_err while the main thread may be writing c->err -- is copied to
  // c->err by finish, on the main thread)
  struct Pending { bool active = false; std::thread th; int rc = 0; Bz2State *S = nullptr; uint32_t base = 0; DBuf kb; std::string err; } pend;
  // (no return leaves a worker behind, nor the device buffer of a batch that was never booked)
  struct Joiner { Pending &p; ~Joiner() { if (p.th.joinable()) p.th.join(); if (p.active && p.kb.p) { hipFree(p.kb.p); p.kb.p = nullptr; } } } joiner{pend};
  uint32_t nbatch = 0;
  // waits for the batch whose stage B is under way and books its bit strings: the plans' piece numbers become numbers in B->kept
  auto finish = [&]() -> int {
    if (!pend.active) return 0;
    if (pend.th.joinable()) pend.th.join();
    pend.active = false;
    if (pend.rc) { if (pend.kb.p) { hipFree(pend.kb.p); pend.kb.p = nullptr; } if (!pend.err.empty()) c->err = pend.err; return pend.rc; }
    Bz2State *S = pend.S;
    const uint32_t bufno = (uint32_t)B->kept_bufs.size();
    B->kept_bufs.push_back(pend.kb);
    for (uint32_t s = 0; s < S->nsb; s++) B->kept[pend.base + s] = {S->h_res[8ull * s + 7], S->h_woff[s], S->h_crc[s], bufno};
    return 0;
  };
  // A batch's pieces get their numbers in B->kept when the batch is launched (what the numbers stand for -- bits, place, CRC -- arrives
  // with finish): the next batch, also the first of the segments' pass, is planned without waiting for this one's entropy stage.
  auto book = [&](uint32_t k0, uint32_t k1, int pass, const std::vector<uint64_t> &starts, const std::vector<uint32_t> &lens) -> uint32_t {
    const uint32_t base = (uint32_t)B->kept.size();
    B->kept.resize((size_t)base + starts.size());
    for (uint32_t k = k0; k < k1; k++) {
      BlkPlan &P = B->plans[plan_base + k];
      for (int t = 2 * pass; t < 2 * pass + 2; t++)
        for (uint32_t &sb : P.tac[t]) {
          if (sb & KEPT) { sb &= ~KEPT; continue; }
          sb += base;
          bool known = false;
          for (const auto &pc : P.pieces) known |= pc.kept == sb;
          if (!known) P.pieces.push_back({starts[sb - base], lens[sb - base], sb});
        }
    }
    return base;
  };
  // pass 0: the single block and its four quarters; pass 1: the segments (those that are not one of the former)
  for (int pass = 0; pass < (option == 2 ? 2 : 1); pass++) {
    if (pass == 1) {
      BZ_HIP(hipStreamSynchronize(st2));
      c->tmark("bz:segment");                                              // what of it was not hidden
      seg.resize(seg_total);
      BZ_HIP(hipMemcpy(seg.data(), B->seg.p, 4 * seg_total, hipMemcpyDeviceToHost));
      BZ_HIP(hipMemcpy(seg_cnt.data(), B->seg_cnt.p, 8ull * nblk, hipMemcpyDeviceToHost));
    }
    // Batches of this pass.  The entropy stage of the LAST batch of a call has nothing to hide behind: when batches are pipelined the
    // last pass can end with a short batch ("bz_tail_pct" of the pass; measured: no gain, what the tail saves the extra batch costs -- default 0), the rest is cut evenly.
    std::vector<uint64_t> targets;
    {
      uint64_t pass_est = 0;
      for (uint32_t k = 0; k < nblk; k++) pass_est += 2 * ((uint64_t)blen[k] + blen[k] / 4 + 8) + 24;       // (single + quarters, or the two segmentations)
      const bool last_pass = pass == (option == 2 ? 1 : 0);
      const uint64_t tail = pipelined && last_pass && nblk >= 8 ? pass_est * (uint64_t)(c->knob_bz_tail_pct < 0 ? 0 : c->knob_bz_tail_pct > 50 ? 50 : c->knob_bz_tail_pct) / 100 : 0;
      const uint64_t body = pass_est - tail;
      const uint64_t nb = (body + batch_elems - 1) / batch_elems;
      for (uint64_t i = 0; i < nb; i++) targets.push_back(body / nb + body / (64 * nb) + 1);
      if (tail) targets.push_back(batch_elems);                                                              // (the rest)
    }
    size_t bi = 0;
    uint32_t k0 = 0;
    while (k0 < nblk) {
      std::vector<uint64_t> starts; std::vector<uint32_t> lens;
      uint64_t est = 0;
      uint32_t k1 = k0;
      const uint64_t limit_cur = bi < targets.size() && targets[bi] < batch_elems ? targets[bi] : batch_elems;
      bi++;
      for (; k1 < nblk; k1++) {
        if (pass == 0) { BlkPlan Pn; Pn.start = bstart[k1]; Pn.len = blen[k1]; B->plans.push_back(std::move(Pn)); }
        BlkPlan &P = B->plans[plan_base + k1];
        const uint64_t bs = bstart[k1]; const uint32_t bl = blen[k1];
        const size_t first_sub = starts.size();
        auto sub_of = [&](uint64_t s0, uint32_t l0) -> uint32_t {                       // pieces that several tactics share are encoded once
          for (const auto &pc : P.pieces) if (pc.start == s0 && pc.len == l0) return pc.kept | KEPT;
          for (size_t i = first_sub; i < starts.size(); i++) if (starts[i] == s0 && lens[i] == l0) return (uint32_t)i;
          starts.push_back(s0); lens.push_back(l0);
          return (uint32_t)(starts.size() - 1);
        };
        uint64_t e_blk = 0;
        const size_t before = starts.size();
        std::vector<uint32_t> t0, t1;
        if (pass == 0) {
          t0.push_back(sub_of(bs, bl));                                                  // single
          if (option == 2) {
            const uint32_t size = bl / 4;                                                // parts_4 :1237-1253
            uint32_t stop = 0;
            for (uint32_t count = 1; count <= 4; count++) { const uint32_t start = stop + 1; stop = count == 4 ? bl : count * size; t1.push_back(sub_of(bs + start - 1, stop - start + 1)); }
          }
        } else {
          for (int t = 0; t < 2; t++) {                                                  // segmented_1 / _2 :1265-1297
            std::vector<uint32_t> &tv = t ? t1 : t0;
            const uint32_t cnt = seg_cnt[2 * k1 + t], *sp = seg.data() + seg_off[2 * k1 + t];
            if (cnt == 0) tv.push_back(sub_of(bs, 0));
            uint32_t index_start = 1;
            for (uint32_t q = 0; q < cnt; q++) { tv.push_back(sub_of(bs + index_start - 1, sp[q] - index_start + 1)); index_start = sp[q] + 1; }
          }
        }
        for (size_t i = before; i < starts.size(); i++) e_blk += (uint64_t)lens[i] + lens[i] / 4 + 8;
        if (k1 > k0 && est + e_blk > limit_cur) { starts.resize(before); lens.resize(before); if (pass == 0) B->plans.pop_back(); break; }
        est += e_blk;
        P.tac[2 * pass] = std::move(t0); P.tac[2 * pass + 1] = std::move(t1);
      }
      if (!starts.empty()) {
        // Stage A of this batch (RLE_1, rotation sort, MTF: bound by memory) on the main stream, next to stage B of the batch
        // before (rankings, entropy search, bits: chains of short dependent steps, bound by how many sub-blocks are in flight).
        Bz2State *S = pipelined && (nbatch & 1) ? B->slot1 : B;
        nbatch++;
        if ((rc = bz_transform(c, S, st, true, d_in, starts, lens)) || (rc = bz_mtf(c, S, st, true))) { finish(); return rc; }
        BZ_HIP(hipStreamSynchronize(st));
        if ((rc = finish())) return rc;                                       // the batch before: its worker has had all of stage A to finish
        pend.S = S; pend.base = book(k0, k1, pass, starts, lens); pend.rc = 0; pend.kb = DBuf();
        auto stage_b = [c, option, &pend](hipStream_t sb, bool marks) {
          Bz2State *S2 = pend.S;
          int r = bz_entropy_emit(c, S2, sb, marks, option);
          if (!r) r = dbuf_ensure(c, pend.kb, 4 * (S2->nwords + 16));
          if (!r && hipMemcpyAsync(pend.kb.p, S2->words.p, 4 * S2->nwords, hipMemcpyDeviceToDevice, sb) != hipSuccess) r = ZADA_E_HIP_;
          if (!r && hipStreamSynchronize(sb) != hipSuccess) r = ZADA_E_HIP_;
          pend.rc = r;
        };
        pend.active = true;
        if (pipelined) pend.th = std::thread([c, stage_b, B, &pend] { hipSetDevice(c->device); tls_err = &pend.err; stage_b(B->st_pipe, false); tls_err = nullptr; });
        else { stage_b(st, true); c->tmark("bz:keep"); if ((rc = finish())) return rc; }
      } else {
        for (uint32_t k = k0; k < k1; k++) for (int t = 2 * pass; t < 2 * pass + 2; t++) for (uint32_t &sb : B->plans[plan_base + k].tac[t]) sb &= ~KEPT;
      }
      k0 = k1;
      const double done = ((double)pass + (double)k0 / (double)nblk) / (option == 2 ? 2.0 : 1.0);
      if (fb && fb((int)(prog0 + (prog1 - prog0) * done), user)) { finish(); hipStreamSynchronize(st2); return ZADA_ABORTED; }
    }
  }
  if ((rc = finish())) return rc;
  for (size_t q = plan_base; q < B->plans.size(); q++) {
    uint64_t mn = ~0ull;
    for (int t = 0; t < 4; t++) {
      uint64_t bits = 0;
      for (uint32_t sb : B->plans[q].tac[t]) bits += B->kept[sb].bits;
      if (!B->plans[q].tac[t].empty() && bits < mn) mn = bits;
    }
    B->min_bits_sum += mn;
  }
  return 0;
}
// Encode_Block for every piece of every tactic of the blocks.  Out of device memory (two batches in flight hold two full states, sized for
// "bz_batch_melems" elements each): what the failed attempt booked is dropped, and the call goes again -- first without the second batch
// in flight (its state is freed; the context stays unpipelined), then with batches half the size, down to 32 Mi elements.
static int bz_blocks_encode(Ctx *c, int option, const uint8_t *d_in, const std::vector<uint64_t> &bstart, const std::vector<uint32_t> &blen,
                            zada_feedback_fn fb, void *user, double prog0, double prog1) {
  Bz2State *B = bz_state(c);
  const size_t plans0 = B->plans.size(), kept0 = B->kept.size(), bufs0 = B->kept_bufs.size(), trace0 = B->trace.size();
  const uint64_t bits0 = B->min_bits_sum;
  bool pipelined = c->knob_bz_pipeline != 0 && !B->no_pipeline;
  uint64_t melems = (uint64_t)(c->knob_bz_batch_melems > 0 ? c->knob_bz_batch_melems : 640);
  for (;;) {
    const int rc = bz_blocks_encode_once(c, option, d_in, bstart, blen, fb, user, prog0, prog1, pipelined, melems << 20);
    if (rc != ZADA_E_NOMEM) return rc;
    hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2);
    for (size_t i = bufs0; i < B->kept_bufs.size(); i++) if (B->kept_bufs[i].p) hipFree(B->kept_bufs[i].p);
    B->kept_bufs.resize(bufs0); B->kept.resize(kept0); B->plans.resize(plans0); B->trace.resize(trace0); B->min_bits_sum = bits0;
    if (pipelined) {
      if (B->slot1) { bz_free_state(B->slot1); B->slot1 = nullptr; }
      pipelined = false; B->no_pipeline = true;
    } else if (melems > 32) melems /= 2;
    else return rc;
  }
}

// per block and tactic: bits, number of pieces, and the pieces' CRCs folded from zero (the combined CRC after the block is
// rotl(before, pieces) xor that value, :1023-1026); bits = all ones for a tactic the block does not have
static void bz_fill_table(Bz2State *B, std::vector<uint64_t> &tab) {
  tab.assign(12 * B->plans.size(), 0);
  for (size_t q = 0; q < B->plans.size(); q++)
    for (int t = 0; t < 4; t++) {
      const std::vector<uint32_t> &v = B->plans[q].tac[t];
      uint64_t bits = 0; uint32_t fold = 0;
      for (uint32_t sb : v) { bits += B->kept[sb].bits; fold = ((fold << 1) | (fold >> 31)) ^ B->kept[sb].crc; }
      tab[12 * q + 3 * t] = v.empty() ? ~0ull : bits; tab[12 * q + 3 * t + 1] = v.size(); tab[12 * q + 3 * t + 2] = fold;
    }
}

// the chosen pieces of the plans, bit-shifted to their places: the range's bytes from stream byte bit_begin / 8 on
static int bz_assemble_range(Ctx *c, int option, const uint8_t *choice, uint64_t bit_begin, int flags, uint32_t footer_crc, uint8_t *d_out, uint64_t cap, uint64_t *nbytes_out) {
  Bz2State *B = bz_state(c);
  hipStream_t st = c->stream;
  int rc;
  const int level = option == 0 ? 1 : option == 1 ? 4 : 9;
  const uint64_t base_bits = (flags & 1) ? 0 : (bit_begin & ~7ull);
  uint64_t bitpos = bit_begin - base_bits, total = bitpos;
  for (size_t q = 0; q < B->plans.size(); q++) for (uint32_t sb : B->plans[q].tac[choice[q]]) total += B->kept[sb].bits;
  if (flags & 2) total += 80;
  const uint64_t nbytes = (total + 7) / 8, nw = (nbytes + 3) / 4;
  if (nbytes_out) *nbytes_out = nbytes;
  if (nbytes > cap) { c->err = "output buffer too small"; return ZADA_E_INVALID; }
  if ((rc = dbuf_ensure(c, B->outw, 4 * nw + 64)) || (rc = dbuf_ensure(c, B->extra, 64))) return rc;
  BZ_HIP(hipMemsetAsync(B->outw.p, 0, 4 * nw + 64, st));
  if (flags & 1) {
    const uint32_t head = ((uint32_t)'B' << 24) | ((uint32_t)'Z' << 16) | ((uint32_t)'h' << 8) | (uint32_t)('0' + level);   // :1380-1387
    BZ_HIP(hipMemcpyAsync(B->outw.p, &head, 4, hipMemcpyHostToDevice, st));
  }
  std::vector<std::vector<CopyJob>> jobs(B->kept_bufs.size());
  std::vector<std::vector<uint64_t>> first(B->kept_bufs.size());
  std::vector<uint64_t> dstw(B->kept_bufs.size(), 0);
  for (size_t q = 0; q < B->plans.size(); q++)
    for (uint32_t sb : B->plans[q].tac[choice[q]]) {
      const KeptSub &K = B->kept[sb];
      CopyJob J; J.src_word = K.woff; J.dpos = bitpos; J.bits = K.bits; J.first_dst_word = bitpos >> 5;
      first[K.buf].push_back(dstw[K.buf]);
      dstw[K.buf] += ((bitpos + K.bits + 31) >> 5) - (bitpos >> 5);
      jobs[K.buf].push_back(J);
      bitpos += K.bits;
    }
  for (size_t b = 0; b < jobs.size(); b++) {
    if (jobs[b].empty()) continue;
    first[b].push_back(dstw[b]);
    if ((rc = dbuf_ensure(c, B->jobs, sizeof(CopyJob) * jobs[b].size())) || (rc = dbuf_ensure(c, B->job_first, 8 * first[b].size()))) return rc;
    BZ_HIP(hipMemcpyAsync(B->jobs.p, jobs[b].data(), sizeof(CopyJob) * jobs[b].size(), hipMemcpyHostToDevice, st));
    BZ_HIP(hipMemcpyAsync(B->job_first.p, first[b].data(), 8 * first[b].size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bz_assemble, dim3((uint32_t)((dstw[b] + 255) / 256)), dim3(256), 0, st, B->jobs.as<CopyJob>(), B->job_first.as<uint64_t>(), (uint32_t)jobs[b].size(),
                       B->kept_bufs[b].as<uint32_t>(), B->outw.as<uint32_t>());
    BZ_HIP(hipStreamSynchronize(st));          // the job tables are reused by the next buffer
  }
  if (flags & 2) {      // Write_Stream_Footer :1391-1403
    uint32_t foot[4] = {0x17724538u, 0x50900000u | (footer_crc >> 16), footer_crc << 16, 0};
    BZ_HIP(hipMemcpyAsync(B->extra.p, foot, 16, hipMemcpyHostToDevice, st));
    CopyJob J; J.src_word = 0; J.dpos = bitpos; J.bits = 80; J.first_dst_word = bitpos >> 5;
    uint64_t jf[2] = {0, ((bitpos + 80 + 31) >> 5) - (bitpos >> 5)};
    if ((rc = dbuf_ensure(c, B->jobs, sizeof(CopyJob))) || (rc = dbuf_ensure(c, B->job_first, 16))) return rc;
    BZ_HIP(hipMemcpyAsync(B->jobs.p, &J, sizeof J, hipMemcpyHostToDevice, st));
    BZ_HIP(hipMemcpyAsync(B->job_first.p, jf, 16, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bz_assemble, dim3(1), dim3(256), 0, st, B->jobs.as<CopyJob>(), B->job_first.as<uint64_t>(), 1u, B->extra.as<uint32_t>(), B->outw.as<uint32_t>());
  }
  c->tmark("bz:assemble");
  if (((uintptr_t)d_out & 3) != 0 || nw * 4 > cap) {      // swap in place, then copy the bytes
    hipLaunchKernelGGL(k_bz_words_to_bytes, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, st, B->outw.as<uint32_t>(), nw, B->outw.as<uint32_t>());
    BZ_HIP(hipMemcpyAsync(d_out, B->outw.p, nbytes, hipMemcpyDeviceToDevice, st));
  } else hipLaunchKernelGGL(k_bz_words_to_bytes, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, st, B->outw.as<uint32_t>(), nw, (uint32_t *)d_out);
  BZ_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // namespace zada

// The tactic of every block along the stream (:1312-1318) and the combined CRC (:1023-1026).  tab: 12 values per block, see
// bz_fill_table.  Pure host arithmetic: every rank of a multi-GPU run replays it on the gathered tables.
extern "C" void zada_bz2_select(uint64_t nblk, const uint64_t *tab, uint64_t bitpos_in, uint32_t crc_in, uint8_t *choice, uint64_t *bitpos_out, uint32_t *crc_out) {
  uint64_t bitpos = bitpos_in;
  uint32_t crc = crc_in;
  for (uint64_t q = 0; q < nblk; q++) {
    const uint64_t phase = bitpos & 7, *t = tab + 12 * q;
    int best = 0; uint64_t best_idx = 0;
    for (int k = 0; k < 4; k++) {
      if (t[3 * k] == ~0ull) continue;
      const uint64_t idx = (phase + t[3 * k]) / 8;                                   // destination_index: whole bytes written
      if (k == 0 || idx < best_idx) { best = k; best_idx = idx; }
    }
    if (choice) choice[q] = (uint8_t)best;
    const uint32_t rot = (uint32_t)(t[3 * best + 1] & 31u);
    crc = (rot ? ((crc << rot) | (crc >> (32 - rot))) : crc) ^ (uint32_t)t[3 * best + 2];
    bitpos += t[3 * best];
  }
  if (bitpos_out) *bitpos_out = bitpos;
  if (crc_out) *crc_out = crc;
}

namespace zada {

static void bz_set_trace(Bz2State *B, const uint8_t *choice) {
  B->trace.clear();
  for (size_t q = 0; q < B->plans.size(); q++) {
    B->trace.push_back(B->plans[q].start); B->trace.push_back(B->plans[q].len); B->trace.push_back(choice[q]); B->trace.push_back(B->plans[q].tac[choice[q]].size());
  }
}

// BZip2.Encoding.Encode (:87-1431) of n bytes at d_in; the stream goes to d_out (cap bytes).  Returns ZADA_OK, ZADA_INEFFICIENT
// when the stream is not smaller than the input (it is still delivered if it fits), ZADA_ABORTED, or an error.
int bz2_encode_device(Ctx *c, int option, const uint8_t *d_in, uint64_t n, int64_t size_hint, uint8_t *d_out, uint64_t cap, uint64_t *out_len,
                      zada_feedback_fn fb, void *user) {
  Bz2State *B = bz_state(c);
  int rc;
  if (cap < 64) { c->err = "output buffer too small"; return ZADA_E_INVALID; }
  bz_reset_call(c);
  B->trace.clear();
  if (fb && fb(0, user)) return ZADA_ABORTED;
  const uint64_t span_max = (uint64_t)(c->knob_bz_span_mib > 0 ? c->knob_bz_span_mib : 1024) << 20;
  // ---- spans: the block chain is walked a stretch of the stream at a time (32-bit scans); a span starts where a block starts ----
  uint64_t pos0 = 0;
  do {
    const uint64_t span_len = n - pos0 < span_max ? n - pos0 : span_max;
    std::vector<uint64_t> bstart; std::vector<uint32_t> blen;
    uint64_t next = pos0;
    if ((rc = bz_span_blocks(c, option, d_in, pos0, span_len, pos0 + span_len == n, size_hint, ~0ull, bstart, blen, &next))) return rc;
    if (bstart.empty()) { c->err = "bzip2: span shorter than a block"; return ZADA_E_INVALID; }
    if ((rc = bz_blocks_encode(c, option, d_in, bstart, blen, fb, user, 3.0 + 95.0 * (double)pos0 / (double)(n ? n : 1), 3.0 + 95.0 * (double)next / (double)(n ? n : 1)))) return rc;
    pos0 = next;
    if (B->min_bits_sum + 32 + 80 > cap * 8) {      // cannot fit any more, whatever is chosen
      const uint64_t least = (B->min_bits_sum + 32 + 80 + 7) / 8;
      if (out_len) *out_len = least;
      bz_reset_call(c);
      if (least >= n) return ZADA_INEFFICIENT;      // Compression_inefficient (zip-compress.adb:479-486): not smaller than the input
      c->err = "output buffer too small"; return ZADA_E_INVALID;
    }
  } while (pos0 < n);
  std::vector<uint64_t> tab;
  bz_fill_table(B, tab);
  std::vector<uint8_t> choice(B->plans.size());
  uint64_t bit_end = 0; uint32_t crc = 0;
  zada_bz2_select(B->plans.size(), tab.data(), 32, 0, choice.data(), &bit_end, &crc);
  bz_set_trace(B, choice.data());
  uint64_t nbytes = (bit_end + 80 + 7) / 8;
  if (out_len) *out_len = nbytes;
  if (nbytes > cap) {
    bz_reset_call(c);
    if (nbytes >= n) return ZADA_INEFFICIENT;
    c->err = "output buffer too small"; return ZADA_E_INVALID;
  }
  rc = bz_assemble_range(c, option, choice.data(), 32, 3, crc, d_out, cap, &nbytes);
  bz_reset_call(c);
  if (rc) return rc;
  if (fb && fb(100, user)) return ZADA_ABORTED;
  return nbytes >= n ? ZADA_INEFFICIENT : ZADA_OK;
}

// ---- many small entries (zipada's usual workload, tools/zipada.adb:126-134), one launch sequence: every entry is a stream of
//      its own with ONE block (the caller only sends entries short enough for that), so the entries are simply the blocks of a
//      call; each gets its own header, choice (from bit 32) and footer.  h_out: host buffer for the streams, entry q's at
//      out_off[q] (4-byte aligned), out_bytes[q] long.  Returns ZADA_E_TOO_LARGE if h_out (cap bytes) cannot take them.
int bz2_batch_encode(Ctx *c, int option, const uint8_t *d_arena, uint32_t E, const uint64_t *starts, const uint32_t *lens, uint8_t *h_out, uint64_t cap,
                     uint64_t *out_off, uint64_t *out_bytes) {
  Bz2State *B = bz_state(c);
  hipStream_t st = c->stream;
  int rc;
  bz_reset_call(c);
  B->trace.clear();
  std::vector<uint64_t> bstart(starts, starts + E);
  std::vector<uint32_t> blen(lens, lens + E);
  if ((rc = bz_blocks_encode(c, option, d_arena, bstart, blen, nullptr, nullptr, 0, 0))) return rc;
  std::vector<uint64_t> tab;
  bz_fill_table(B, tab);
  const int level = option == 0 ? 1 : option == 1 ? 4 : 9;
  std::vector<uint8_t> choice(E);
  std::vector<uint32_t> extra(4ull * E);
  std::vector<uint64_t> woff(E + 1);
  uint64_t w = 0;
  for (uint32_t q = 0; q < E; q++) {
    uint64_t bit_end = 0; uint32_t crc = 0;
    zada_bz2_select(1, tab.data() + 12ull * q, 32, 0, &choice[q], &bit_end, &crc);
    extra[4ull * q] = ((uint32_t)'B' << 24) | ((uint32_t)'Z' << 16) | ((uint32_t)'h' << 8) | (uint32_t)('0' + level);
    extra[4ull * q + 1] = 0x17724538u; extra[4ull * q + 2] = 0x50900000u | (crc >> 16); extra[4ull * q + 3] = crc << 16;
    woff[q] = w;
    out_off[q] = 4 * w; out_bytes[q] = (bit_end + 80 + 7) / 8;
    w += (bit_end + 80 + 31) / 32 + 1;
  }
  woff[E] = w;
  bz_set_trace(B, choice.data());
  if (4 * w > cap) { bz_reset_call(c); return ZADA_E_TOO_LARGE; }
  if ((rc = dbuf_ensure(c, B->outw, 4 * w + 64)) || (rc = dbuf_ensure(c, B->extra, 16ull * E + 64))) return rc;
  BZ_HIP(hipMemsetAsync(B->outw.p, 0, 4 * w + 64, st));
  BZ_HIP(hipMemcpyAsync(B->extra.p, extra.data(), 16ull * E, hipMemcpyHostToDevice, st));
  // copy jobs: per source buffer (the kept bit strings of every batch of pieces, and the headers / footers)
  const size_t nsrc = B->kept_bufs.size() + 1;
  std::vector<std::vector<CopyJob>> jobs(nsrc);
  std::vector<std::vector<uint64_t>> first(nsrc);
  std::vector<uint64_t> dstw(nsrc, 0);
  auto add = [&](size_t src, uint64_t src_word, uint64_t dpos, uint64_t bits) {
    CopyJob J; J.src_word = src_word; J.dpos = dpos; J.bits = bits; J.first_dst_word = dpos >> 5;
    first[src].push_back(dstw[src]);
    dstw[src] += ((dpos + bits + 31) >> 5) - (dpos >> 5);
    jobs[src].push_back(J);
  };
  for (uint32_t q = 0; q < E; q++) {
    uint64_t bitpos = 32 * woff[q];
    add(nsrc - 1, 4ull * q, bitpos, 32);
    bitpos += 32;
    for (uint32_t sb : B->plans[q].tac[choice[q]]) { const KeptSub &K = B->kept[sb]; add(K.buf, K.woff, bitpos, K.bits); bitpos += K.bits; }
    add(nsrc - 1, 4ull * q + 1, bitpos, 80);
  }
  for (size_t b = 0; b < nsrc; b++) {
    if (jobs[b].empty()) continue;
    first[b].push_back(dstw[b]);
    if ((rc = dbuf_ensure(c, B->jobs, sizeof(CopyJob) * jobs[b].size())) || (rc = dbuf_ensure(c, B->job_first, 8 * first[b].size()))) return rc;
    BZ_HIP(hipMemcpyAsync(B->jobs.p, jobs[b].data(), sizeof(CopyJob) * jobs[b].size(), hipMemcpyHostToDevice, st));
    BZ_HIP(hipMemcpyAsync(B->job_first.p, first[b].data(), 8 * first[b].size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bz_assemble, dim3((uint32_t)((dstw[b] + 255) / 256)), dim3(256), 0, st, B->jobs.as<CopyJob>(), B->job_first.as<uint64_t>(), (uint32_t)jobs[b].size(),
                       b + 1 == nsrc ? B->extra.as<uint32_t>() : B->kept_bufs[b].as<uint32_t>(), B->outw.as<uint32_t>());
    BZ_HIP(hipStreamSynchronize(st));
  }
  c->tmark("bz:assemble");
  hipLaunchKernelGGL(k_bz_words_to_bytes, dim3((uint32_t)((w + 255) / 256)), dim3(256), 0, st, B->outw.as<uint32_t>(), w, B->outw.as<uint32_t>());
  BZ_HIP(hipMemcpyAsync(h_out, B->outw.p, 4 * w, hipMemcpyDeviceToHost, st));
  BZ_HIP(hipStreamSynchronize(st));
  bz_reset_call(c);
  return 0;
}

// ---- one stream over several contexts / GPUs: a context takes the blocks that start inside its range ----
int bz2_range_open(Ctx *c, int option, const uint8_t *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t stream_total, uint64_t start, uint64_t own_end,
                   uint64_t *next_start, uint64_t *nblocks) {
  Bz2State *B = bz_state(c);
  int rc;
  bz_reset_call(c);
  B->trace.clear();
  B->rg_bstart.clear(); B->rg_blen.clear();
  B->rg_option = option; B->rg_in = d_buf - buf_off;
  const uint64_t buf_end = buf_off + buf_len, span_max = (uint64_t)(c->knob_bz_span_mib > 0 ? c->knob_bz_span_mib : 1024) << 20;
  uint64_t pos0 = start;
  if (start < buf_off || start > buf_end || own_end > stream_total) { c->err = "bzip2 range: start outside the buffer"; return ZADA_E_INVALID; }
  while (pos0 < own_end || (stream_total == 0 && B->rg_bstart.empty())) {
    const uint64_t span_len = buf_end - pos0 < span_max ? buf_end - pos0 : span_max;
    std::vector<uint64_t> bstart; std::vector<uint32_t> blen;
    uint64_t next = pos0;
    if ((rc = bz_span_blocks(c, option, B->rg_in, pos0, span_len, pos0 + span_len == stream_total, (int64_t)stream_total, own_end, bstart, blen, &next))) return rc;
    if (bstart.empty()) { c->err = "bzip2 range: the bytes behind the range do not hold its last block"; return ZADA_E_INVALID; }
    B->rg_bstart.insert(B->rg_bstart.end(), bstart.begin(), bstart.end());
    B->rg_blen.insert(B->rg_blen.end(), blen.begin(), blen.end());
    pos0 = next;
    if (stream_total == 0) break;
  }
  if (next_start) *next_start = pos0;
  if (nblocks) *nblocks = B->rg_bstart.size();
  return 0;
}
int bz2_range_encode(Ctx *c) {
  Bz2State *B = bz_state(c);
  return bz_blocks_encode(c, B->rg_option, B->rg_in, B->rg_bstart, B->rg_blen, nullptr, nullptr, 0, 0);
}
uint64_t bz2_range_table(Ctx *c, uint64_t *tab, uint64_t cap_blocks) {
  Bz2State *B = bz_state(c);
  std::vector<uint64_t> t;
  bz_fill_table(B, t);
  if (tab && B->plans.size() <= cap_blocks) memcpy(tab, t.data(), 8 * t.size());
  return B->plans.size();
}
int bz2_range_assemble(Ctx *c, const uint8_t *choice, uint64_t nblk, uint64_t bit_begin, int flags, uint32_t footer_crc, uint8_t *d_out, uint64_t cap, uint64_t *nbytes) {
  Bz2State *B = bz_state(c);
  if (nblk != B->plans.size()) { c->err = "bzip2 range: one choice per block"; return ZADA_E_INVALID; }
  for (uint64_t q = 0; q < nblk; q++) if (choice[q] > 3 || B->plans[q].tac[choice[q]].empty()) { c->err = "bzip2 range: no such tactic"; return ZADA_E_INVALID; }
  bz_set_trace(B, choice);
  return bz_assemble_range(c, B->rg_option, choice, bit_begin, flags, footer_crc, d_out, cap, nbytes);
}

uint64_t bz2_last_blocks(Ctx *c, uint64_t *dst, uint64_t cap_items) {
  Bz2State *B = bz_state(c);
  const uint64_t k = B->trace.size() < cap_items ? B->trace.size() : cap_items;
  if (dst && k) memcpy(dst, B->trace.data(), 8 * k);
  return B->trace.size();
}

}  // namespace zada

struct zada_ctx { zada::Ctx c; };
using namespace zada;

// ---- one stream over several contexts (include/zada.h) ----
extern "C" int zada_bz2_range_open(zada_ctx *z, int method, const void *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t stream_total, uint64_t start, uint64_t own_end,
                                   uint64_t *next_start, uint64_t *nblocks) {
  if (!z || method < ZADA_BZIP2_1 || method > ZADA_BZIP2_3) return ZADA_E_INVALID;
  hipSetDevice(z->c.device);
  z->c.tbegin();
  z->c.tmark("bz:begin");
  return bz2_range_open(&z->c, method - ZADA_BZIP2_1, (const uint8_t *)d_buf, buf_len, buf_off, stream_total, start, own_end, next_start, nblocks);
}
extern "C" int zada_bz2_range_encode(zada_ctx *z) {
  if (!z) return ZADA_E_INVALID;
  hipSetDevice(z->c.device);
  return bz2_range_encode(&z->c);
}
extern "C" uint64_t zada_bz2_range_table(zada_ctx *z, uint64_t *tab, uint64_t cap_blocks) { return z ? bz2_range_table(&z->c, tab, cap_blocks) : 0; }
extern "C" int zada_bz2_range_assemble(zada_ctx *z, const uint8_t *choice, uint64_t nblk, uint64_t bit_begin, int flags, uint32_t footer_crc, void *d_out, uint64_t cap,
                                       uint64_t *nbytes) {
  if (!z) return ZADA_E_INVALID;
  hipSetDevice(z->c.device);
  const int rc = bz2_range_assemble(&z->c, choice, nblk, bit_begin, flags, footer_crc, (uint8_t *)d_out, cap, nbytes);
  z->c.tmark("bz:end");
  z->c.tend();
  return rc;
}

// ---- test hooks: run a list of sub-blocks of a host buffer through the stages, then fetch any table of the state ----
extern "C" int zada_bz2_run(zada_ctx *z, const uint8_t *in, uint64_t n, uint32_t nsb, const uint64_t *starts, const uint32_t *lens, int option, int stages) {
  if (!z || (!in && n) || option < 0 || option > 2) return ZADA_E_INVALID;
  Ctx *c = &z->c;
  hipSetDevice(c->device);
  uint8_t *d_in = nullptr;
  if (hipMalloc(&d_in, n + 64) != hipSuccess) return ZADA_E_NOMEM;
  hipMemcpy(d_in, in, n, hipMemcpyHostToDevice);
  hipMemset(d_in + n, 0, 64);
  std::vector<uint64_t> s(starts, starts + nsb);
  std::vector<uint32_t> l(lens, lens + nsb);
  Bz2State *B = bz_state(c);
  int rc = bz_transform(c, B, c->stream, true, d_in, s, l);
  if (rc == 0 && stages >= 2) rc = bz_mtf(c, B, c->stream, true);
  if (rc == 0 && stages >= 3) rc = bz_entropy_emit(c, B, c->stream, true, option);
  hipStreamSynchronize(c->stream);
  hipFree(d_in);
  return rc;
}
extern "C" int zada_bz2_fetch(zada_ctx *z, const char *name, void *dst, uint64_t cap, uint64_t *nbytes) {
  if (!z || !name) return ZADA_E_INVALID;
  Ctx *c = &z->c;
  Bz2State *B = bz_state(c);
  hipSetDevice(c->device);
  const uint32_t nsb = B->nsb;
  const void *src = nullptr; uint64_t len = 0;
  uint32_t tmp[4];
  if (!strcmp(name, "n")) { src = B->n.p; len = 4ull * nsb; }
  else if (!strcmp(name, "bwt_index")) { src = B->bwt_index.p; len = 4ull * nsb; }
  else if (!strcmp(name, "crc")) { src = B->crc.p; len = 4ull * nsb; }
  else if (!strcmp(name, "inuse")) { src = B->inuse.p; len = 32ull * nsb; }
  else if (!strcmp(name, "rle")) { src = B->rle.p; len = B->ntot; }      // the move-to-front indices once stage 2 has run
  else if (!strcmp(name, "bwt")) { src = B->bwt.p; len = B->ntot; }
  else if (!strcmp(name, "mtf_n")) { src = B->mtf_n.p; len = 4ull * nsb; }
  else if (!strcmp(name, "soff")) { src = B->soff.p; len = 4ull * nsb; }
  else if (!strcmp(name, "sym")) { src = B->sym.p; len = 2ull * ((uint64_t)B->ntot + 2ull * nsb + 2); }
  else if (!strcmp(name, "res")) { src = B->res.p; len = 32ull * nsb; }
  else if (!strcmp(name, "sel_off")) { src = B->sel_off.p; len = 4ull * (nsb + 1); }
  else if (!strcmp(name, "sel")) { src = B->sel.p; len = B->selcap; }
  else if (!strcmp(name, "lens")) { src = B->lens.p; len = 6ull * BZ_LSTRIDE * nsb; }
  else if (!strcmp(name, "bwt_m")) { len = 8ull * B->m_hist.size(); if (nbytes) *nbytes = len; if (len > cap) return ZADA_E_INVALID; memcpy(dst, B->m_hist.data(), len); return 0; }
  else if (!strcmp(name, "dbg")) { src = B->dbg.p; len = 64ull * nsb; }
  else if (!strcmp(name, "woff")) { src = B->woff.p; len = 4ull * (nsb + 1); }
  else if (!strcmp(name, "words")) { src = B->words.p; len = 4ull * B->nwords; }
  else if (!strcmp(name, "info")) { tmp[0] = B->ntot; tmp[1] = (uint32_t)B->bwt_rounds; tmp[2] = nsb; tmp[3] = (uint32_t)B->gl_rows; if (cap < 16) return ZADA_E_INVALID; memcpy(dst, tmp, 16); if (nbytes) *nbytes = 16; return 0; }
  else return ZADA_E_INVALID;
  if (nbytes) *nbytes = len;
  if (len > cap) return ZADA_E_INVALID;
  if (len && hipMemcpy(dst, src, len, hipMemcpyDeviceToHost) != hipSuccess) return ZADA_E_HIP;
  return 0;
}
