// zada_lz.hip -- LZ77 stage of the encoder on gfx950 (hand-written HIP, wave64).
//
// Reference semantics: zip_lib/lz77.adb:460-943 (Info-Zip deflate_slow).  The sequential
// algorithm is refactored into data-parallel passes that compute the SAME token stream:
//
//   k_prev_links     per 32 KiB segment, for the reference's 15-bit hash (UPDATE_HASH, :553-557) and for 16-bit
//                    hashes of the first 4 .. 3+NLEVELS bytes: stable LDS radix sort of the positions; links to
//                    the previous position of the same bucket (what INSERT_STRING's prev[] holds, :566-573),
//                    nearest true 3 .. 2+NLEVELS byte match of every position, sorted orders, tail tables.  From 64 MiB on
//                    a workgroup takes a RUN of segments and makes the cross links of all but the run's first itself
//   k_cross_links    links the first position of each bucket to the previous segment's tail (with runs: a run's first segment)
//   k_bloom4         per segment a Bloom filter of its four-byte values (round 6)
//   k_cross_dist     continues the nearest-match searches that did not end inside their own segment: a sweep over all positions that asks the
//                    previous segment's filter before a level-4 walk and walks at most 16 steps, a second pass over the list of the walks still open
//   k_cross_scan     the longest of those walks, by reading the text backwards with a whole wave
//   k_bucket_limits  the chain-length limits (max_chain_length, and a quarter of it: :733-735) as distances
//   k_match          Longest_Match (:715-825), BOUNDED, for every position; window + links staged in LDS;
//                    full-chain and quarter-chain results, or a guess when the budget ran out
//   k_parse_spec     lazy-evaluation parser (:827-933), one lane per 512-byte chunk, speculatively
//                    started in the fresh state at each chunk boundary; marks the guesses it lands on
//   k_parse_fix      re-parses from the previous chunk's true exit state until it meets a
//                    history-free state of the speculative parse (splice); iterated to a fixpoint
//   k_match_demand   exact Longest_Match of the marked guesses, one wave per position over the sorted order
//   k_parse_spec_exact   (round 6) short lists of flagged chunks: one wave per chunk, the exact search inside the parse
//   k_tok_count / k_tok_compact   gather the true tokens into one global atom array
//   lz_stage         host driver: first pass, then parse / demand rounds until a parse has used exact values only
//
// Why the result is identical to the sequential reference: see DESIGN.md "LZ77 stage".
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include "zada_logic.h"
#include "zada_internal.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace zada {

// global memory accepts any byte address: one load instead of aligned pieces
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
// value of lane j (j wave uniform) of a vector register, as a scalar
#define RL(v, j) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (j)))
#define RL64(v, j) ((uint64_t)RL((uint32_t)(v), (j)) | ((uint64_t)RL((uint32_t)((v) >> 32), (j)) << 32))

// --------------------------------------------------------------------------------------------
// Layout of the LZ buffer: ONE stream [0, n), or a BATCH of independent streams (Zip entries), each starting at a multiple
// of 32 KiB and taking whole segments.  segend[seg] = offset of the end of the entry the segment belongs to, bit 31 set for
// the entry's first segment.  An entry's first segment has no previous segment (no links, planes or limits reach back),
// its first position is never a match source (NIL, lz77.adb:467) and every bound the kernels take from "the end of the
// input" (lz77.adb:858-865, 887-893) is the end of the entry.
// --------------------------------------------------------------------------------------------
// Workgroups are dealt to the eight XCDs in turn and every XCD has its own L2.  Neighbouring segments read each other's
// tables (the chains continue in the segment before), so neighbouring segments should share an L2: workgroup b takes the
// piece (b % 8) * (n / 8) + b / 8 of the grid's n pieces (the last n % 8 keep their place): every XCD walks a contiguous eighth.
#ifndef ZADA_NO_XCD_MAP
__device__ __forceinline__ uint32_t xcd_block() {
  const uint32_t n = gridDim.x, full = n & ~7u, b = blockIdx.x;
  return b < full ? (b & 7u) * (full >> 3) + (b >> 3) : b;
}
#else
__device__ __forceinline__ uint32_t xcd_block() { return blockIdx.x; }
#endif
__device__ __forceinline__ uint64_t lay_end(const Layout &L, uint64_t seg) { return L.segend ? (uint64_t)(L.segend[seg] & 0x7FFFFFFFu) : L.n; }
__device__ __forceinline__ bool lay_first(const Layout &L, uint64_t seg) { return L.segend ? (L.segend[seg] >> 31) != 0 : seg == 0; }
// inserted positions of a segment (every position with three bytes before the end of its entry: :841-843, 887-893)
__device__ __forceinline__ uint32_t lay_inserted(const Layout &L, uint64_t seg) {
  const uint64_t base = seg << 15, e = lay_end(L, seg), ie = e >= 2 ? e - 2 : 0;
  return ie > base ? (uint32_t)((ie - base) < 32768ull ? (ie - base) : 32768ull) : 0u;
}

// --------------------------------------------------------------------------------------------
// k_prev_links
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash3(const uint8_t *__restrict__ in, uint64_t p) {
  return (((uint32_t)in[p] << 10) ^ ((uint32_t)in[p + 1] << 5) ^ (uint32_t)in[p + 2]) & 0x7FFFu;
}

__device__ __forceinline__ uint32_t load24(const uint8_t *__restrict__ in, uint64_t p) {
  return (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
}
// eight bytes at p with three aligned dword loads (in is 4-byte aligned; the buffer is padded past n)
__device__ __forceinline__ uint64_t load8(const uint8_t *__restrict__ in, uint64_t p) {
  const uint32_t *w = (const uint32_t *)(in + (p & ~3ull));
  const uint32_t a = w[0], b = w[1], c = w[2], s = (uint32_t)(p & 3);
  return (uint64_t)__builtin_amdgcn_alignbyte(b, a, s) | ((uint64_t)__builtin_amdgcn_alignbyte(c, b, s) << 32);
}
__device__ __forceinline__ uint32_t hash3_of(uint64_t v) {
  return ((((uint32_t)v & 0xFF) << 10) ^ ((((uint32_t)v >> 8) & 0xFF) << 5) ^ (((uint32_t)v >> 16) & 0xFF)) & 0x7FFFu;
}
// 16-bit hash of the first L (4..6) bytes: level-L chains hold the candidates that can reach length >= L
__device__ __forceinline__ uint32_t hashL_of(uint64_t v, int L) {
  const uint32_t x = (uint32_t)v;
  const uint32_t y = (uint32_t)(v >> 32) & (L > 5 ? 0xFFFFu : L > 4 ? 0xFFu : 0u);
  return (x * 2654435761u + y * 0x9E3779B1u) >> 16;
}
__device__ __forceinline__ uint32_t hashL(const uint8_t *__restrict__ in, uint64_t p, int L) { return hashL_of(load8(in, p), L); }
__device__ __forceinline__ bool sameL(const uint8_t *__restrict__ in, uint64_t p, uint64_t q, int L) {
  const uint64_t mask = (1ull << (8 * L)) - 1ull;
  return ((load8(in, p) ^ load8(in, q)) & mask) == 0;
}

// Eight window bytes at any byte offset from three aligned dwords.  (gfx950 also accepts misaligned
// ds_read addresses, but measured 35 % slower in k_match than aligned pieces + alignbyte.)
__device__ __forceinline__ uint64_t lds_u64_at(const uint8_t *b, uint32_t o) {
  const uint32_t *w = (const uint32_t *)(b + (o & ~3u));
  const uint32_t x = w[0], y = w[1], z = w[2], s = o & 3u;
  return (uint64_t)__builtin_amdgcn_alignbyte(y, x, s) | ((uint64_t)__builtin_amdgcn_alignbyte(z, y, s) << 32);
}

// Four bytes at byte offset t of an LDS array whose last aligned word starts at t & ~3 (t <= size - 4): the second word is only read when
// the offset is not aligned, so that t = size - 4 does not touch the word behind the array (the staged 32 KiB end where the
// kernels' LDS allocation ends).
__device__ __forceinline__ uint32_t lds_u32_tail(const uint8_t *b, uint32_t t) {
  const uint32_t *wp = (const uint32_t *)(b + (t & ~3u));
  const uint32_t sh = t & 3u;
  return __builtin_amdgcn_alignbyte(wp[sh ? 1 : 0], wp[0], sh);
}

// --------------------------------------------------------------------------------------------
// Stable LSD radix sort of a segment's positions by a 16-bit key, entirely in LDS and registers.
// 16 waves; in every pass wave w owns the elements [w*2048, w*2048+2048) of the pass's input order and
// lane handles i = w*2048 + it*64 + lane (it = 0..31).  A lane keeps its 32 (element, key) pairs packed
// in registers for the whole pass, so the scatter may overwrite the arrays the pass was loaded from.
// --------------------------------------------------------------------------------------------
#ifdef ZADA_PL_STATS
__device__ unsigned long long g_sort_dbg[8];
#define SP_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_sort_dbg[k], t_ - tsp); tsp = t_; } } while (0)
#else
#define SP_STAMP(k) do {} while (0)
#endif
// Barrier for LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0): stores and loads share the
// counter on gfx9), and k_prev_links has scattered stores in flight in front of nearly every barrier -- tables, planes, sorted orders -- that nobody in
// the workgroup reads back: this one waits for the LDS operations (lgkmcnt) only and leaves the stores draining behind the next phase.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int NDIG, int SHIFT, bool LINEAR, int NPR>
__device__ __forceinline__ void sort_pass(const uint32_t (&pr)[NPR], uint32_t *dst,
                                          uint32_t *cnt /*[16][NDIG]*/, uint32_t *wsum /*[16]*/, uint32_t i0, int rem) {
  // This lane's element `it` is i0 + 64*it (i0 = w*2048 + lane) and exists iff 64*it < rem.  Element and key travel as
  // one 32-bit word, element | key << 16: one LDS write per element and pass, one read in the next.  LINEAR (first
  // pass): the elements are still in place, pr holds the keys only, two per register; otherwise pr[it] is the word.
  auto word = [&](int it) -> uint32_t {
    if (LINEAR) return (i0 + it * 64) | (((pr[it >> 1] >> (16 * (it & 1))) & 0xFFFFu) << 16);
    return pr[it];
  };
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
#ifdef ZADA_PL_STATS
  unsigned long long tsp = clock64();
#endif
  for (int i = tid; i < NDIG * 16; i += 1024) cnt[i] = 0;
  lds_barrier();
  SP_STAMP(0);
  uint32_t *mycnt = cnt + w * NDIG;
  // phase A: per (wave, digit) histogram; the value an element gets back is its rank among the wave's elements with
  // the same digit, in element order: the wave's instructions are issued in element order, and within one
  // instruction the LDS unit serialises lanes that hit the same counter in lane order (checked on the hardware by
  // tests/probes/lds_atomic_order.hip and, end to end, by every parity test).  Ranks are kept packed two per register.
  uint32_t rk[16];
#pragma unroll
  for (int k = 0; k < 16; k++) rk[k] = 0;
#pragma unroll
  for (int it = 0; it < 32; it++) {
    const uint32_t d = (word(it) >> (16 + SHIFT)) & (uint32_t)(NDIG - 1);
    uint32_t r = 0;
    if (it * 64 < rem) r = atomicAdd(&mycnt[d], 1u);
    rk[it >> 1] |= r << (16 * (it & 1));
  }
  lds_barrier();
  SP_STAMP(1);
  // phase B: exclusive scan over (digit major, wave minor)
  {
    constexpr int PER = NDIG * 16 / 1024;           // 4 (256 digits) or 2 (128 digits)
    uint32_t v[PER], s = 0;
    for (int k = 0; k < PER; k++) { const int idx = tid * PER + k; v[k] = cnt[(idx & 15) * NDIG + (idx >> 4)]; s += v[k]; }
    uint32_t incl = s;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[w] = incl;
    lds_barrier();
    uint32_t base = 0;
    for (int k = 0; k < w; k++) base += wsum[k];
    uint32_t run = base + incl - s;
    for (int k = 0; k < PER; k++) { const int idx = tid * PER + k; cnt[(idx & 15) * NDIG + (idx >> 4)] = run; run += v[k]; }
  }
  lds_barrier();
  SP_STAMP(2);
  // phase C: stable scatter
#pragma unroll
  for (int it = 0; it < 32; it++) {
    if (it * 64 < rem) {
      const uint32_t wd = word(it), d = (wd >> (16 + SHIFT)) & (uint32_t)(NDIG - 1);
      dst[mycnt[d] + ((rk[it >> 1] >> (16 * (it & 1))) & 0xFFFFu)] = wd;
    }
  }
  lds_barrier();
  SP_STAMP(3);
}

// Per 32 KiB segment (grid = segments, block = 1024).  n_ins = number of inserted positions (n - 2).
//
// The reference walks, for position p, the chain of earlier positions with the same 15-bit hash
// (nearest first, at most `chain` of them, within 32 505/32 506 bytes: lz77.adb:813-860).  The match kernel
// visits far fewer candidates with the same result thanks to what this kernel prepares:
//   * level 3: the nearest earlier position with the same THREE bytes (plane d[0]); the sorted order
//     S3 / tags T3 / buckets bsc3 of the 15-bit hash (for the next segment's continuation in
//     k_cross_dist and for the chain-length limits of k_bucket_limits);
//   * levels 4 .. 3+NLEVELS (16-bit hashes of the first L bytes): the chain links of the level
//     (16-bit distance to the previous position of the same bucket) and, for all but the last level,
//     the nearest earlier position sharing L bytes (planes d[1..]).  A candidate can only beat a
//     length-L match if it shares L+1 bytes, so the match kernel only walks the LAST level's chain.
// Every level: sort the segment's positions by the level's hash (two counting passes, keys travelling
// with the positions so that nothing is gathered from global memory), turn the sorted order into
// position-indexed links P in LDS, then walk those links position-parallel against the segment's bytes
// (also in LDS); every output plane is written with coalesced stores.
// markers in the level-3 plane for k_cross_dist: the search goes on in the previous segment (p has / has no earlier position
// with its 15-bit hash in its own segment); only the head of the chain at exactly MAX_DIST is left to check
constexpr uint32_t DIST3_CONTINUE = 0xFFFF, DIST3_HEADCHK = 0xFFFE, DIST3_CONT_FIRST = 0xFFFD, DISTL_CONTINUE = 0x8000;
// levels >= 4: the walk inside the segment was given up after WALK_CAP candidates (a position whose bucket is full of another
// string: one or two per segment, and the whole workgroup waited for them); k_cross_dist walks it, next to thousands of others
#ifndef ZADA_WALK_CAP
#define ZADA_WALK_CAP 16
#endif
constexpr uint32_t DISTL_GAVEUP = 0x7FFF, WALK_CAP = ZADA_WALK_CAP;
static_assert(MAX_DIST < 0x7FFF, "the give-up marker is no distance");
constexpr uint32_t HEAVY_STRIDE = 2048, HEAVY_CAP = HEAVY_STRIDE - 1;       // per segment: the list of its heavy 15-bit buckets
static_assert(MAX_DIST < 0x8000, "continue markers of the planes");
// LDS of k_prev_links: the sort's two halves (128 KiB), its counters (16 KiB + 64 B); with runs of segments per workgroup the previous segment's bytes
// are staged from 131 072 on (32 KiB: over the counters, which are dead during the walks of the one level that needs them).
constexpr int PL_LDS = 144 * 1024 + 64, PL_LDS_RUNS = 160 * 1024, PL_PREV_OFF = 131072;
constexpr int PL_HB_OFF = PL_PREV_OFF + 12288;                          // the bit map of bucket heads (4 KiB), clear of the level's bucket-start masks
static_assert(PL_PREV_OFF + 32768 <= PL_LDS_RUNS, "k_prev_links: LDS map of the fused cross links");
template <bool RUNS>
__global__ void __launch_bounds__(1024) k_prev_links(const uint8_t *__restrict__ in, Layout L, int kfull, int kquarter,
                                                     LevelPtrs lv,
                                                     uint16_t *__restrict__ S3, uint8_t *__restrict__ T3, uint32_t *__restrict__ bsc3,
                                                     DistPlanes dp, RunPtrs rp, unsigned long long *__restrict__ dbg, uint32_t *__restrict__ segmax,
                                                     uint16_t *__restrict__ heavy, uint32_t seg0, uint32_t run_len, uint32_t seg_end) {
  // run_len consecutive segments per workgroup, one after the other (seg0 + blockIdx.x * run_len ...; up to seg_end).  From the second segment of
  // a run on the workgroup makes the CROSS LINKS itself ("fused"): the first member of each of a level's buckets is linked to the last member
  // of that bucket in the segment before -- whose tails table this very workgroup wrote a moment ago -- while the sorted order is still in LDS
  // (neighbouring lanes hold neighbouring keys: the table is read in ascending order, sector by sector, instead of being staged as 128 KB per
  // segment and level by k_cross_links), and the searches for the nearest four-byte match that end at such a link are settled against the
  // previous segment's bytes (staged next to the segment's own, where the sort's counters were).  k_cross_links is left with the first segment of every run.
  // With runs (RUNS) a workgroup initialises the tails tables of its segments ("no member": 0xFFFF), so that the next segment can take an entry as it
  // stands; without them the kernel is what it was (tables not initialised, their readers check an entry by hashing the position it names).
#ifdef ZADA_PL_STATS
  unsigned long long tprev = clock64(); int tph = 8;
#define PL_STAMP() do { lds_barrier(); if (threadIdx.x == 0) { unsigned long long t = clock64(); atomicAdd(&dbg[tph], t - tprev); tprev = t; } tph++; } while (0)
#else
#define PL_STAMP() do {} while (0)
#endif
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint16_t *A = (uint16_t *)smem;                 // 64 KiB
  uint16_t *B = A + 32768;                        // 64 KiB
  uint32_t *AB = (uint32_t *)smem;                // A and B as one array of 32 768 words (the sorts)
  uint32_t *cnt = (uint32_t *)(B + 32768);        // 16 KiB
  uint32_t *wsum = cnt + 4096;                    // 64 B
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint8_t *pbL = smem + PL_PREV_OFF;                                       // fused: the previous segment's bytes (over the sort's counters, when they are dead)
#pragma unroll 1
  for (uint32_t jrun = 0; jrun < (RUNS ? run_len : 1u); jrun++) {
  const uint64_t seg = (uint64_t)blockIdx.x * run_len + jrun + seg0, base = seg * 32768ull;      // (seg0: the launch covers the segments of one piece of the input)
  if (seg >= seg_end) break;
  const uint32_t m = lay_inserted(L, seg);
  const bool first_seg = lay_first(L, seg);
  const bool fused = RUNS && jrun > 0 && !first_seg;                            // cross links from this workgroup (it has just finished segment seg - 1)
  const bool prev_first = fused && lay_first(L, seg - 1);

#ifdef ZADA_PL_STATS
  tph = 8;
#endif
  uint32_t *bsc = bsc3 + seg * 32768ull;          // bucket start | count << 16
  uint16_t *s3 = S3 + seg * 32768ull;
  uint8_t *t3 = T3 + seg * 32768ull;              // top three bits of byte 0: what the 15-bit hash drops
  uint16_t *hvy = heavy + seg * HEAVY_STRIDE;     // [0]: number of heavy buckets (0xFFFF: too many to list), then their hashes
  const uint32_t heavy_thr = kquarter / 2 > 0 ? (uint32_t)kquarter / 2 : 1u;
#ifdef ZADA_OLD_INIT
  for (int l = 0; l < NLEVELS; l++) {
    uint16_t *tail = lv.tails[l] + seg * 65536ull;                  // 65536 buckets per level
    for (int i = tid; i < 65536 / 8; i += 1024) ((uint4 *)tail)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
  }
  for (uint32_t e = tid; e < m; e += 1024) dp.dlim[base + e] = 0xFFFFFFFFu;   // no chain-length limit unless k_bucket_limits finds one
#endif
  // (The tails tables -- last position of each of the 65 536 buckets of a level -- are NOT initialised: only occupied buckets are
  // written below, and k_cross_links tells a stale entry from a tail by hashing the position it names.  "No chain-length limit",
  // the default of dlim, is a memset on the second stream: lz_shard.)
  const uint8_t *sin = in + base;
  PL_STAMP();   // 8: table init
  uint32_t *F = cnt;                               // level 3: bucket-start bitmask of the sorted order, 1024 words (+1 spill word)
  uint16_t *LW = (uint16_t *)(cnt + 1040);         // per word: last non-empty word at or before it
  const uint8_t *sb = (const uint8_t *)A;          // the segment's bytes (after the keys in A are dead)
  uint16_t *P = B;                                 // position-indexed links (after the sorted positions in B are dead)
  auto lb8 = [&](uint32_t e) -> uint64_t {         // the bytes of the segment at e, from LDS: at least five are valid
    // (two aligned words hold bytes e .. e+4 whatever the alignment; the levels compare 3 .. 5 bytes)
    const uint32_t *wp = (const uint32_t *)(sb + (e & ~3u));
    const uint32_t a = wp[0], b = wp[1], s = e & 3u;
    return (uint64_t)__builtin_amdgcn_alignbyte(b, a, s) | ((uint64_t)(b >> (8u * s)) << 32);
  };
  static_assert(3 + NLEVELS <= 5 + 1, "lb8 yields five valid bytes");
#pragma unroll 1
  for (int lvl = 0; lvl <= NLEVELS; lvl++) {
    const int L = 3 + lvl;
    // this lane's element `it` of a sweep is i0 + 64*it and exists iff 64*it < rem.  i0 is made opaque so that
    // the 32-fold unrolled sweeps below recompute their addresses (base + immediate offset) per level
    // instead of having hundreds of loop-invariant values hoisted out of the level loop and spilled.
    uint32_t i0 = (uint32_t)w * 2048 + lane;
    asm volatile("" : "+v"(i0));
    const int rem = (int)m - (int)i0;

    // ---- sort: AB[i] := element | key << 16 in (key, position) order (A and B as one array of 32-bit words) ----
    {
      {
        uint32_t key[16];
#pragma unroll
        for (int k = 0; k < 16; k++) key[k] = 0;
        const uint32_t sh = i0 & 3u;
        // the loads are not guarded (the buffer is padded past its end), so that eight elements' words are in flight together
        // instead of one global round trip per element; what lies beyond the segment's elements gets key 0 and is not sorted.
        // From the second level on the segment's bytes are still in LDS (A, staged for the level before: nothing has written there since).
        auto build = [&](auto wp) {
#pragma unroll
          for (int g8 = 0; g8 < 4; g8++) {
            uint32_t wa[8], wb[8], wc[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const int it = g8 * 8 + q; wa[q] = wp[it * 16]; wb[q] = wp[it * 16 + 1]; wc[q] = wp[it * 16 + 2]; }
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int it = g8 * 8 + q;
              const uint64_t v = (uint64_t)__builtin_amdgcn_alignbyte(wb[q], wa[q], sh) | ((uint64_t)__builtin_amdgcn_alignbyte(wc[q], wb[q], sh) << 32);
              uint32_t k = (lvl == 0) ? hash3_of(v) : hashL_of(v, L);
              if (!(it * 64 < rem)) k = 0;
              key[it >> 1] |= k << (16 * (it & 1));
            }
          }
        };
        if (lvl == 0) {
          build((const uint32_t *)(sin + (i0 & ~3u)));
          // The segment's tables are initialised HERE, behind the key loads and in front of the sorts: loads wait for the stores issued before them
          // (one counter), and the two sort passes that follow need no global memory -- the stores drain behind them.
          for (int i = tid; i < 32768 / 4; i += 1024) ((uint4 *)bsc)[i] = make_uint4(0, 0, 0, 0);
          if (RUNS) {
            for (int l = 0; l < NLEVELS; l++) {
              uint16_t *tl = lv.tails[l] + seg * 65536ull;                       // 65536 buckets per level: "no member"
              for (int i = tid; i < 65536 / 8; i += 1024) ((uint4 *)tl)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
            }
          }
        } else build((const uint32_t *)(sb + (i0 & ~3u)));
        sort_pass<256, 0, true>(key, AB, cnt, wsum, i0, rem);
      }
      uint32_t pr[32];
      {
        const uint32_t *pp = AB + i0;
#pragma unroll
        for (int it = 0; it < 32; it++) pr[it] = (it * 64 < rem) ? pp[it * 64] : 0u;
      }
      if (lvl == 0) sort_pass<128, 8, false>(pr, AB, cnt, wsum, i0, rem);
      else sort_pass<256, 8, false>(pr, AB, cnt, wsum, i0, rem);
    }
    // (the one place where stores of different lanes to the same address follow each other -- a bucket's record over the table's zero --
    // gets a full barrier; the zeros have had two sort passes to arrive)
    if (lvl == 0) __syncthreads();
    PL_STAMP();
    // ---- links: element | last-of-bucket << 15 | distance to the bucket's previous element << 16, in registers ----
    uint16_t *tail = lvl > 0 ? lv.tails[lvl - 1] + seg * 65536ull : nullptr;
    const bool want_runs = lvl == 0 || lvl == NLEVELS;   // bucket boundaries of the sorted order are needed
    uint32_t ed[32];
    // keyed (fused cross links): a bucket's first member has no link yet -- its key travels in the link's place (and from there into the link array P) and a bit
    // map of such heads, HB, says which entries of P are keys: the cross-link phase below then needs neither the bytes nor a hash
    const bool keyed = RUNS && fused && lvl > 0;
    uint32_t *HB = (uint32_t *)(smem + PL_HB_OFF);
    uint32_t firstmask = 0;                                            // bit `it`: the lane's element `it` starts a bucket
    if (keyed) HB[tid] = 0;                                            // (the sort's counters are dead; the barrier in front of the scatter lies between this and its use)
    {
      const uint32_t *pp = AB + i0;
      uint32_t *pf = F + (i0 >> 5);
#pragma unroll
      for (int it = 0; it < 32; it++) {
        uint32_t x = 0;
        bool first = false;
        if (it * 64 < rem) {
          const uint32_t p0 = pp[it * 64], e = p0 & 0xFFFFu, k = p0 >> 16;
          uint32_t pm = 0;
          first = i0 + it * 64 == 0;
          if (!first) { pm = pp[it * 64 - 1]; first = (pm >> 16) != k; }
          const bool last = (it * 64 + 1 == rem) || ((pp[it * 64 + 1] >> 16) != k);
          uint32_t d = 0;
          if (!first) { const uint32_t e0 = pm & 0xFFFFu; if (!(first_seg && e0 == 0)) d = e - e0; }     // NIL = position 0, lz77.adb:467
          else if (keyed) d = k;
          firstmask |= (uint32_t)first << it;
          x = e | ((uint32_t)last << 15) | (d << 16);
          if (lvl > 0 && last) tail[k] = (uint16_t)e;
        }
        ed[it] = x;
        if (want_runs) {
          const unsigned long long mk = __ballot(first);
          if (lane == 0) { pf[it * 2] = (uint32_t)mk; pf[it * 2 + 1] = (uint32_t)(mk >> 32); }
        }
      }
    }
    if (want_runs) {
      if (tid == 0) { cnt[2048] = 0; cnt[2049] = 0; } // (free between the sorts: the largest bucket of the segment, the number of heavy ones; level 3)
      lds_barrier();
      {
        // LW[wd] = last word index <= wd whose F word is non-zero (word 0 always is: element 0 starts a bucket)
        uint32_t v = F[tid] != 0 ? (uint32_t)tid : 0u;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off); if (lane >= off) v = v > t ? v : t; }
        if (lane == 63) wsum[w] = v;
        lds_barrier();
        uint32_t before = 0;
        for (int k = 0; k < w; k++) before = before > wsum[k] ? before : wsum[k];
        LW[tid] = (uint16_t)(v > before ? v : before);
      }
      lds_barrier();
      // start of i's bucket = highest set bit of F at or below i
      auto bucket_start = [&](uint32_t i) -> uint32_t {
        const uint32_t wd = i >> 5;
        const uint32_t own = F[wd] & (0xFFFFFFFFu >> (31 - (i & 31)));
        if (own) return (wd << 5) + 31 - __clz((int)own);
        const uint32_t pw = LW[wd - 1];
        return (pw << 5) + 31 - __clz((int)F[pw]);
      };
      if (lvl == 0) {
        uint32_t mx = 0;                               // largest bucket of the segment: k_bucket_limits skips segments without long ones
#pragma unroll
        for (int it = 0; it < 32; it++) {
          const uint32_t i = i0 + it * 64;
          if ((ed[it] >> 15) & 1u) {
            const uint32_t bs = bucket_start(i), hb = AB[i] >> 16, members = i - bs + 1;
            bsc[hb] = bs | (members << 16);
            mx = mx > members ? mx : members;
            // heavy buckets (half a quarter chain and more): the only ones k_bucket_limits has to look at -- a bucket needs a
            // quarter chain's worth of members over two segments to get a limit, so it is heavy in one of the two
            if (members >= heavy_thr) { const uint32_t kh = atomicAdd(&cnt[2049], 1u); if (kh < HEAVY_CAP) hvy[1 + kh] = (uint16_t)hb; }
          }
        }
        for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(mx, off); mx = mx > o ? mx : o; }
        if (lane == 0 && mx) atomicMax(&cnt[2048], mx);
      } else {
        // last level: the bucket of a position is a contiguous run of the sorted order, which the demand pass
        // of the match kernel scans instead of chasing links.  Written out: the sorted order S, and per
        // position its index in S and the number of bucket members before it (planes, staged as one word per position in AB).
        lds_barrier();                             // keys (A) and sorted positions (B) are dead from here
        uint16_t *sK = rp.S + seg * 32768ull;
#pragma unroll
        for (int it = 0; it < 32; it++) {
          if (it * 64 < rem) {
            const uint32_t i = i0 + it * 64, e = ed[it] & 0x7FFFu;
            sK[i] = (uint16_t)e;
            AB[e] = i | ((i - bucket_start(i)) << 16);       // (one scatter for both planes)
          }
        }
        lds_barrier();
        for (uint32_t i = tid; i < (m + 7) / 8; i += 1024) {
          const uint4 p = ((const uint4 *)AB)[2 * i], q = ((const uint4 *)AB)[2 * i + 1];
          uint4 lo, hi;
          lo.x = (p.x & 0xFFFFu) | (p.y << 16); lo.y = (p.z & 0xFFFFu) | (p.w << 16); lo.z = (q.x & 0xFFFFu) | (q.y << 16); lo.w = (q.z & 0xFFFFu) | (q.w << 16);
          hi.x = (p.x >> 16) | (p.y & 0xFFFF0000u); hi.y = (p.z >> 16) | (p.w & 0xFFFF0000u); hi.z = (q.x >> 16) | (q.y & 0xFFFF0000u); hi.w = (q.z >> 16) | (q.w & 0xFFFF0000u);
          ((uint4 *)(rp.idx + base))[i] = lo;
          ((uint4 *)(rp.cnt + base))[i] = hi;
        }
      }
    }
    lds_barrier();                                 // keys (A) and sorted positions (B) are dead from here
    if (lvl == 0 && tid == 0) { segmax[seg] = cnt[2048]; hvy[0] = cnt[2049] <= HEAVY_CAP ? (uint16_t)cnt[2049] : (uint16_t)0xFFFF; }
    if (!keyed) { for (uint32_t i = tid; i < (m + 16 + 15) / 16; i += 1024) ((uint4 *)A)[i] = ((const uint4 *)sin)[i]; }   // A := bytes (keyed: after the cross links, A holds the table first)
#pragma unroll
    for (int it = 0; it < 32; it++) {
      if (it * 64 < rem) {
        const uint32_t e = ed[it] & 0x7FFFu;
        P[e] = (uint16_t)(ed[it] >> 16);
        if (keyed && ((firstmask >> it) & 1u)) atomicOr(&HB[e >> 5], 1u << (e & 31u));
        if (lvl == 0) s3[i0 + it * 64] = (uint16_t)e;
      }
    }
    lds_barrier();
    if (lvl == 0) {
#pragma unroll
      for (int it = 0; it < 32; it++) {              // T3: tags for the cross-segment continuation (k_cross_dist)
        if (it * 64 < rem) {
          const uint32_t e = ed[it] & 0x7FFFu, b0 = sb[e], b1 = sb[e + 1];
          t3[i0 + it * 64] = (uint8_t)((b0 >> 5) | ((b1 & 7u) << 3) | ((b0 & 3u) << 6));
        }
      }
    } else {
      if (keyed) {
        // Cross links, fused: the first member of each of the level's buckets is linked to the bucket's last member in the segment before -- what
        // k_cross_links does, but with the links where they are, in LDS, and the previous segment's tails table (which this workgroup wrote itself,
        // initialised) staged half at a time, coalesced, in the sort's first half (the segment's bytes are staged there afterwards).  (Looked up
        // straight from memory -- two-byte gathers, 64 addresses an instruction -- the same entries cost 47 000 cycles per level: 90 cycles an
        // instruction in the memory pipeline; with the keys hashed again from the bytes, position by position, 50 000: LDS instructions.)  A lane
        // takes the heads among its 32 positions from the bit map, four at a time.  Round r takes the keys with bit 15 = r: a head that has been
        // linked holds a distance (<= MAX_DIST < 2^15) or 0 and is never taken for a key of round 1.
        const uint16_t *tab = (const uint16_t *)A;
        const uint16_t *tprev = lv.tails[lvl - 1] + (seg - 1) * 65536ull;
        const uint32_t hb = HB[tid], e00 = (uint32_t)tid * 32u;
#pragma unroll 1
        for (uint32_t r = 0; r < 2; r++) {
          if (r) lds_barrier();                      // (the half before this one has been read)
          {
            const uint4 *src = (const uint4 *)(tprev + r * 32768u);
            const uint4 v0 = src[tid], v1 = src[tid + 1024], v2 = src[tid + 2048], v3 = src[tid + 3072];
            ((uint4 *)A)[tid] = v0; ((uint4 *)A)[tid + 1024] = v1; ((uint4 *)A)[tid + 2048] = v2; ((uint4 *)A)[tid + 3072] = v3;
          }
          lds_barrier();
          uint32_t w = hb;
          while (w) {
            uint32_t eb[4], kk[4], tt[4];
            bool vv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { vv[q] = w != 0; eb[q] = e00 + (vv[q] ? (uint32_t)__ffs((int)w) - 1u : 0u); w &= w - 1u; }
#pragma unroll
            for (int q = 0; q < 4; q++) kk[q] = P[eb[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) tt[q] = tab[kk[q] & 32767u];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              if (!vv[q] || (kk[q] >> 15) != r) continue;
              // the bucket's last member in the previous segment, if it has one there, within MAX_DIST, and not position 0 of an entry (:467)
              const uint32_t t = tt[q], dx = eb[q] + 32768u - t;
              P[eb[q]] = (t != 0xFFFFu && dx <= (uint32_t)MAX_DIST && !(prev_first && t == 0)) ? (uint16_t)dx : (uint16_t)0;
            }
          }
        }
        lds_barrier();
        if (lvl < NLEVELS) { for (uint32_t i = tid; i < (m + 16 + 15) / 16; i += 1024) ((uint4 *)A)[i] = ((const uint4 *)sin)[i]; }   // A := bytes, for the walks and the next level's keys
      }
      uint16_t *prevl = lv.prev[lvl - 1];
      for (uint32_t i = tid; i < (m + 7) / 8; i += 1024) ((uint4 *)(prevl + base))[i] = ((const uint4 *)P)[i];
    }
    PL_STAMP();
    // ---- nearest earlier position of the segment with the same L bytes: walk the links, newest first ----
    // Most positions are settled by their first candidate; the few that have to walk through hash
    // collisions are compacted into a queue and walked densely (all lanes busy) in rounds.
    if (lvl < NLEVELS) {
      const uint64_t lmask = (1ull << (8 * L)) - 1ull;
      uint16_t *plane = dp.d[lvl] + base;
      // not found inside this segment: continue in k_cross_dist.  Level 3 (lvl 0) only looks TOO_FAR back: a three-byte match
      // farther away is dropped by the parser anyway (lz77.adb:867-871) and any longer match is found through the levels
      // above, so plane d[0] holds the nearest three-byte match within 4096, or none (0), or MAX_DIST for the one candidate
      // the reference accepts at exactly that distance (the head of the 15-bit chain, :850 vs :820; k_cross_dist).
      // Levels >= 4 say where to continue: the chain ended at q, the first member of the bucket in this segment, whose link
      // k_cross_links will point into the previous segment (0x8000 | e - q; distances proper stay below 0x8000).
      // Fused (levels >= 4): q's link already points into the previous segment -- or is none, when its bucket has no member in reach there, and then
      // nothing has (older members lie farther back still).  The link leads to the most recent position there with the same hash, which usually IS the
      // nearest match: its bytes are compared here (the previous segment's are staged in LDS); a hash collision leaves the search to k_cross_dist.
      auto dflt_of = [&](uint32_t e, uint32_t q) -> uint32_t {
        if (first_seg) return 0u;
        if (lvl == 0) return e < (uint32_t)TOO_FAR ? (q == e ? DIST3_CONT_FIRST : DIST3_CONTINUE) : (q == e ? DIST3_HEADCHK : 0u);
        if (RUNS && fused) {
          const uint32_t step = P[q];
          if (step == 0) return 0u;
          const uint32_t t = q + 32768u - step, d = e + 32768u - t;
          if (d > (uint32_t)MAX_DIST) return 0u;
          if (t <= 32768u - 4u) {
            const uint32_t theirs = lds_u32_tail(pbL, t);                               // (t <= 32764: nothing behind the staged bytes is read)
            if (theirs == (uint32_t)lb8(e)) return d;
          }
        }
        return DISTL_CONTINUE | (e - q);
      };
      static_assert(NLEVELS <= 2, "the candidate compare of the fused cross links takes four bytes");
      constexpr uint32_t QCAP = 4000;
      uint32_t *Qa = (uint32_t *)(smem + 32800), *Qb = Qa + QCAP;   // behind the 32 784 staged bytes
      uint32_t *qn = RUNS ? (uint32_t *)(smem + 32800 + 8 * QCAP) : wsum;     // (RUNS: the counters' place may hold the previous segment's bytes)
      static_assert(32800 + 8 * QCAP + 8 <= 65536, "queues and their counters behind the staged bytes");
      const unsigned long long ltm = (1ull << lane) - 1ull;
      if (tid < 2) qn[tid] = 0;
      if (RUNS && fused && lvl > 0) {
        const uint4 *bs = (const uint4 *)(in + base - 32768ull);
        for (int i = tid; i < 32768 / 16; i += 1024) ((uint4 *)(smem + PL_PREV_OFF))[i] = bs[i];
      }
      lds_barrier();
      // up to `maxs` further candidates of position e, starting behind q; true = settled (dl valid)
      auto walk = [&](uint32_t e, uint32_t &q, uint64_t mine, uint32_t maxs, uint32_t &dl) -> bool {
        uint32_t step = P[q];
        for (uint32_t sidx = 0; sidx < maxs; sidx++) {
          if (step - 1u >= q) { dl = dflt_of(e, q); return true; }      // no link, or (fused) one that leaves the segment: step > q
          q -= step;
          const uint32_t dist = e - q;
          step = P[q];                                              // next link and this candidate's bytes in one LDS round trip
          const uint64_t theirs = lb8(q);
          // beyond MAX_DIST nothing qualifies (level 3: beyond TOO_FAR nothing matters)
          if (dist > (uint32_t)(lvl == 0 ? TOO_FAR : MAX_DIST)) { dl = 0; return true; }
          if ((theirs & lmask) == mine) { dl = dist; return true; }
        }
        return false;
      };
      auto push = [&](bool pend, uint32_t entry, uint32_t *dstq, uint32_t *cntp) -> bool {   // false: queue full, not stored
        const unsigned long long mk = __ballot(pend);
        if (mk == 0) return true;
        uint32_t b0 = 0;
        const int leader = __ffsll((long long)mk) - 1;
        if (lane == leader) b0 = atomicAdd(cntp, (uint32_t)__popcll(mk));
        b0 = __shfl(b0, leader);
        const uint32_t idx = b0 + __popcll(mk & ltm);
        if (pend && idx < QCAP) { dstq[idx] = entry; return true; }
        return !pend;
      };
      // a walk is followed to its end -- or, where k_cross_dist comes by afterwards (levels >= 4, not an entry's first segment), as far
      // as WALK_CAP candidates beyond the rounds of eight below
      const bool capped = lvl > 0 && !first_seg && WALK_CAP > 0;
      const uint32_t maxs_last = capped ? WALK_CAP : 1u << 30;
      for (uint32_t e0 = 0; e0 < m; e0 += 4096) {                   // first candidate of every position, four positions per lane at a time
        uint32_t ee[4], st[4];
        uint64_t mn[4], th[4];
        bool ex[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t e = e0 + 1024u * j + tid; ex[j] = e < m; ee[j] = ex[j] ? e : 0u; }
#pragma unroll
        for (int j = 0; j < 4; j++) { mn[j] = lb8(ee[j]) & lmask; st[j] = P[ee[j]]; }          // (no conditions: the LDS reads overlap)
#pragma unroll
        for (int j = 0; j < 4; j++) th[j] = lb8(st[j] <= ee[j] ? ee[j] - st[j] : ee[j]) & lmask;   // (no link, or one that leaves the segment: its own bytes)
        // fused: a position whose own link leaves the segment (a bucket's first member -- every other position or so) has its candidate in the previous
        // segment, at the distance the link says; its four bytes are fetched with everything else, without a condition
        uint32_t pw[4] = {0, 0, 0, 0};
        if (RUNS && fused && lvl > 0) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t t = st[j] > ee[j] ? ee[j] + 32768u - st[j] : 0u, tr = t <= 32768u - 4u ? t : 0u;
            pw[j] = lds_u32_tail(pbL, tr);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t e = ee[j], step = st[j];
          uint32_t dl = 0, q = e;
          bool pend = false;
          if (!(ex[j] && step - 1u < e)) {
            if (RUNS && fused && lvl > 0) {
              // no link: nothing in reach in the previous segment either; a link into it: the candidate there, compared (dflt_of (e, e) without its branches)
              if (step != 0) dl = (e + 32768u - step <= 32768u - 4u && pw[j] == (uint32_t)mn[j]) ? step : DISTL_CONTINUE;
            } else dl = dflt_of(e, e);
          }
          if (ex[j] && step - 1u < e) {
            q = e - step;
            if (lvl == 0) {
              // the head of the 15-bit chain: accepted up to exactly MAX_DIST (:850); the walk behind it only to TOO_FAR
              if (th[j] == mn[j] && (step <= (uint32_t)TOO_FAR || step == (uint32_t)MAX_DIST)) dl = step;
              else if (step >= (uint32_t)TOO_FAR) dl = 0;
              else pend = true;
            } else {
              if (step > (uint32_t)MAX_DIST) dl = 0;
              else if (th[j] == mn[j]) dl = step;
              else pend = true;
            }
          }
          if (!push(pend, e | (q << 16), Qa, &qn[0])) { if (!walk(e, q, mn[j], maxs_last, dl)) dl = DISTL_GAVEUP; pend = false; }   // (queue full)
          if (ex[j] && !pend) plane[e] = (uint16_t)dl;
        }
      }
      lds_barrier();
      PL_STAMP();
      for (int cur = 0;; cur ^= 1) {
        const uint32_t nq = qn[cur] < QCAP ? qn[cur] : QCAP;
        if (nq == 0) break;
#ifdef ZADA_PL_STATS
        if (tid == 0 && lvl == 1) { atomicAdd(&dbg[24], 1ull); atomicAdd(&dbg[25], (unsigned long long)nq); if (nq <= 1024) atomicAdd(&dbg[26], (unsigned long long)nq); if (cur == 0 && qn[0] > QCAP) atomicAdd(&dbg[30], (unsigned long long)(qn[0] - QCAP)); }
        unsigned long long tq0 = clock64();
#endif
        lds_barrier();
        if (tid == 0) qn[cur ^ 1] = 0;
        lds_barrier();
        if (nq > 1024) {
          // Up to four entries per lane, eight candidates each, in step with each other: the LDS reads of the four walks
          // overlap (a walk is one dependent LDS round trip per candidate).
          static_assert(QCAP <= 4096, "four entries per lane cover the queue");
          const uint32_t *Qs = cur ? Qb : Qa;
          uint32_t e4[4], q4[4], st4[4], dl4[4];
          uint64_t mn4[4];
          bool val[4], act[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t idx = 1024u * j + tid;
            val[j] = idx < nq;
            const uint32_t ent = Qs[val[j] ? idx : 0u];
            e4[j] = ent & 0x7FFFu; q4[j] = ent >> 16; act[j] = val[j]; dl4[j] = 0;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) { mn4[j] = lb8(e4[j]) & lmask; st4[j] = P[q4[j]]; }
          for (int sidx = 0; sidx < 8; sidx++) {
            uint32_t qx[4], ns[4];
            uint64_t th[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (act[j] && st4[j] - 1u >= q4[j]) { dl4[j] = dflt_of(e4[j], q4[j]); act[j] = false; }    // the chain ends inside the segment
              qx[j] = act[j] ? q4[j] - st4[j] : q4[j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { ns[j] = P[qx[j]]; th[j] = lb8(qx[j]) & lmask; }      // next link and this candidate's bytes
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (act[j]) {
                q4[j] = qx[j]; st4[j] = ns[j];
                const uint32_t dist = e4[j] - q4[j];
                // beyond MAX_DIST nothing qualifies (level 3: beyond TOO_FAR nothing matters)
                if (dist > (uint32_t)(lvl == 0 ? TOO_FAR : MAX_DIST)) { dl4[j] = 0; act[j] = false; }
                else if (th[j] == mn4[j]) { dl4[j] = dist; act[j] = false; }
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (val[j] && !act[j]) plane[e4[j]] = (uint16_t)dl4[j];
            push(val[j] && act[j], e4[j] | (q4[j] << 16), cur ? Qa : Qb, &qn[cur ^ 1]);
          }
        } else {
        const uint32_t maxs = maxs_last;                             // the last round
        for (uint32_t i0q = 0; i0q < nq; i0q += 1024) {
          const uint32_t idx = i0q + tid;
          bool pend = false;
          uint32_t e = 0, q = 0, dl = 0;
          if (idx < nq) {
            const uint32_t ent = (cur ? Qb : Qa)[idx];
            e = ent & 0x7FFFu; q = ent >> 16;
#ifdef ZADA_PL_STATS
            const uint32_t q_in = q; uint32_t hops = 0;
            if (lvl == 1) { uint32_t qq = q, stp = P[qq]; while (stp != 0 && e - (qq - stp) <= (uint32_t)MAX_DIST) { qq -= stp; hops++; if ((lb8(qq) & lmask) == (lb8(e) & lmask)) break; stp = P[qq]; } atomicAdd(&dbg[27], (unsigned long long)hops); atomicMax(&dbg[28], (unsigned long long)hops); if (hops > 64) atomicAdd(&dbg[29], 1ull); }
            (void)q_in;
#endif
            pend = !walk(e, q, lb8(e) & lmask, maxs, dl);
            if (pend && capped) { dl = DISTL_GAVEUP; pend = false; }
            if (!pend) plane[e] = (uint16_t)dl;
          }
          push(pend, e | (q << 16), cur ? Qa : Qb, &qn[cur ^ 1]);
        }
#ifdef ZADA_PL_STATS
        lds_barrier();
        if (tid == 0 && lvl == 1) atomicAdd(&dbg[31], clock64() - tq0);
#endif
        }
        lds_barrier();
      }
    }
    // (a full barrier, global memory included: the next segment of the run reads the tails tables this one has written)
    __syncthreads();
    PL_STAMP();
  }
  }   // the run's segments
}

// Cross-segment links: grid = (segments - 1, levels), block = 1024.  The workgroup of (segment s, level l) stages
// the tails table of segment s-1 (65 536 x u16: last position of every bucket) in LDS and links every
// position of segment s that has no predecessor inside its own segment to that tail.  Staging the table
// makes the random look-ups LDS reads; as 2-byte global gathers they fetched a whole line each.
// Cross-segment links: grid = (segments - 1, levels), block = 1024.  The workgroup of (segment s, level l) stages
// the tails table of segment s-1 (65 536 x u16: last position of every bucket) in LDS and links every
// position of segment s that has no predecessor inside its own segment to that tail.  Staging the table
// makes the random look-ups LDS reads; as 2-byte global gathers they fetched a whole line each.
//
// For the levels that have a plane (nearest match of exactly that many bytes) the same look-up settles most of the searches
// that did not end inside the segment: the chain of p's bucket goes on, in the previous segment, at that bucket's tail --
// the most recent position there with the same hash, which usually IS a match.  The previous segment's bytes are staged
// next to the table (32 KiB; the two fill the CU's LDS exactly) and the candidate is compared here; only a hash collision
// leaves the search to k_cross_dist, which would otherwise pay two dependent global round trips for each of them (a
// third of all positions).  (A/B, ZADA_CL_SPLIT: the table in 2 / 4 parts with as many workgroups per CU is 6 % / 45 %
// slower: every part re-reads the positions.)
constexpr int CL_LDS_PLANE = 131072 + 32768;
// (seg0: the launch takes the segments seg0 + 1 ..: one piece of an input that is still arriving, lz_shard)
__global__ void __launch_bounds__(1024) k_cross_links(const uint8_t *__restrict__ in, Layout L, LevelPtrs lv, DistPlanes dp, uint32_t seg_first, uint32_t seg_stride) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint16_t *tl = (uint16_t *)smem;                                 // 128 KiB
  const uint8_t *pb = smem + 131072;                               // the previous segment's bytes (levels with a plane)
  // (the segments seg_first, seg_first + seg_stride, ...: with runs of segments per workgroup in k_prev_links, the first segment of every run)
  const uint64_t seg = (uint64_t)blockIdx.x * seg_stride + seg_first, base = seg * 32768ull, pbase = base - 32768ull;
  const int l = blockIdx.y, tid = threadIdx.x;
  const bool has_plane = l + 1 < NLEVELS;
  if (lay_first(L, seg)) return;                                   // an entry's first segment has nothing before it
  const uint32_t m = lay_inserted(L, seg);
  const bool prev_first = lay_first(L, seg - 1);
#ifndef ZADA_OLD_INIT
  const uint32_t m_prev = lay_inserted(L, seg - 1);
#endif
  {
    const uint4 *src = (const uint4 *)(lv.tails[l] + (seg - 1) * 65536ull);
    for (int i = tid; i < 65536 / 8; i += 1024) ((uint4 *)tl)[i] = src[i];
#ifdef ZADA_OLD_INIT
    if (has_plane)
#endif
    {
      const uint4 *bs = (const uint4 *)(in + pbase);
      for (int i = tid; i < 32768 / 16; i += 1024) ((uint4 *)(smem + 131072))[i] = bs[i];
    }
  }
  __syncthreads();
#ifndef ZADA_OLD_INIT
  // Only the occupied buckets of the table were written by k_prev_links; the others hold whatever the memory held.  An entry t is
  // the bucket's tail iff position t of the previous segment was inserted and hashes to the bucket: if any inserted position does,
  // the bucket is occupied and its entry was written in this call.  (The last bytes of the segment are not all staged: global loads.)
  auto is_tail = [&](uint32_t t, uint32_t key) -> bool {
    if (t >= m_prev) return false;
    uint64_t v;
    if (t <= 32768u - 12u) v = lds_u64_at(pb, t); else v = load8(in, pbase + t);
    return hashL_of(v, 4 + l) == key;
  };
#endif
  uint16_t *prevl = lv.prev[l] + base;
  uint16_t *plane = has_plane ? dp.d[1 + l] + base : nullptr;
  const uint8_t *sin = in + base;
  const uint32_t cmask = 0xFFFFFFFFu;                              // (the only level with a plane, l = 0, compares four bytes)
  static_assert(NLEVELS <= 2, "the candidate compare of k_cross_links takes four bytes");
  for (uint32_t e0 = tid; e0 < m; e0 += 8192) {                    // 8 positions per lane in flight: the loop is latency bound
    uint32_t pv[8], pl[8];
    uint64_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint32_t e = e0 + 1024 * k; pv[k] = e < m ? (uint32_t)prevl[e] : 1u; pl[k] = (has_plane && e < m) ? (uint32_t)plane[e] : 0u; }
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint32_t e = e0 + 1024 * k; v[k] = (pv[k] == 0 || (pl[k] & DISTL_CONTINUE)) ? load8(sin, e) : 0ull; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const bool cont = (pl[k] & DISTL_CONTINUE) != 0;
      if (pv[k] != 0 && !cont) continue;
      const uint32_t e = e0 + 1024 * k;
      const uint32_t key = hashL_of(v[k], 4 + l);
      const uint32_t t = tl[key];
      const uint64_t q = pbase + t, d = base + e - q;
#ifdef ZADA_OLD_INIT
      const bool reach = t != 0xFFFFu && d <= (uint64_t)MAX_DIST && !(prev_first && t == 0);    // position 0 is never a match source (:467)
#else
      const bool reach = is_tail(t, key) && d <= (uint64_t)MAX_DIST && !(prev_first && t == 0); // position 0 is never a match source (:467)
#endif
      if (pv[k] == 0 && reach) prevl[e] = (uint16_t)d;
      if (cont) {
        // the search for the nearest position sharing 4 + l bytes goes on at the tail: none in reach -> none at all (older
        // members of the bucket are farther still); a true match -> settled; a hash collision -> left to k_cross_dist
        if (!reach) plane[e] = 0;
        else if (t <= 32768u - 4u) {
          const uint32_t theirs = lds_u32_tail(pb, t);                                // (t <= 32764: nothing behind the staged bytes is read)
          if (((theirs ^ (uint32_t)v[k]) & cmask) == 0) plane[e] = (uint16_t)d;
        }
      }
    }
  }
}

// Cross-segment continuation of the nearest 3 .. K-1 byte matches (runs after k_cross_links): one
// thread per inserted position of the segments >= 1; only positions marked "continue" do any work.
// (one wave per workgroup: a lane with a long bucket to scan only holds up its own wave)
#ifndef ZADA_CD_THREADS
#define ZADA_CD_THREADS 64
#endif
constexpr int CD_THREADS = ZADA_CD_THREADS;
#ifdef ZADA_CD_STATS
__device__ unsigned long long g_cd_dbg[16];
#endif
// Round 6: which four-byte values a segment holds, as a Bloom filter (2^17 bits = 16 KiB per segment, 0.5 byte per input byte; one bit per value).
// Nine in ten of k_cross_dist's level-4 walks find nothing (measured at 256 MiB: 8.7 M walks, 2.75 steps each, 8.6 % find a match): the value
// of a bucket's first member does not occur in the previous segment, but the bucket there is not empty -- a hash collision, and if it is with a
// frequent string the walk goes through all its occurrences, two 64-byte sectors a step, with the wave's other 63 lanes waiting.  A walk now
// asks the previous segment's filter first: a clear bit means the value is not there, and that IS the answer (no four-byte match).
constexpr uint32_t BLOOM_BITS = 1u << 17, BLOOM_WORDS = BLOOM_BITS / 32;
// (Two bits per value in one 64-byte block of the filter, measured: 9 % instead of 18 % of the absent values pass, the sweep 6.80 instead of 6.88 ms, the filters'
// build 0.62 instead of 0.46 ms -- the same sum; one bit it is.  Another multiplier than hashL_of's: a bucket's collisions spread over the filter.)
__device__ __forceinline__ uint32_t bloom4_of(uint32_t x) { return (x * 0x85EBCA6Bu) >> 15; }
__global__ void __launch_bounds__(256) k_bloom4(const uint8_t *__restrict__ in, Layout L, uint32_t *__restrict__ bloom, uint32_t seg0) {
  __shared__ uint32_t bits[BLOOM_WORDS];
  const uint64_t seg = (uint64_t)blockIdx.x + seg0, base = seg * 32768ull;
  const uint32_t m = lay_inserted(L, seg);
  const int tid = threadIdx.x;
  for (int i = tid; i < (int)BLOOM_WORDS; i += 256) bits[i] = 0;
  __syncthreads();
  const uint32_t *w = (const uint32_t *)(in + base);               // (the buffer is padded behind the input)
  for (uint32_t i = tid; i < 32768u / 4u; i += 256) {
    const uint32_t a = w[i], b = w[i + 1], e = 4u * i;
    const uint32_t x[4] = {a, __builtin_amdgcn_alignbyte(b, a, 1), __builtin_amdgcn_alignbyte(b, a, 2), __builtin_amdgcn_alignbyte(b, a, 3)};
#pragma unroll
    for (int j = 0; j < 4; j++) if (e + (uint32_t)j < m) { const uint32_t h = bloom4_of(x[j]); atomicOr(&bits[h >> 5], 1u << (h & 31u)); }
  }
  __syncthreads();
  uint4 *dst = (uint4 *)(bloom + seg * BLOOM_WORDS);
  for (int i = tid; i < (int)BLOOM_WORDS / 4; i += 256) dst[i] = ((const uint4 *)bits)[i];
}

// (Round 6, measured and dropped: the level-3 look-up and the level-4 walk of a position as two state machines side by side in the lane, their loads
// issued together in every turn, a batch's eight sorted-order entries fetched with its tags -- three round trips instead of four for the look-up,
// the longer of the two chains instead of their sum.  Bit-exact, and SLOWER: 16.0 against 12.7 ms for the phase at 1 GiB.  The kernel moves 64-byte
// sectors for two to eight useful bytes at 3.4 TB/s; what the entries' 16 bytes per batch add in sectors costs more than the shorter chain saves.)
// Round 6, the level-4 walks in two passes: the SWEEP (CD_SWEEP, a lane per position as before) asks the Bloom filter and walks at most CD_CAP steps;
// a walk that is still open then -- and a walk that k_prev_links gave up, which starts again from the position itself -- goes on a list (its plane keeps
// the marker), and CD_LIST takes the list with every lane busy and no limit: the long walks are a few per thousand, but in the sweep each of them held a
// wave of 64 positions for its whole length.  CD_ALL is the kernel as it was (no filter, no limit, no list): it follows the other two and does something
// only if the list has overflowed (the count stays on the device).
struct CdTail { const uint32_t *bloom; uint32_t *list; uint32_t *count; uint32_t cap; uint32_t *list2; uint32_t *count2; uint32_t cap2; };
enum { CD_SWEEP = 0, CD_LIST = 1, CD_ALL = 2 };
#ifndef ZADA_CD_CAP
#define ZADA_CD_CAP 16
#endif
#ifndef ZADA_CD_CAP2
#define ZADA_CD_CAP2 256
#endif
constexpr uint32_t CD_CAP = ZADA_CD_CAP, CD_CAP2 = ZADA_CD_CAP2;
template <int MODE>
__global__ void __launch_bounds__(CD_THREADS) k_cross_dist(const uint8_t *__restrict__ in, Layout L, LevelPtrs lv,
                                                    const uint16_t *__restrict__ S3, const uint8_t *__restrict__ T3, const uint32_t *__restrict__ bsc3,
                                                    DistPlanes dp, uint64_t p0, uint64_t p1, CdTail tl) {
  // (the sweep's waves in XCD order -- xcd_block(), every XCD a contiguous eighth of the stream, so that a previous segment's tables sit in one L2 -- measured: cross
  // phase 9.93 / 9.74 against 9.89 / 9.86 ms, nothing)
  uint64_t p = p0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;            // (the positions [p0, p1): all from 32 768 on, or one piece's)
  if (MODE == CD_LIST) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, nc = *tl.count, nl = nc < tl.cap ? nc : tl.cap;    // (what did not fit was walked in the sweep)
    if (i >= nl) return;
    p = tl.list[i];
  }
  if (p >= p1 || p >= L.n) return;
  const uint64_t seg = p >> 15, pbase = (seg - 1) * 32768ull;
  if (lay_first(L, seg) || (uint32_t)(p & 32767u) >= lay_inserted(L, seg)) return;
  const bool prev_first = lay_first(L, seg - 1);
  // Everything a position can need is loaded up front, coalesced and independent of each other (the kernel is a chain of
  // load latencies: looking at the markers first and loading afterwards was 12 % slower); then only the positions
  // k_prev_links / k_cross_links left a marker for go on.
  const uint32_t d3_stored = dp.d[0][p];
  uint32_t dl_first[NLEVELS > 1 ? NLEVELS - 1 : 1], link_first[NLEVELS > 1 ? NLEVELS - 1 : 1];
#pragma unroll
  for (int l = 0; l + 1 < NLEVELS; l++) { dl_first[l] = dp.d[1 + l][p]; link_first[l] = lv.prev[l][p]; }
  const uint64_t mine = *(const u64u *)(in + p);
  // (only where the head of the chain may lie exactly MAX_DIST back; p >= 32 768 > MAX_DIST)
  const uint32_t far24 = (d3_stored == DIST3_HEADCHK || d3_stored == DIST3_CONT_FIRST) ? *(const u32u *)(in + p - (uint64_t)MAX_DIST) & 0xFFFFFFu : 0xFFFFFFFFu;
  bool need = d3_stored >= DIST3_CONT_FIRST;
#ifdef ZADA_CD_SKIP3      /* timing experiments only (wrong results): the kernel without its level-3 / level >= 4 part */
  need = false;
#endif
#pragma unroll
  for (int l = 0; l + 1 < NLEVELS; l++) need = need || (dl_first[l] & DISTL_CONTINUE) || dl_first[l] == DISTL_GAVEUP;
#ifdef ZADA_CD_SKIP4
  need = d3_stored >= DIST3_CONT_FIRST;
#endif
  if (!need) return;
  // (the filter's bit is asked for here, so that it arrives while level 3 is looked up)
  uint32_t bloom_bit = 1u;
  if (MODE == CD_SWEEP && tl.bloom && NLEVELS > 1 && (dl_first[0] & DISTL_CONTINUE)) {
    const uint32_t hb = bloom4_of((uint32_t)mine);
    bloom_bit = (tl.bloom[(seg - 1) * BLOOM_WORDS + (hb >> 5)] >> (hb & 31u)) & 1u;
  }
  // level 3 first: the previous segment's bucket of the 15-bit hash, newest first, as far back as TOO_FAR (see k_prev_links)
  uint32_t dprev = d3_stored >= DIST3_CONT_FIRST ? 0u : d3_stored;
  const uint32_t my24 = (uint32_t)mine & 0xFFFFFFu;
  const uint32_t b0 = my24 & 0xFF, b1 = (my24 >> 8) & 0xFF;
  const uint32_t h = ((b0 << 10) ^ (b1 << 5) ^ (my24 >> 16)) & 0x7FFFu;
#ifdef ZADA_CD_NOL3   /* timing experiment only (wrong results): the kernel without its level-3 look-ups */
  if (false) {
#else
  if (d3_stored == DIST3_CONTINUE || d3_stored == DIST3_CONT_FIRST) {
#endif
    const uint32_t bsc = bsc3[pbase + h];
    const uint32_t pst = bsc & 0xFFFF, pct = bsc >> 16;
    const uint16_t *ps = S3 + pbase;
    const uint8_t *pt = T3 + pbase;
    const uint32_t mytag = (b0 >> 5) | ((b1 & 7u) << 3) | ((b0 & 3u) << 6);
    uint32_t d3 = 0;
    // newest first, eight tags per load: only candidates whose tag agrees are looked at (their bytes decide)
    const uint64_t tagx8 = 0x0101010101010101ull * mytag;
    for (uint32_t hi = pct; hi > 0 && d3 == 0;) {
      const uint32_t lo = hi >= 8 ? hi - 8 : 0;                      // candidates [lo, hi) of the bucket: byte i <-> candidate lo + i
      // the oldest candidate of this batch of eight already too far?  (positions ascend inside a bucket)
      const uint64_t x = *(const u64u *)(pt + pst + lo) ^ tagx8;
      // zero-byte detection, exact per byte
      uint64_t z = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);
      if (hi - lo < 8) z &= ~(~0ull << (8 * (hi - lo)));              // bytes past candidate hi-1 belong to the next bucket
      bool stop = false;
      while (z) {
        const int byte = 7 - (__builtin_clzll(z) >> 3);                // highest = newest
        z &= ~(0xFFull << (8 * byte));
        const uint32_t j = lo + (uint32_t)byte;                        // candidate index in the bucket
        const uint64_t q = pbase + ps[pst + j], d = p - q;
        if ((prev_first && q == pbase) || d > (uint64_t)TOO_FAR) { stop = true; break; }   // position 0 is never a match source (:467)
        if ((*(const u32u *)(in + q) & 0xFFFFFFu) == my24) { d3 = (uint32_t)d; break; }
      }
      if (stop) break;
      // nothing among these eight: go on only while the next older candidates can still be within TOO_FAR
      if (lo > 0 && p - (pbase + ps[pst + lo - 1]) > (uint64_t)TOO_FAR) break;
      hi = lo;
    }
    dprev = d3;
  }
  // The one candidate beyond TOO_FAR that matters: the reference accepts the HEAD of the 15-bit chain at a distance of
  // exactly MAX_DIST (:850), everything behind it only below (:820).  Here: the candidate lies in the previous segment
  // (own-segment heads are k_prev_links' first candidates), p has no same-hash predecessor in its own segment and the
  // candidate is the last member of its bucket.
  if (dprev == 0 && (d3_stored == DIST3_HEADCHK || d3_stored == DIST3_CONT_FIRST) && (p & 32767u) < (uint32_t)MAX_DIST) {
    const uint64_t q = p - (uint64_t)MAX_DIST;
    if (!(prev_first && q == pbase) && far24 == my24) {
      const uint32_t prv = bsc3[pbase + h];
      const bool q_last = (prv >> 16) != 0 && S3[pbase + (prv & 0xFFFF) + (prv >> 16) - 1] == (uint16_t)(q & 32767);
      if (q_last) dprev = (uint32_t)MAX_DIST;
    }
  }
  if (dprev != d3_stored) dp.d[0][p] = (uint16_t)dprev;
  // levels >= 4: follow the level's chain (it crosses into the previous segment after k_cross_links).
  // The levels are nested: an L byte match is never nearer than the nearest L-1 byte match (when that one is known:
  // level 3 only looks TOO_FAR back).
#pragma unroll
  for (int l = 0; l + 1 < NLEVELS; l++) {
    uint32_t dl = dl_first[l];
#ifdef ZADA_CD_NOL4   /* timing experiment only (wrong results): the kernel without its level-4 walks */
    if (false) {
#else
    if ((dl & DISTL_CONTINUE) || dl == DISTL_GAVEUP) {
#endif
      // the chain of p's bucket ended, inside p's segment, at q0 (the bucket's first member there): go on from its link
      // (a walk k_prev_links gave up: from p itself)
      const bool gaveup = dl == DISTL_GAVEUP;
      uint64_t q = gaveup ? p : p - (dl & 0x7FFFu);
      dl = 0;
      // (the plane of a listed position keeps its marker: CD_LIST starts the walk again; a full list: the walk goes on here)
      bool may_list = (MODE == CD_SWEEP && tl.list != nullptr) || (MODE == CD_LIST && tl.list2 != nullptr);
      uint32_t *const o_list = MODE == CD_LIST ? tl.list2 : tl.list, *const o_count = MODE == CD_LIST ? tl.count2 : tl.count;
      const uint32_t o_cap = MODE == CD_LIST ? tl.cap2 : tl.cap, step_cap = MODE == CD_LIST ? CD_CAP2 : CD_CAP;
      // (one reservation for all lanes that come here in the same turn: atomics on one address are served one after the other, chip-wide)
      auto to_list = [&]() -> bool {
        const unsigned long long act = __ballot(1);
        const int leader = __ffsll((long long)act) - 1, lane_ = (int)(threadIdx.x & 63);
        uint32_t base = 0;
        if (lane_ == leader) base = atomicAdd(o_count, (uint32_t)__popcll(act));
        base = (uint32_t)__shfl((int)base, leader);
        const uint32_t k = base + (uint32_t)__popcll(act & ((1ull << lane_) - 1ull));
        if (k < o_cap) { o_list[k] = (uint32_t)p; return true; }
        may_list = false; return false;
      };
      if (MODE == CD_SWEEP && may_list && gaveup && to_list()) return;   // (a walk through the position's own segment first: the filter says nothing about that)
      if (MODE == CD_SWEEP && l == 0 && !bloom_bit) { }                  // the value is not in the previous segment: no four-byte match
      else {
        const uint64_t mask = (1ull << (8 * (4 + l))) - 1ull;
        uint32_t d = q == p ? link_first[l] : (uint32_t)lv.prev[l][q];
        uint32_t nsteps = 0;
        for (;;) {
          if (d == 0) break;
          q -= d;
          if (p - q > (uint64_t)MAX_DIST) break;
          if (may_list && nsteps == step_cap && to_list()) return;
          const uint32_t dn = lv.prev[l][q];                            // the next link and this candidate's bytes in one round trip
          const uint64_t theirs = *(const u64u *)(in + q);
          nsteps++;
          if (p - q >= (uint64_t)dprev && ((theirs ^ mine) & mask) == 0) { dl = (uint32_t)(p - q); break; }
          d = dn;
        }
#ifdef ZADA_CD_STATS
        atomicAdd(&g_cd_dbg[0], 1ull); atomicAdd(&g_cd_dbg[1], (unsigned long long)nsteps); if (dl) atomicAdd(&g_cd_dbg[2], 1ull);
        if (nsteps == 0) atomicAdd(&g_cd_dbg[3], 1ull); if (nsteps > 8) atomicAdd(&g_cd_dbg[4], 1ull); if (nsteps > 64) atomicAdd(&g_cd_dbg[5], 1ull);
        if (dl == 0) atomicAdd(&g_cd_dbg[6], (unsigned long long)nsteps);
        if (dl_first[l] == DISTL_GAVEUP) atomicAdd(&g_cd_dbg[7], 1ull);
        if (nsteps > 32) atomicAdd(&g_cd_dbg[8], 1ull); if (nsteps > 128) atomicAdd(&g_cd_dbg[9], 1ull); if (nsteps > 256) atomicAdd(&g_cd_dbg[10], 1ull);
        if (nsteps > 512) atomicAdd(&g_cd_dbg[11], 1ull); if (nsteps > 1024) atomicAdd(&g_cd_dbg[12], 1ull); atomicMax(&g_cd_dbg[13], (unsigned long long)nsteps);
#endif
      }
      dp.d[1 + l][p] = (uint16_t)dl;
    }
    dprev = dl;
  }
}

// The longest level-4 walks (CD_LIST gave them up after CD_CAP2 steps: a bucket shared with a frequent string, walked one dependent load after the
// other -- the longest of a GiB, some two thousand steps, WAS the list pass: 2 ms for one lane).  What such a walk looks for is the nearest earlier
// position with the same four bytes, not farther than MAX_DIST: every such position is on the chain (same bytes, same bucket), and the first one the walk
// meets is the nearest.  So one WAVE reads the text backwards from the position, 64 positions at a time, eight loads in flight: at most 508 steps whose
// loads do not depend on each other (a lane takes four neighbouring candidates from one 8-byte load: sixteen turns for the whole window), instead of a chain of any length.  (The position's own segment is read too: for a walk that left it there is no
// match in it, and a walk k_prev_links gave up starts there anyway.)
__global__ void __launch_bounds__(256) k_cross_scan(const uint8_t *__restrict__ in, Layout L, DistPlanes dp, CdTail tl) {
  const uint32_t lane = threadIdx.x & 63, nc = *tl.count2, nl = nc < tl.cap2 ? nc : tl.cap2;
  for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < nl; i += (gridDim.x * blockDim.x) >> 6) {
    const uint64_t p = tl.list2[i], seg = p >> 15;
    // the oldest candidate: MAX_DIST back, not in front of the entry, and never the entry's first position (:467)
    uint64_t first = (seg - 1) << 15;                                       // (a listed position is not in an entry's first segment)
    if (lay_first(L, seg - 1)) first += 1;
    const uint64_t lo = p - (uint64_t)MAX_DIST > first ? p - (uint64_t)MAX_DIST : first;
    const uint32_t mine = *(const u32u *)(in + p);
    uint32_t found = 0;
    // a turn: the 2 048 candidates [hi - 2 048, hi), newest first, in eight groups of 256 -- a lane takes four neighbouring candidates from one 8-byte load
    for (uint64_t hi = p; hi > lo && !found; hi = hi - lo > 2048 ? hi - 2048 : lo) {
      uint64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint64_t q0 = hi - 256ull * (uint64_t)(u + 1) + 4ull * lane;         // (may wrap below zero: then it is not < hi)
        v[u] = (q0 < hi && q0 + 3 >= lo) ? *(const u64u *)(in + q0) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint64_t q0 = hi - 256ull * (uint64_t)(u + 1) + 4ull * lane;
        uint32_t mk = 0;
        if (q0 < hi && q0 + 3 >= lo) {
#pragma unroll
          for (int j = 0; j < 4; j++) if ((uint32_t)(v[u] >> (8 * j)) == mine && q0 + (uint64_t)j >= lo && q0 + (uint64_t)j < hi) mk |= 1u << j;
        }
        const unsigned long long m = __ballot(mk != 0);
        if (m && !found) {
          const int top = 63 - __builtin_clzll(m);                                  // the newest candidates are the highest lane's
          const uint32_t tm = RL(mk, top);
          const uint64_t q = hi - 256ull * (uint64_t)(u + 1) + 4ull * (uint64_t)top + (uint64_t)(31 - __builtin_clz(tm));
          found = (uint32_t)(p - q);
        }
      }
    }
    if (lane == 0) dp.d[1][p] = (uint16_t)found;
  }
}

// Chain-length limits as distances.  "Within the first k chain elements" <=> "not farther than the k-th
// predecessor in the 15-bit hash bucket" (own segment, then the previous one; anything older is out of
// range anyway).  Only buckets with at least kquarter elements over the two segments matter, so the
// per-position default (no limit) is written by k_prev_links and this kernel touches the few long
// buckets: one workgroup per segment, long buckets listed in LDS, then processed by all threads.
__global__ void __launch_bounds__(256) k_bucket_limits(Layout L, int kfull, int kquarter, const uint16_t *__restrict__ S3,
                                                       const uint32_t *__restrict__ bsc3, uint32_t *__restrict__ dlim, uint32_t *__restrict__ dlim_bits, const uint32_t *__restrict__ segmax,
                                                       const uint16_t *__restrict__ heavy, uint32_t seg0) {
  __shared__ uint32_t list[2048];
  __shared__ uint32_t nlist;
  __shared__ uint32_t lbits[1024];                   // the segment's share of dlim_bits (its 32 768 positions are its own 1 024 words), put together here
  const uint64_t seg = (uint64_t)blockIdx.x + seg0, base = seg * 32768ull;
  const int tid = threadIdx.x;
  const bool has_prev = !lay_first(L, seg);
  // a bucket matters only with at least kquarter members in this segment and the previous one together (see above): none
  // can have that many if the largest buckets of the two segments do not add up to it -- the usual case
  {
    const uint32_t mo = segmax[seg], mp = has_prev ? segmax[seg - 1] : 0u;
    if (mo == 0 || mo - 1 + mp < (uint32_t)kquarter) return;          // (the segment's words of dlim_bits stay as the memset left them: no limits)
  }
  for (int i = tid; i < 1024; i += 256) lbits[i] = 0;
  // (plain stores at the end: the words are this workgroup's alone, and 50 M scattered global atomics -- every twentieth position of the benchmark
  // stream has a limit -- cost the kernel 0.8 ms)
  auto flush_bits = [&]() { __syncthreads(); for (int i = tid; i < 1024; i += 256) { const uint32_t w = lbits[i]; if (w) dlim_bits[(base >> 5) + i] = w; } };
  const uint32_t *own = bsc3 + base, *prv = has_prev ? bsc3 + base - 32768 : nullptr;
  const uint16_t *s3 = S3 + base, *p3 = has_prev ? S3 + base - 32768 : nullptr;
  auto qualifies = [&](uint32_t h) -> bool {
    const uint32_t c = own[h] >> 16, cp = prv ? prv[h] >> 16 : 0u;
    return c > 0 && c - 1 + cp >= (uint32_t)kquarter;
  };
  auto process = [&]() {                             // the limits of the members of the listed buckets, all threads
    const uint32_t nl = nlist < 2048 ? nlist : 2048;
    for (uint32_t li = 0; li < nl; li++) {
      const uint32_t h = list[li];
      const uint32_t st = own[h] & 0xFFFF, c = own[h] >> 16;
      const uint32_t pst = prv ? prv[h] & 0xFFFF : 0u, cp = prv ? prv[h] >> 16 : 0u;
      const uint32_t r0 = (uint32_t)kquarter > cp ? (uint32_t)kquarter - cp : 0u;
      for (uint32_t r = r0 + tid; r < c; r += 256) {
        const uint32_t e = s3[st + r];
        uint32_t lim[2];
        const uint32_t ks[2] = {(uint32_t)kfull, (uint32_t)kquarter};
        for (int t = 0; t < 2; t++) {
          const uint32_t k = ks[t];
          uint64_t d = 0xFFFF;
          if (r >= k) d = e - s3[st + r - k];
          else if (cp >= k - r) d = (base + e) - (base - 32768 + p3[pst + cp - (k - r)]);
          lim[t] = d < 0xFFFF ? (uint32_t)d : 0xFFFFu;
        }
        dlim[base + e] = lim[0] | (lim[1] << 16);
        atomicOr(&lbits[e >> 5], 1u << (e & 31u));
      }
    }
  };
  // Candidates: a bucket with kquarter members over the two segments has half of them in one of the two, i.e. it is on this
  // segment's list of heavy buckets or on the previous one's (written by k_prev_links) -- a few dozen look-ups instead of a sweep
  // over the 2 x 32 768 bucket records.  (Lists that overflowed -- short chains, where nearly every bucket is heavy -- : the sweep.)
  const uint32_t heavy_thr = kquarter / 2 > 0 ? (uint32_t)kquarter / 2 : 1u;
  const uint16_t *ho = heavy + seg * HEAVY_STRIDE, *hp = has_prev ? heavy + (seg - 1) * HEAVY_STRIDE : nullptr;
  const uint32_t no = ho[0], np = hp ? hp[0] : 0u;
  if (no != 0xFFFFu && np != 0xFFFFu) {
    for (int phase = 0; phase < 2; phase++) {
      if (tid == 0) nlist = 0;
      __syncthreads();
      const uint16_t *hl = phase == 0 ? ho : hp;
      const uint32_t nh = phase == 0 ? no : np;
      for (uint32_t i = tid; i < nh; i += 256) {
        const uint32_t h = hl[1 + i];
        if (phase == 1 && (own[h] >> 16) >= heavy_thr) continue;          // (on this segment's own list: done in the first phase)
        if (qualifies(h)) { const uint32_t k = atomicAdd(&nlist, 1u); if (k < 2048) list[k] = h; }
      }
      __syncthreads();
      process();
      __syncthreads();
    }
    flush_bits();
    return;
  }
  for (uint32_t h0 = 0; h0 < 32768; h0 += 2048) {        // rounds of 2048 buckets: the list cannot overflow
    if (tid == 0) nlist = 0;
    __syncthreads();
    for (uint32_t h = h0 + tid; h < h0 + 2048; h += 256) {
      if (qualifies(h)) { const uint32_t k = atomicAdd(&nlist, 1u); if (k < 2048) list[k] = h; }
    }
    __syncthreads();
    process();
    __syncthreads();
  }
  flush_bits();
}

// --------------------------------------------------------------------------------------------
// k_match : Longest_Match for every position
// --------------------------------------------------------------------------------------------
// Block of MB positions [B, B+MB); LDS holds input bytes [WB, B+MB+272) and the chain links
// (distance to previous same-hash position) of [WB, B+MB), WB = B - HALO (clamped at 0).
#ifndef ZADA_FAST
#define ZADA_FAST 12
#endif
#ifndef ZADA_CMP_TURNS
#define ZADA_CMP_TURNS 2
#endif
static_assert(ZADA_FAST % 2 == 0, "the fast phase is unrolled in pairs of steps");
constexpr int MB = 16384;
constexpr int HALO = 32512;                       // >= MAX_DIST, multiple of 16
constexpr int WBYTES = HALO + MB + 272;           // 49168
constexpr int WLINKS = HALO + MB;                 // 48896
constexpr int MATCH_LDS = WBYTES + WLINKS * 2 + 16 + 16 * 128 * 8;   // window, links, work counter, one queue of start records per wave
static_assert(MATCH_LDS <= 160 * 1024, "k_match: one workgroup per CU");

#define LDS_U16(b, o) ((uint32_t)(b)[(o)] | ((uint32_t)(b)[(o) + 1] << 8))

// First pass of the match finder: a BOUNDED search for every position (see lz_stage: "demand driven").
// 16 KiB of positions per workgroup, window bytes and last-level links in LDS.  Persistent lanes: every lane owns
// one position at a time; a lane whose position is finished (or has used up its budget) takes the next one at
// once.  Each round = ZADA_FAST "fast" steps (follow the link, test the two bytes a candidate must share to beat
// `best`, :754-757) with no side paths, then one batched "slow" phase for everything rare: eight-byte compares of
// the survivors, improvements, limits, results.
__global__ void __launch_bounds__(1024) k_match(const uint8_t *__restrict__ in, Layout L,
                                                const uint16_t *__restrict__ prevd,
                                                DistPlanes dp,
                                                MatchPair *__restrict__ M,
                                                int nice_cfg, int budget, unsigned long long *__restrict__ dbg,
                                                uint16_t *__restrict__ resume, int short_budget) {
  // Every position of the block is searched for at most `budget` rounds of ZADA_FAST chain steps; a search cut
  // short leaves its best so far as a guess (M_GUESS) that k_match_demand replaces if a parse ever lands on it.
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *win = (uint32_t *)smem;                               // WBYTES bytes
  uint16_t *lnk = (uint16_t *)(smem + WBYTES);                    // WLINKS * 2 bytes
  uint32_t *next_pos = (uint32_t *)(smem + WBYTES + WLINKS * 2);     // work counter
#ifdef ZADA_MATCH_STATS
  unsigned long long t_start = clock64(), t_empty = 0, iters = 0;
#endif
  const uint64_t B = (uint64_t)xcd_block() * MB;
  const uint64_t WB = B >= (uint64_t)HALO ? B - HALO : 0;
  const uint32_t woff = (uint32_t)(B - WB);                        // window index of position B
  const uint64_t n = lay_end(L, B >> 15);                          // end of the stream / of the entry this block lies in
  const uint32_t cnt = n > B ? (uint32_t)((n - B) < (uint64_t)MB ? (n - B) : (uint64_t)MB) : 0u;
  const int tid = threadIdx.x;
  if (cnt == 0) return;
  // stage the window (the input buffer is padded with >= 512 zero bytes past n)
  {
    const uint32_t nb = woff + cnt + 272;                          // bytes to stage
    const uint4 *src = (const uint4 *)(in + WB);
    uint4 *dst = (uint4 *)win;
    for (uint32_t i = tid; i < (nb + 15) / 16; i += 1024) dst[i] = src[i];
    const uint32_t nl = woff + cnt;                                // links to stage (u16 each)
    const uint4 *ls = (const uint4 *)(prevd + WB);
    uint4 *ld = (uint4 *)lnk;
    // "no predecessor" (0) becomes the distance 0xFFFF: beyond every limit, so that the walk's limit test covers it
    auto nil2 = [](uint32_t x) -> uint32_t { return x | ((x & 0xFFFFu) ? 0u : 0xFFFFu) | ((x >> 16) ? 0u : 0xFFFF0000u); };
    for (uint32_t i = tid; i < (nl * 2 + 15) / 16; i += 1024) { uint4 v = ls[i]; v.x = nil2(v.x); v.y = nil2(v.y); v.z = nil2(v.z); v.w = nil2(v.w); ld[i] = v; }
    if (tid < 4) next_pos[tid] = 0;
  }
  __syncthreads();
  const uint8_t *win8 = (const uint8_t *)win;
  // Per-lane walker: state 0 = FREE (needs a position), 1 = WALK (fast filter steps), 2 = EVENT
  // (its current candidate passed the two-byte filter and/or its chain ended / hit a limit), 3 = EVENT whose
  // candidate is still being compared.
  // FAST PHASE: ZADA_FAST filter steps with no side paths; SLOW PHASE: everything rare, once per round.
  uint32_t wi = woff, cur = woff, ncur = woff, bdist = 0, rq = 0, kpos = 0, s_end = 0;
  int best = 2, la = 3, nice = 3, state = 0;
  uint32_t lim_cur = 0, lim_full = 0;
  bool have_q = false, exhausted = false;
  uint32_t cmp_off = 0;                                            // bytes already compared (state 3)
  int age = 0, mybudget = budget;                                  // rounds spent on the current position / allowed for it
  static_assert(NLEVELS == 2 && MB == 16384 && MAX_DIST < 32768, "packing of the queue entries");
  constexpr uint32_t QCAP = 128;                                   // entries per wave: fewer than 64 left before a refill of at most 64
  uint32_t *queue = next_pos + 4 + (threadIdx.x >> 6) * (2 * QCAP);
  const int lane = threadIdx.x & 63;
  uint32_t q_head = 0, q_tail = 0;                                 // (uniform over the wave)
  bool blk_done = false;                                           // the block has no positions left to take
  for (;;) {
    // ---- fetch ----
    // Positions are taken 64 at a time by the whole wave: every lane works out how one position's search starts
    // (all lanes busy, instead of the few that happen to be free), the positions without a chain to walk are
    // finished on the spot, and the others wait, packed in 8 bytes, in a queue of the wave in LDS from which free
    // lanes pick them up.
    const bool need = (state == 0) && !exhausted;
    const unsigned long long need_mk = __ballot(need);
    if (need_mk) {
      const uint32_t nneed = (uint32_t)__popcll(need_mk);
      if (nneed > q_tail - q_head && !blk_done) {                  // (at most one refill per round)
        uint32_t k0 = 0;
        if (lane == 0) k0 = atomicAdd(&next_pos[0], 64u);
        k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k0);
        if (k0 >= cnt) blk_done = true;
        else {
          const uint32_t k = k0 + (uint32_t)lane;
          bool ok = false;
          uint32_t e_lo = 0, e_hi = 0;
          if (k < cnt) {
            const uint64_t rem = n - (B + k);
            const int la_ = rem < 258 ? (int)rem : 258;            // Longest_Match never returns more
            const int nice_ = nice_cfg < la_ ? nice_cfg : la_;     // lz77.adb:858-860
            // limits of this position's walk, as distances, and the nearest 3 .. K-1 byte matches (k_prev_links)
            const uint32_t dlimv = dp.limits(B + k);
            const uint32_t df = dlimv & 0xFFFF, dq = dlimv >> 16;
            uint32_t dl[NLEVELS];
#pragma unroll
            for (int l = 0; l < NLEVELS; l++) dl[l] = dp.d[l][B + k];
            const uint32_t lf = dl[0] == (uint32_t)MAX_DIST ? (uint32_t)MAX_DIST : (df < (uint32_t)(MAX_DIST - 1) ? df : (uint32_t)(MAX_DIST - 1));   // :850 / :820-822
            const uint32_t lq = dq < lf ? dq : lf;                                                       // :733-735
            // best candidate of each length 3 .. K-1 = the nearest position sharing that many bytes; the
            // levels are nested (a 4-byte match is a 3-byte match), so they are valid in order
            int b_ = 2; uint32_t bd_ = 0, qbest = 0, qlev = 0;
            bool chain_ok = true;
            // (every level on its own: plane d[0] only knows three-byte matches up to TOO_FAR back, so "no three-byte
            // match" there does not mean "no four-byte match"; a valid level l implies the levels below it)
#pragma unroll
            for (int l = 0; l < NLEVELS; l++) {
              const bool v = la_ >= 3 + l && dl[l] != 0 && dl[l] <= lf;
              if (v) { b_ = 3 + l; bd_ = dl[l]; if (dl[l] <= lq) { qbest = ((uint32_t)(3 + l) << 16) | dl[l]; qlev = 1u + l; } }
              chain_ok = v;
            }
            // chain_ok: a candidate of length K-1 exists, so longer ones may: walk the level-K chain
            const bool hq = chain_ok && bd_ > lq;
            const uint32_t d0 = lnk[woff + k];                       // (0xFFFF = none)
            ok = chain_ok && b_ < nice_ && d0 <= lf;
            if (!ok) {
              const uint32_t packed = b_ >= 3 ? ((uint32_t)b_ << 16) | bd_ : 0u;
              // quarter-chain result: the best level whose candidate lies within the quarter limit
              MatchPair r; r.full = packed; r.quarter = (hq || !chain_ok) ? qbest : packed;
              M[B + k] = r;
            }
            // Positions deep inside a match (the nearest four-byte match lies at the same distance as for the two positions
            // before) are hardly ever looked at by the parser -- and hold most of the chain steps: they get a single round.
            // The result does not depend on the budget (a parse that does land there demands the exact search).
            bool inner = false;
#ifndef ZADA_NO_INNER
            if (ok && k >= 2 && short_budget > 0) { const uint32_t m1 = dp.d[NLEVELS - 1][B + k - 1], m2 = dp.d[NLEVELS - 1][B + k - 2]; inner = dl[NLEVELS - 1] == m1 && m1 == m2; }
#endif
            // (hq is recomputed at pick-up: for a queued position it is bd_ > lq)
            e_lo = k | ((uint32_t)(b_ - 2) << 14) | ((uint32_t)inner << 16) | (bd_ << 17);
            e_hi = lf | (lq << 15) | (qlev << 30);
          }
          const unsigned long long okm = __ballot(ok);
          if (ok) { uint32_t *e = queue + 2 * ((q_tail + (uint32_t)__popcll(okm & ((1ull << lane) - 1ull))) & (QCAP - 1)); e[0] = e_lo; e[1] = e_hi; }
          q_tail += (uint32_t)__popcll(okm);
        }
      }
      // free lanes take the queue's entries in lane order
      const uint32_t avail = q_tail - q_head, rank = (uint32_t)__popcll(need_mk & ((1ull << lane) - 1ull));
      if (need) {
        if (rank < avail) {
          const uint32_t *e = queue + 2 * ((q_head + rank) & (QCAP - 1));
          const uint32_t e_lo = e[0], e_hi = e[1];
          kpos = e_lo & 0x3FFFu; wi = woff + kpos; age = 0;
          const uint64_t rem = n - (B + kpos);
          la = rem < 258 ? (int)rem : 258;
          nice = nice_cfg < la ? nice_cfg : la;
          best = 2 + (int)((e_lo >> 14) & 3u); bdist = e_lo >> 17;
          mybudget = ((e_lo >> 16) & 1u) ? short_budget : budget;
          lim_full = e_hi & 0x7FFFu;
          const uint32_t lim_q = (e_hi >> 15) & 0x7FFFu, qlev = e_hi >> 30;
          have_q = bdist > lim_q;
          // quarter-chain result so far: the best level whose candidate lies within the quarter limit
          rq = 0;
          if (qlev != 0) {
            const uint32_t qd = (int)qlev + 2 == best ? bdist : (uint32_t)dp.d[0][B + kpos];   // (level 3 within the limit, level 4 not: rare)
            rq = ((qlev + 2u) << 16) | qd;
          }
          lim_cur = have_q ? lim_full : lim_q;
          const uint32_t d0 = lnk[wi];
          if (!have_q && d0 > lim_q) { have_q = true; rq = ((uint32_t)best << 16) | bdist; lim_cur = lim_full; }
          cur = wi - d0;
          s_end = LDS_U16(win8, wi + (uint32_t)best - 1u);
          state = 1;
        } else if (blk_done) {
#ifdef ZADA_MATCH_STATS
          if (t_empty == 0) t_empty = clock64();
#endif
          exhausted = true;
        }
      }
      q_head += nneed < avail ? nneed : avail;
    }
    if (!__any(state != 0)) { if (__all(exhausted)) break; continue; }
    // ---- fast phase ----
    // A step: the candidate's link and the two bytes at best-1, best (:754-755); the walk stops at a candidate
    // that passes the filter, or whose successor lies beyond the limit (which includes "no successor").  Six vector
    // instructions: the current and the next candidate swap registers from step to step instead of being copied,
    // and what stopped the walk is worked out afterwards, in the slow phase.
    {
      bool walk = state == 1, odd_stop = false;
      typedef const __attribute__((address_space(3))) uint8_t *lds_bytes;
      const uint32_t off = (uint32_t)(uintptr_t)(lds_bytes)win8 + (uint32_t)best - 1u;   // LDS address of byte best-1 of candidate 0
      const int lim_pos = (int)wi - (int)lim_cur;                  // candidates below this index are too far (:819-822)
      uint32_t ca = cur, cb = ncur;
#pragma unroll
      for (int it = 0; it < ZADA_FAST; it += 2) {
        if (walk) {
#ifdef ZADA_MATCH_STATS
          iters++;
#endif
          const uint32_t dn = lnk[ca];
          uint32_t aa = ca + off;
          asm("" : "+v"(aa));                                     // (keeps the + 1 of the second byte in the instruction's offset field)
          const lds_bytes pa = (lds_bytes)(uintptr_t)aa;
          const uint32_t c16 = (uint32_t)pa[0] | ((uint32_t)pa[1] << 8);
          cb = ca - dn;
          walk = !(c16 == s_end || (int)cb < lim_pos);
        }
        if (walk) {
#ifdef ZADA_MATCH_STATS
          iters++;
#endif
          const uint32_t dn = lnk[cb];
          uint32_t ab = cb + off;
          asm("" : "+v"(ab));
          const lds_bytes pb = (lds_bytes)(uintptr_t)ab;
          const uint32_t c16 = (uint32_t)pb[0] | ((uint32_t)pb[1] << 8);
          ca = cb - dn;
          walk = !(c16 == s_end || (int)ca < lim_pos);
          odd_stop = !walk;
        }
      }
      // a walk that stopped in an odd step has its candidate in cb and the successor in ca
      cur = odd_stop ? cb : ca;
      ncur = odd_stop ? ca : cb;
      if (state == 1 && !walk) state = 2;
    }
    // ---- slow phase ----
    if (__any(state >= 2)) {
      // what stopped the walk
      bool ev_pass = false, ev_end = false, ev_lim = false;
      if (state >= 2) {
        ev_pass = LDS_U16(win8, cur + (uint32_t)best - 1u) == s_end;   // bytes best-1, best agree (:754-755)
        ev_end = lnk[cur] == 0xFFFFu;                                   // chain exhausted
        ev_lim = (int)ncur < (int)wi - (int)lim_cur;                    // next candidate beyond the quarter / full limit (:819-822)
      }
      // The compare takes at most two turns of eight bytes in a round; a longer one goes on in the next rounds (state 3,
      // `cmp_off` bytes done) instead of keeping the other lanes of the wave waiting.
      bool cmpa = (state >= 2) && ev_pass;
      uint32_t off = state == 3 ? cmp_off : 0u;
      int len = 0;
#pragma unroll
      for (int t = 0; t < ZADA_CMP_TURNS; t++) {
        if (cmpa) {
#ifdef ZADA_MATCH_STATS
          iters++;
#endif
          const uint64_t x = lds_u64_at(win8, cur + off) ^ lds_u64_at(win8, wi + off);
          if (x) { len = (int)off + (int)(__builtin_ctzll(x) >> 3); cmpa = false; }
          else { off += 8; if ((int)off >= la) { len = la; cmpa = false; } }
        }
      }
      if (cmpa) { cmp_off = off; state = 3; }
      else if (state == 3) state = 2;
      if (state == 2) {
        len = len < la ? len : la;
        const bool improved = ev_pass && len > best;               // :812-817
        if (improved) {
          best = len; bdist = wi - cur;
          const uint32_t a = wi + (uint32_t)best;
          s_end = LDS_U16(win8, a - 1);
        }
        bool fin = (improved && len >= nice) || ev_end;            // :815, :820
        const uint32_t packed = best >= 3 ? ((uint32_t)best << 16) | bdist : 0u;
        if (!fin && ev_lim) {
          if (!have_q) { have_q = true; rq = packed; lim_cur = lim_full; }                 // quarter-chain result (:733-735)
          if (wi - ncur > lim_full) fin = true;                    // :819-822
        }
        if (fin) {
          MatchPair r; r.full = packed; r.quarter = have_q ? rq : packed;
          M[B + kpos] = r;
          state = 0;
        } else {
          cur = ncur;
          state = 1;
        }
      }
    }
    // first pass: a search that has had its share of rounds is cut short, its best so far becomes a guess
    if (state == 1 && ++age >= mybudget) {
      const uint32_t packed = best >= 3 ? ((uint32_t)best << 16) | bdist : 0u;
      // (k_match_demand takes the search up where it stops here: the next candidate, as a distance, goes with the guess)
      MatchPair r; r.full = packed | M_GUESS | (have_q ? M_HAVEQ : 0u); r.quarter = (have_q ? rq : packed) | (M_BEAT_MAX << M_BEAT_SHIFT);   // (length to beat: none asked yet)
      M[B + kpos] = r;
      resume[B + kpos] = (uint16_t)(wi - cur);
      state = 0;
    }
  }
#ifdef ZADA_MATCH_STATS
  {
    unsigned long long t_end = clock64();
    atomicAdd(&dbg[0], iters);                                   // lane-iterations
    atomicMax(&dbg[1], t_end - t_start);                         // longest lane lifetime (cycles)
    if ((threadIdx.x & 63) == 0) { atomicAdd(&dbg[2], t_end - t_start); atomicAdd(&dbg[3], 1ull); atomicAdd(&dbg[4], t_end - t_empty); }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(&dbg[5], clock64() - t_start); atomicAdd(&dbg[6], 1ull); }
  }
#endif
}

// --------------------------------------------------------------------------------------------
// Demand pass: the exact Longest_Match of the positions a parse has landed on while they only held a guess.
// These are few and their chains are the longest, so a lane-per-position walk would leave the machine idle
// behind a handful of 4096-step pointer chases.  Instead one WAVE takes a position and scans its candidates 64 at a
// time: the members of a bucket of the last level are a contiguous run of the segment's sorted order (RunPtrs),
// nearest first when read backwards, in the position's own segment and then in the previous one - no links.
// The sequential rule "a candidate replaces the best only if strictly longer, stop at nice_match" (:812-817)
// becomes, per batch of 64: the maximum length, at its nearest occurrence, or the nearest one reaching
// nice_match.  The quarter-chain snapshot (:733-735) is taken when the batch crosses the quarter distance.
// --------------------------------------------------------------------------------------------
#ifndef ZADA_DM_THREADS
#define ZADA_DM_THREADS 512
#endif
static_assert(NLEVELS >= 2, "the links of level 0 are dead after k_cross_dist: their array carries the resume points of the guesses");
#ifndef ZADA_DM_AHEAD
#define ZADA_DM_AHEAD 1
#endif
#ifndef ZADA_DMB
#define ZADA_DMB 4096
#endif
constexpr int DM_THREADS = ZADA_DM_THREADS, DMB = ZADA_DMB, DM_SLICE = 256, DM_AHEAD = ZADA_DM_AHEAD;
constexpr int DM_WBYTES = HALO + DMB + 272;
#ifdef ZADA_DM_STATS
__device__ unsigned long long g_dm_dbg[8];
#endif
struct ScanDesc {                                  // a position whose candidates have to be scanned (32 bytes, in LDS)
  uint16_t k, la, idx1, c1, idx2, c2, lim_full, lim_q;
  uint16_t bdist, best_hq;                         // best | reach << 14 | have_q << 15
  uint32_t rq, og_full, og_quarter;
};
constexpr int DM_LDS = DM_WBYTES + DMB * 2 + DM_SLICE * (int)sizeof(ScanDesc) + 64;
static_assert(MB % DMB == 0 && DM_WBYTES % 16 == 0 && sizeof(ScanDesc) == 32, "demand blocks tile the first-pass blocks");
// (six waves per SIMD, i.e. three workgroups per CU: without the bound the compiler takes 85 registers and only two fit)
__global__ void __launch_bounds__(DM_THREADS, 6) k_match_demand(const uint8_t *__restrict__ in, Layout L, DistPlanes dp, RunPtrs rp,
                                                             const uint16_t *__restrict__ tailsK, MatchPair *__restrict__ M, int nice_cfg,
                                                             const uint32_t *__restrict__ blk_demand, uint32_t *__restrict__ dbits, uint8_t *__restrict__ chg,
                                                             const ExitState *__restrict__ spec_exits, const uint16_t *__restrict__ resume, bool use_beat) {
  if (blk_demand[xcd_block()] == 0) return;
#ifdef ZADA_DM_STATS   /* cycles of a block's phases (thread 0's clock at the barriers that are there anyway): staging, list, phase A, phase B */
  unsigned long long dm_t = clock64();
#define DM_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_dm_dbg[k], t_ - dm_t); dm_t = t_; } } while (0)
#else
#define DM_STAMP(k) do {} while (0)
#endif
  const uint64_t B = (uint64_t)xcd_block() * DMB;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *win = (uint32_t *)smem;                               // DM_WBYTES bytes
  uint16_t *list = (uint16_t *)(smem + DM_WBYTES);                // the marked positions of the block, up to DMB
  ScanDesc *desc = (ScanDesc *)(smem + DM_WBYTES + DMB * 2);      // DM_SLICE entries
  uint32_t *ctr = (uint32_t *)(smem + DM_WBYTES + DMB * 2 + DM_SLICE * sizeof(ScanDesc));
  const uint64_t WB = B >= (uint64_t)HALO ? B - HALO : 0;
  const uint32_t woff = (uint32_t)(B - WB);
  const uint64_t n = lay_end(L, B >> 15);
  const uint32_t cnt = n > B ? (uint32_t)((n - B) < (uint64_t)DMB ? (n - B) : (uint64_t)DMB) : 0u;
  const bool seg_first = lay_first(L, B >> 15), prev_first = (B >> 15) > 0 && lay_first(L, (B >> 15) - 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Round 6: the window is staged BESIDE phase A of the first slice, by the threads phase A does not use (it has one lane per marked position, 256 at
  // most, and is a chain of dependent loads of its own; the two hashes it took from the window come from memory there).  -DZADA_DM_STAGE_FIRST: all
  // threads stage first, as before.
#ifdef ZADA_DM_STAGE_FIRST
  constexpr bool STAGE_BESIDE = false;
#else
  constexpr bool STAGE_BESIDE = DM_THREADS > DM_SLICE;
#endif
  auto stage = [&](uint32_t t0, uint32_t stride) {
    const uint32_t nb = woff + cnt + 272;
    const uint4 *src = (const uint4 *)(in + WB);
    uint4 *dst = (uint4 *)win;
    for (uint32_t i = t0; i < (nb + 15) / 16; i += stride) dst[i] = src[i];
  };
  if (!STAGE_BESIDE) stage((uint32_t)tid, DM_THREADS);
  const uint8_t *win8 = (const uint8_t *)win;
  // a changed value: the speculative parses that used the guess have to be redone (the chunk's own, and the previous
  // chunk's if it ran over into this position)
  // `reach`: the previous chunk's speculative parse ran over into this position (looked up in phase A, where it costs
  // nothing: here it would be a load on the wave's critical path)
  auto store_result = [&](uint64_t p, uint32_t full, uint32_t quarter, uint32_t og_full, uint32_t og_quarter, bool reach) {
    MatchPair r; r.full = full; r.quarter = quarter;
    M[p] = r;
    const uint32_t beat = use_beat ? og_quarter >> M_BEAT_SHIFT : 0u, ogq = og_quarter & M_VALUE;      // (the first parse leaves no length to beat: every change counts)
    if ((og_full & M_BYSPEC) && ((full != (og_full & M_VALUE) && (full >> 16) > beat) || (quarter != ogq && (quarter >> 16) > beat))) {
      const uint64_t ch = p / PCHUNK;
      chg[ch] = 1;
      if (reach) chg[ch - 1] = 1;
    }
  };
  if (tid == 0) ctr[0] = 0;
  __syncthreads();
  DM_STAMP(0);
  // the marked positions of the block, from the bit map the parsers keep (one bit per position: 512 bytes instead of the
  // block's 32 KB of match records); the block's bits are cleared for the next round
  if (tid < DMB / 32) {
    uint32_t *wp = dbits + (B >> 5) + tid;
    uint32_t bits = *wp;
    if (bits) {
      *wp = 0;
      uint32_t o = atomicAdd(&ctr[0], (uint32_t)__popc(bits));
      while (bits) { const int j = __ffs((int)bits) - 1; bits &= bits - 1; list[o++] = (uint16_t)(tid * 32 + j); }
    }
  }
  __syncthreads();
  DM_STAMP(1);
  const uint32_t nl = ctr[0];
#ifdef ZADA_DM_STATS
  if (tid == 0) { atomicAdd(&g_dm_dbg[4], 1ull); atomicAdd(&g_dm_dbg[5], (unsigned long long)nl); }
#endif
  for (uint32_t l0 = 0; l0 < nl; l0 += DM_SLICE) {
    if (tid == 0) { ctr[1] = 0; ctr[2] = DM_THREADS / 64; }
    __syncthreads();
    // ---- phase A, one lane per marked position: everything that is known about the position (limits, nearest
    //      3..K-1 byte matches, where its bucket lies in the sorted orders) and the start of the search.  Positions
    //      with nothing to scan are finished here; the others leave a descriptor in LDS. ----
    if (STAGE_BESIDE && l0 == 0 && tid >= DM_SLICE) stage((uint32_t)(tid - DM_SLICE), DM_THREADS - DM_SLICE);
    const bool win_ready = !(STAGE_BESIDE && l0 == 0);               // (the first slice's phase A runs while the window is on its way)
    if ((uint32_t)tid < (uint32_t)DM_SLICE && l0 + tid < nl) {
      const uint32_t k = list[l0 + tid];
      const uint64_t p = B + k;
      const uint32_t wi = woff + k;
      const uint64_t rem = n - p;
      const int la = rem < 258 ? (int)rem : 258;                   // Longest_Match never returns more
      const int nice = nice_cfg < la ? nice_cfg : la;              // lz77.adb:858-860
      const uint64_t seg = p >> 15, pbase = (seg << 15) - 32768;
      const uint32_t dlimv = dp.limits(p);
      uint32_t dl[NLEVELS];
#pragma unroll
      for (int l = 0; l < NLEVELS; l++) dl[l] = dp.d[l][p];
      const MatchPair og = M[p];
      const uint32_t idx1 = rp.idx[p], c1 = rp.cnt[p];
      uint32_t t = 0xFFFFu;
      if (!seg_first) {
        // (only the occupied buckets of a tails table are written: an entry is the bucket's tail iff it names an inserted position of
        // the previous segment that hashes to the bucket -- see k_cross_links; a tail out of reach leaves nothing to scan there)
        const uint32_t key = hashL_of(win_ready ? lds_u64_at(win8, wi) : *(const u64u *)(in + p), 3 + NLEVELS);
        const uint32_t tt = tailsK[(seg - 1) * 65536ull + key];
        const uint64_t q = pbase + tt;
        if (tt < lay_inserted(L, seg - 1) && p - q <= (uint64_t)MAX_DIST &&
            hashL_of(win_ready ? lds_u64_at(win8, (uint32_t)(q - WB)) : *(const u64u *)(in + q), 3 + NLEVELS) == key) t = tt;
      }
      uint32_t idx2 = 0, c2 = 0;
      if (t != 0xFFFFu) { idx2 = rp.idx[pbase + t]; c2 = (uint32_t)rp.cnt[pbase + t] + 1u; }
      const uint32_t df = dlimv & 0xFFFF, dq = dlimv >> 16;
      const uint32_t lim_full = dl[0] == (uint32_t)MAX_DIST ? (uint32_t)MAX_DIST : (df < (uint32_t)(MAX_DIST - 1) ? df : (uint32_t)(MAX_DIST - 1));   // :850 / :820-822
      const uint32_t lim_q = dq < lim_full ? dq : lim_full;                                                                                               // :733-735
      int best = 2;
      uint32_t bdist = 0, qbest = 0;
      bool chain_ok = true;
#pragma unroll
      for (int l = 0; l < NLEVELS; l++) {                            // (levels on their own: see k_match)
        const bool v = la >= 3 + l && dl[l] != 0 && dl[l] <= lim_full;
        if (v) { best = 3 + l; bdist = dl[l]; if (dl[l] <= lim_q) qbest = ((uint32_t)(3 + l) << 16) | dl[l]; }
        chain_ok = v;
      }
      bool have_q = chain_ok && bdist > lim_q;
      // The search was begun by k_match: its best so far is the guess, and `resume` the candidate it had come to.
      // The candidates before that one (nearer) are dropped from the two runs.
      uint32_t idx1r = idx1, c1r = c1, idx2r = idx2, c2r = c2;
      if (chain_ok) {
        const uint32_t gl = (og.full & M_VALUE) >> 16;
        if (gl >= 3) { best = (int)gl; bdist = og.full & 0xFFFFu; }
        have_q = (og.full & M_HAVEQ) != 0;
        if (have_q) qbest = og.quarter & M_VALUE;
        const uint64_t q = p - resume[p];
        const uint32_t iq = rp.idx[q];
        const uint32_t skip = q >= (seg << 15) ? idx1 - 1 - iq : c1 + (idx2 - iq);   // candidates nearer than q
        if (skip < c1) { idx1r = idx1 - skip; c1r = c1 - skip; }
        else { const uint32_t s2 = skip - c1 < c2 ? skip - c1 : c2; c1r = 0; idx2r = idx2 - s2; c2r = c2 - s2; }
      }
      const uint64_t chk = p / PCHUNK;
      const bool reach = chk > 0 && spec_exits[chk - 1].pos >= (uint32_t)p;
      if (chain_ok && best < nice && c1r + c2r > 0) {
        ScanDesc ds;
        ds.k = (uint16_t)k; ds.la = (uint16_t)la; ds.idx1 = (uint16_t)idx1r; ds.c1 = (uint16_t)c1r; ds.idx2 = (uint16_t)idx2r; ds.c2 = (uint16_t)c2r;
        ds.lim_full = (uint16_t)lim_full; ds.lim_q = (uint16_t)lim_q; ds.bdist = (uint16_t)bdist; ds.best_hq = (uint16_t)((uint32_t)best | ((uint32_t)reach << 14) | ((uint32_t)have_q << 15));
        ds.rq = qbest; ds.og_full = og.full; ds.og_quarter = og.quarter;
        desc[atomicAdd(&ctr[1], 1u)] = ds;
      } else {
        const uint32_t packed = best >= 3 ? ((uint32_t)best << 16) | bdist : 0u;
        store_result(p, packed, !chain_ok ? qbest : (have_q ? qbest : packed), og.full, og.quarter, reach);
      }
    }
    __syncthreads();
    DM_STAMP(2);
#ifdef ZADA_DM_STATS
    if (tid == 0) atomicAdd(&g_dm_dbg[6], (unsigned long long)ctr[1]);
#endif
    // ---- phase B, one wave per descriptor: scan the candidates, 64 per batch.  The members of the bucket are a
    //      contiguous run of the sorted order, nearest first when read backwards, in the position's own segment
    //      and then in the previous one.  The first batch of the wave's next descriptor is fetched meanwhile. ----
    const uint32_t ns = ctr[1];
    // candidate c of a descriptor's position (nearest first), as a distance; 0 = no such candidate.  Position 0 is never
    // a match source (:467) and ends the chain.
    // The load and its use are kept apart (cand_load returns what the sorted order holds, cand_dist turns it into the
    // distance): the loads of several batches are then in flight together instead of each being waited for.  A lane
    // without a candidate reads the segment's first entry, so that there is no branch around the load.
    auto cand_load = [&](const ScanDesc &ds, uint32_t c) -> uint32_t {
      // (32-bit offsets from the start of the previous segment in the sorted order)
      const uint16_t *sprev = rp.S + ((((B + ds.k) >> 15) << 15) - 32768);
      const bool in1 = c < ds.c1, in2 = !in1 && c - ds.c1 < ds.c2;
      const uint32_t a = in1 ? 32768u + ds.idx1 - 1u - c : (in2 ? (uint32_t)ds.idx2 - (c - ds.c1) : 32768u);
      return sprev[a];
    };
    auto cand_dist = [&](const ScanDesc &ds, uint32_t c, uint32_t raw) -> uint32_t {
      const uint64_t P_ = B + ds.k;
      const uint32_t po = (uint32_t)P_ & 32767u;                                // offset in the segment
      const bool in1 = c < ds.c1, in2 = !in1 && c - ds.c1 < ds.c2;
      const bool q0 = raw == 0 && ((in1 && seg_first) || (in2 && prev_first));   // the candidate is position 0 of the stream
      return q0 ? 0u : (in1 ? po - raw : (in2 ? po + 32768u - raw : 0u));
    };
#ifndef ZADA_DM_STATIC
    // (round 6) the descriptors are handed out through a counter instead of wave by wave in turn: a scan is one batch or sixty, and a wave that drew the long ones
    // kept the other seven waiting at the slice's barrier (parse phase 30.75 -> 30.08 ms per GiB; per block of the first demand pass -- 120 scans --
    // staging 5 k, list 7 k, phase A 11 k, phase B 78 k cycles: -DZADA_DM_STATS)
    auto take = [&]() -> uint32_t { uint32_t t = 0; if (lane == 0) t = atomicAdd(&ctr[2], 1u); return (uint32_t)__builtin_amdgcn_readfirstlane((int)t); };
    uint32_t si = (uint32_t)wave, nxt = si < ns ? take() : ns;
    uint32_t rnext = si < ns ? cand_load(desc[si], (uint32_t)lane) : 0u;
    for (; si < ns; si = nxt, nxt = si < ns ? take() : ns) {
      const ScanDesc ds = desc[si];
      const uint32_t r0 = rnext;
      if (nxt < ns) rnext = cand_load(desc[nxt], (uint32_t)lane);
#else
    uint32_t rnext = (uint32_t)wave < ns ? cand_load(desc[wave], (uint32_t)lane) : 0u;
    for (uint32_t si = (uint32_t)wave; si < ns; si += DM_THREADS / 64) {
      const ScanDesc ds = desc[si];
      const uint32_t r0 = rnext;
      if (si + DM_THREADS / 64 < ns) rnext = cand_load(desc[si + DM_THREADS / 64], (uint32_t)lane);
#endif
      const uint32_t WI = woff + ds.k, LF = ds.lim_full, LQ = ds.lim_q, TT = (uint32_t)ds.c1 + ds.c2;
      const int LA = ds.la, NICE = nice_cfg < LA ? nice_cfg : LA;
      int bst = ds.best_hq & 0x3FFF;
      uint32_t bd = ds.bdist, rqq = ds.rq;
      bool hq = (ds.best_hq >> 15) != 0;
      auto lds_u32 = [&](uint32_t o) -> uint32_t {
        const uint32_t *w = (const uint32_t *)(win8 + (o & ~3u));
        return __builtin_amdgcn_alignbyte(w[1], w[0], o & 3u);
      };
      // filter: a candidate can only beat `bst` if it agrees with the scanned string in byte bst (:754-757); the four
      // bytes bst-3 .. bst are tested (two LDS reads, like two single bytes), which leaves far fewer false survivors
      uint32_t s_end = lds_u32(WI + (uint32_t)bst - 3);
      bool over = false;                                           // search finished
      // One batch of 64 candidates (distances d, nearest first).  All of them are filtered at once against the best
      // so far (:754-757); the survivors are then taken in order, exactly like the sequential walk (:812-822):
      // the whole wave compares one candidate with the scanned string, four bytes per lane, and if it is longer
      // it becomes the best and the remaining survivors are filtered again.
      auto batch = [&](uint32_t d) {
        const bool valid = d != 0;
        const bool inr = valid && d <= LF;
        bool pass = false;
        if (inr) pass = lds_u32(WI - d + (uint32_t)bst - 3) == s_end;
        unsigned long long pm = __ballot(pass);
        while (pm) {
          const int j = __ffsll((long long)pm) - 1;
          const uint32_t dj = RL(d, j);
          if (!hq && dj > LQ) { hq = true; rqq = bst >= 3 ? ((uint32_t)bst << 16) | bd : 0u; }     // the walk crosses the quarter limit (:733-735)
          const uint32_t o = 4u * (uint32_t)lane;
          const uint32_t x = lds_u32(WI - dj + o) ^ lds_u32(WI + o);
          const unsigned long long mm = __ballot(x != 0);
          int len;
          if (mm) { const int l0 = __ffsll((long long)mm) - 1; len = 4 * l0 + (int)(__builtin_ctz(RL(x, l0)) >> 3); }
          else { const uint32_t y = LDS_U16(win8, WI - dj + 256) ^ LDS_U16(win8, WI + 256); len = 256 + (y == 0 ? 2 : ((y & 0xFF) == 0 ? 1 : 0)); }
          len = len < LA ? len : LA;
          if (len > bst) {
            bst = len; bd = dj;
            if (len >= NICE) { over = true; break; }                                             // :815
            s_end = lds_u32(WI + (uint32_t)bst - 3);
            pass = pass && lane > j && lds_u32(WI - d + (uint32_t)bst - 3) == s_end;
          } else pass = pass && lane > j;
          pm = __ballot(pass);
        }
        if (!over && !hq && __any(valid && d > LQ)) { hq = true; rqq = bst >= 3 ? ((uint32_t)bst << 16) | bd : 0u; }
        // beyond the limit, or position 0, or no more candidates: the chain ends (:819-822)
        if (__any(!inr)) over = true;
      };
      // DM_AHEAD batches are kept on their way from the sorted order (a batch is worked off much faster than it arrives)
      uint32_t dq[DM_AHEAD];                                     // what the sorted order holds for the next DM_AHEAD batches
      dq[0] = r0;
#pragma unroll
      for (int u = 1; u < DM_AHEAD; u++) dq[u] = 64u * u < TT ? cand_load(ds, 64u * u + (uint32_t)lane) : 0u;
      for (uint32_t c0 = 0; c0 < TT && !over; c0 += 64 * DM_AHEAD) {
#pragma unroll
        for (int u = 0; u < DM_AHEAD; u++) {
          if (c0 + 64u * u < TT && !over) {
            const uint32_t dcur = cand_dist(ds, c0 + 64u * u + (uint32_t)lane, dq[u]);
            const uint32_t cn = c0 + 64u * (u + DM_AHEAD);
            if (cn < TT) dq[u] = cand_load(ds, cn + (uint32_t)lane);
            batch(dcur);
          }
        }
      }
      if (lane == 0) {
        const uint32_t packed = bst >= 3 ? ((uint32_t)bst << 16) | bd : 0u;
        store_result(B + ds.k, packed, hq ? rqq : packed, ds.og_full, ds.og_quarter, (ds.best_hq >> 14) & 1u);
      }
    }
    __syncthreads();
    DM_STAMP(3);
  }
}

// Safety valve of the demand loop: every remaining guess becomes demanded (lz_stage uses it when the parse keeps
// landing on new guesses round after round, which no ordinary input does).
__global__ void k_demand_all(Layout L, MatchPair *__restrict__ M, uint32_t *__restrict__ blk_demand, uint32_t *__restrict__ dbits) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= L.n || p >= lay_end(L, p >> 15)) return;
  const uint32_t f = M[p].full;
  if (f & M_GUESS) {                               // (a guess already marked has its bit set, or is set again: harmless)
    if (!(f & M_DEMAND)) M[p].full = f | M_DEMAND | M_BYSPEC;
    atomicOr(&dbits[p >> 5], 1u << (p & 31));
    blk_demand[p / DMB] = 1;
  }
}

// --------------------------------------------------------------------------------------------
// parser kernels
// --------------------------------------------------------------------------------------------
// one lane per chunk (the chunk logic itself is in zada_logic.h: parse_spec_chunk / parse_fix_chunk)
// A parse that lands on a guessed match record uses it and asks for the exact value: M_DEMAND on the record, a
// flag per k_match_demand block (so that the demand pass skips blocks without work) and a grand total for the host.
struct DemandMarker {
  MatchPair *M; uint32_t *blk_demand; uint32_t *n_demand; uint32_t by;   // by = M_BYSPEC for the speculative parse, 0 for the splice
  bool track = false;                                                    // the speculative parses from the second round on leave their length to beat (below)
  uint32_t *dbits;                                                       // one bit per position: to be searched in the next demand pass
  __device__ void operator()(uint32_t p, uint32_t full, uint32_t quarter, uint32_t beat) const {
    // Round 6: a speculative parse leaves the length its state has to beat with the guess (the smallest over the parses that land there: the chunk's own
    // and the chunk before's, if it ran over): the exact value changes the parse only if its length exceeds that -- a lazy look-up that was "not longer"
    // with the guess and still is with the exact value decides the same (85 - 93 % of the chunks the third and fourth demand pass flagged, measured).
    // Not in the FIRST parse: one more atomic per landed guess, 21 M of them, cost its 2 M lanes 2 ms for a list 12 % shorter; a guess that a later parse
    // lands on has never been landed on before (the first demand pass made every one of those exact), so every parse that uses it has left its length.
    // (`quarter` is the word as the parse read it: an atomic only where it lowers what stands there, and nothing is waited for)
    if (track && (beat < M_BEAT_MAX ? beat : M_BEAT_MAX) < (quarter >> M_BEAT_SHIFT))
      __hip_atomic_fetch_min(&M[p].quarter, (quarter & M_VALUE) | ((beat < M_BEAT_MAX ? beat : M_BEAT_MAX) << M_BEAT_SHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t want = M_DEMAND | by;
    if ((full & want) == want) return;
    // plain stores: every concurrent writer of these words writes a value that only adds flags
    M[p].full = full | want;
    atomicOr(&dbits[p >> 5], 1u << (p & 31));
    blk_demand[p / DMB] = 1;
    *n_demand = 1;
  }
};

// The parser's look-ups walk forward through the 8-byte match records, landing on every third or so: fetching each
// record on its own pulls the same 64-byte line from HBM several times (PMC: 27 GB per parse of 1 GiB, three times
// the table).  Each lane keeps the line of its last look-up in LDS (lane-interleaved, so lanes never share a bank).
// (Round 6, measured and dropped: the refills as WAVE-WIDE events -- every lane looks at the top of a turn of the parser's loop whether the turn's record is in
// its window of one / two / four lines, and if any lane's is not, all lanes that have left their window's first line fetch a new one together: one latency
// for the wave instead of one per lane and crossing.  The parse phase at 1 GiB: 28.1 / 29.9 / 30.1 ms against 27.6 -- the turns are not waiting for the
// refills' memory latency as much as for their own chain of LDS reads and branches, and the larger windows cost waves per CU.)
#ifndef ZADA_PS_BWORDS
#define ZADA_PS_BWORDS 16
#endif
#ifndef ZADA_PS_TOKS
#define ZADA_PS_TOKS 8
#endif
constexpr uint32_t PS_BWORDS = ZADA_PS_BWORDS, PS_TOKS = ZADA_PS_TOKS;
static_assert((PS_BWORDS == 8 || PS_BWORDS == 16) && (PS_TOKS == 4 || PS_TOKS == 8), "line sizes of the parser's LDS caches");
struct LineFetch {
  const MatchPair *M; uint64_t *slot;              // slot[r * 64] = record r of the cached line, for this lane
  uint32_t tag;
  __device__ MatchPair operator()(uint32_t p) {
    const uint32_t t = p >> 3;
    if (t != tag) {
      const uint4 *src = (const uint4 *)(M + ((uint64_t)t << 3));
      const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
      slot[0 * 64] = (uint64_t)a.x | ((uint64_t)a.y << 32); slot[1 * 64] = (uint64_t)a.z | ((uint64_t)a.w << 32);
      slot[2 * 64] = (uint64_t)b.x | ((uint64_t)b.y << 32); slot[3 * 64] = (uint64_t)b.z | ((uint64_t)b.w << 32);
      slot[4 * 64] = (uint64_t)c.x | ((uint64_t)c.y << 32); slot[5 * 64] = (uint64_t)c.z | ((uint64_t)c.w << 32);
      slot[6 * 64] = (uint64_t)d.x | ((uint64_t)d.y << 32); slot[7 * 64] = (uint64_t)d.z | ((uint64_t)d.w << 32);
      tag = t;
    }
    const uint64_t v = slot[(p & 7u) * 64];
    MatchPair r; r.full = (uint32_t)v; r.quarter = (uint32_t)(v >> 32);
    return r;
  }
  // the literal byte in front of a position: PS_BWORDS * 4 input bytes per lane are kept the same way (64; round 5 measured 32 with the token
  // buffer halved as well -- 7 KB of LDS per wave instead of 10, 22 waves per CU instead of 16 --: the parse phase 27.1 -> 28.0 ms, the refills
  // cost more than the waves bring; -DZADA_PS_BWORDS=8 -DZADA_PS_TOKS=4)
  const uint8_t *in; uint32_t *bslot; uint32_t btag;
  __device__ uint32_t byte(uint32_t p) {
    constexpr uint32_t SH = PS_BWORDS == 16 ? 6 : 5;
    const uint32_t t = p >> SH;
    if (t != btag) {
      const uint4 *src = (const uint4 *)(in + ((uint64_t)t << SH));
      const uint4 a = src[0], b = src[1];
      bslot[0 * 64] = a.x; bslot[1 * 64] = a.y; bslot[2 * 64] = a.z; bslot[3 * 64] = a.w;
      bslot[4 * 64] = b.x; bslot[5 * 64] = b.y; bslot[6 * 64] = b.z; bslot[7 * 64] = b.w;
      if (PS_BWORDS == 16) {
        const uint4 c = src[2], d = src[3];
        bslot[8 * 64] = c.x; bslot[9 * 64] = c.y; bslot[10 * 64] = c.z; bslot[11 * 64] = c.w;
        bslot[12 * 64] = d.x; bslot[13 * 64] = d.y; bslot[14 * 64] = d.z; bslot[15 * 64] = d.w;
      }
      btag = t;
    }
    return (bslot[((p >> 2) & (PS_BWORDS - 1u)) * 64] >> (8 * (p & 3u))) & 0xFFu;
  }
};

// Tokens of a chunk are written eight at a time (two 16-byte stores to the chunk's own, 32-byte aligned token area)
// instead of one scattered 4-byte store each; the eight wait in LDS, lane-interleaved.
struct TokSink {
  uint32_t *dst; uint32_t *buf; uint32_t n;
  __device__ void push(uint32_t t) {
    buf[(n & (PS_TOKS - 1u)) * 64] = t;
    n++;
    if ((n & (PS_TOKS - 1u)) == 0) {
      uint4 a;
      a.x = buf[0]; a.y = buf[64]; a.z = buf[128]; a.w = buf[192];
      uint4 *o = (uint4 *)(dst + n - PS_TOKS);
      o[0] = a;
      if (PS_TOKS == 8) { uint4 b; b.x = buf[256]; b.y = buf[320]; b.z = buf[384]; b.w = buf[448]; o[1] = b; }
    }
  }
  __device__ void flush() { for (uint32_t i = n & ~(PS_TOKS - 1u); i < n; i++) dst[i] = buf[(i & (PS_TOKS - 1u)) * 64]; }
};

// The chunks a demand pass has flagged (chg), as a list: the speculative parse of a later round is launched over the list, every lane with a
// chunk to parse -- flagged chunks are a few per cent, spread evenly, so that a launch over all chunks had nearly every wave run one or two lanes
// for the whole length of a parse (four such rounds cost as much as the first parse of everything).  One counter reservation per workgroup.
__global__ void __launch_bounds__(1024) k_list_flagged(const uint8_t *__restrict__ flag, uint32_t n, uint32_t *__restrict__ list, uint32_t *__restrict__ count) {
  __shared__ uint32_t wcnt[16], wbase;
  const uint32_t i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool f = i < n && flag[i];
  const unsigned long long mk = __ballot(f);
  if (lane == 0) wcnt[w] = (uint32_t)__popcll(mk);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int k = 0; k < 16; k++) { const uint32_t c = wcnt[k]; wcnt[k] = t; t += c; }
    wbase = t ? atomicAdd(count, t) : 0u;
  }
  __syncthreads();
  if (f) list[wbase + wcnt[w] + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = i;
}

#ifdef ZADA_RESPEC_STATS
__device__ unsigned long long g_respec_dbg[4];
#endif
__global__ void k_parse_spec(ParseIO io, uint32_t nchunks, uint32_t *__restrict__ spec_tok, uint32_t *__restrict__ spec_cnt,
                             uint32_t *__restrict__ Fbits, uint32_t *__restrict__ Lbits, ExitState *__restrict__ exits,
                             DemandMarker dm, const uint32_t *__restrict__ list /* null: every chunk */, const uint32_t *__restrict__ list_n,
                             uint32_t exact_max /* lists of up to this many chunks are left to k_parse_spec_exact */) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (list) {                                         // the flagged chunks only (the grid is sized for all of them; what lies beyond the list ends here)
    const uint32_t nl = *list_n;
    if (nl <= exact_max || k >= nl) return;
    k = list[k];
  }
  if (k >= nchunks) return;
  if (io.segend) io.n = io.segend[((uint64_t)k * PCHUNK) >> 15] & 0x7FFFFFFFu;     // a batch: the input ends where the chunk's entry ends
  uint32_t ntok = 0;
  ExitState ex;
#ifdef ZADA_RESPEC_STATS   /* how many of a later round's parses change anything: the chunk's old tokens and exit against the new ones */
  uint32_t old_h = 0, old_n = 0; ExitState old_ex{0, 0};
  if (list) { old_n = spec_cnt[k]; old_ex = exits[k]; const uint32_t *ot = spec_tok + (uint64_t)k * PTOK_STRIDE; for (uint32_t i = 0; i < old_n; i++) old_h = old_h * 0x9E3779B1u + ot[i]; }
#endif
  __shared__ uint64_t lines[8 * 64];
  __shared__ uint32_t tbuf[PS_TOKS * 64];
  __shared__ uint32_t blines[PS_BWORDS * 64];
  LineFetch lf; lf.M = io.M; lf.slot = lines + threadIdx.x; lf.tag = 0xFFFFFFFFu; lf.in = io.in; lf.bslot = blines + threadIdx.x; lf.btag = 0xFFFFFFFFu;
  TokSink ts; ts.dst = spec_tok + (uint64_t)k * PTOK_STRIDE; ts.buf = tbuf + threadIdx.x; ts.n = 0;
  parse_spec_chunk_to(io, k, PCHUNK, ts, Fbits, Lbits, ex, dm, lf);
  ts.flush();
  ntok = ts.n;
  spec_cnt[k] = ntok;
  exits[k] = ex;
#ifdef ZADA_RESPEC_STATS
  if (list) {
    __threadfence();
    uint32_t new_h = 0; const uint32_t *nt = spec_tok + (uint64_t)k * PTOK_STRIDE; for (uint32_t i = 0; i < ntok; i++) new_h = new_h * 0x9E3779B1u + nt[i];
    atomicAdd(&g_respec_dbg[0], 1ull);
    if (new_h == old_h && ntok == old_n && ex.pos == old_ex.pos && ex.kind == old_ex.kind) atomicAdd(&g_respec_dbg[1], 1ull);
  }
#endif
}

// --------------------------------------------------------------------------------------------
// Round 6: the speculative parse of the chunks a demand pass has flagged, with the EXACT search inside the parse.
// One wave per flagged chunk.  All 64 lanes run the chunk's parse in step with each other (one state, uniform control flow); where the
// parse lands on a guess the wave searches that position to the end there and then -- the demand pass's scan (64 candidates of the
// bucket's run in the sorted order per batch, four filter bytes, whole-wave compares, the quarter-chain snapshot where the batch
// crosses the quarter distance), resumed where k_match stopped, with the candidates' bytes read from memory instead of a staged
// window -- writes the exact record and parses on with it.  Such a parse has used exact values only: the chunk is never flagged again,
// and the rounds "parse the flagged chunks with guesses, search the guesses they landed on, flag, parse again" (four of them on the
// benchmark stream, each a handful of launches that last as long as one lane's parse whatever the number of chunks) become one launch.
// The chunk's match records are fetched into LDS by the whole wave first (the parse is a chain of dependent look-ups).
// --------------------------------------------------------------------------------------------
struct WaveExact {
  const uint8_t *in; Layout L; DistPlanes dp; RunPtrs rp; const uint16_t *tailsK; MatchPair *M; int nice_cfg; const uint16_t *resume;
  __device__ __forceinline__ uint32_t g32(uint64_t o) const { return *(const u32u *)(in + o); }
  // Longest_Match of position p over the full chain and its quarter-chain snapshot, started from the guess `og` (wave uniform)
  __device__ MatchPair search(uint64_t p, MatchPair og) const {
    const int lane = threadIdx.x & 63;
    const uint64_t seg = p >> 15, n = lay_end(L, seg);
    const bool seg_first = lay_first(L, seg), prev_first = seg > 0 && lay_first(L, seg - 1);
    const uint64_t rem = n - p;
    const int la = rem < 258 ? (int)rem : 258;                   // Longest_Match never returns more
    const int nice = nice_cfg < la ? nice_cfg : la;              // lz77.adb:858-860
    const uint64_t pbase = (seg << 15) - 32768;
    const uint32_t dlimv = dp.limits(p);
    uint32_t dl[NLEVELS];
#pragma unroll
    for (int l = 0; l < NLEVELS; l++) dl[l] = dp.d[l][p];
    const uint32_t idx1 = rp.idx[p], c1 = rp.cnt[p];
    uint32_t t = 0xFFFFu;
    if (!seg_first) {
      // (only the occupied buckets of a tails table are written: an entry is the bucket's tail iff it names an inserted position of
      // the previous segment that hashes to the bucket -- see k_cross_links; a tail out of reach leaves nothing to scan there)
      const uint32_t key = hashL_of(*(const u64u *)(in + p), 3 + NLEVELS);
      const uint32_t tt = tailsK[(seg - 1) * 65536ull + key];
      const uint64_t q = pbase + tt;
      if (tt < lay_inserted(L, seg - 1) && p - q <= (uint64_t)MAX_DIST && hashL_of(*(const u64u *)(in + q), 3 + NLEVELS) == key) t = tt;
    }
    uint32_t idx2 = 0, c2 = 0;
    if (t != 0xFFFFu) { idx2 = rp.idx[pbase + t]; c2 = (uint32_t)rp.cnt[pbase + t] + 1u; }
    const uint32_t df = dlimv & 0xFFFF, dq = dlimv >> 16;
    const uint32_t LF = dl[0] == (uint32_t)MAX_DIST ? (uint32_t)MAX_DIST : (df < (uint32_t)(MAX_DIST - 1) ? df : (uint32_t)(MAX_DIST - 1));   // :850 / :820-822
    const uint32_t LQ = dq < LF ? dq : LF;                                                                                                       // :733-735
    int bst = 2;
    uint32_t bd = 0, rqq = 0;
    bool chain_ok = true;
#pragma unroll
    for (int l = 0; l < NLEVELS; l++) {                            // (levels on their own: see k_match)
      const bool v = la >= 3 + l && dl[l] != 0 && dl[l] <= LF;
      if (v) { bst = 3 + l; bd = dl[l]; if (dl[l] <= LQ) rqq = ((uint32_t)(3 + l) << 16) | dl[l]; }
      chain_ok = v;
    }
    bool hq = chain_ok && bd > LQ;
    // The search was begun by k_match: its best so far is the guess, and `resume` the candidate it had come to.
    // The candidates before that one (nearer) are dropped from the two runs.
    uint32_t i1 = idx1, n1 = c1, i2 = idx2, n2 = c2;
    if (chain_ok) {
      const uint32_t gl = (og.full & M_VALUE) >> 16;
      if (gl >= 3) { bst = (int)gl; bd = og.full & 0xFFFFu; }
      hq = (og.full & M_HAVEQ) != 0;
      if (hq) rqq = og.quarter & M_VALUE;
      const uint64_t q = p - resume[p];
      const uint32_t iq = rp.idx[q];
      const uint32_t skip = q >= (seg << 15) ? idx1 - 1 - iq : c1 + (idx2 - iq);   // candidates nearer than q
      if (skip < c1) { i1 = idx1 - skip; n1 = c1 - skip; }
      else { const uint32_t s2 = skip - c1 < c2 ? skip - c1 : c2; n1 = 0; i2 = idx2 - s2; n2 = c2 - s2; }
    }
    MatchPair r;
    if (!(chain_ok && bst < nice && n1 + n2 > 0)) {
      const uint32_t packed = bst >= 3 ? ((uint32_t)bst << 16) | bd : 0u;
      r.full = packed; r.quarter = !chain_ok ? rqq : (hq ? rqq : packed);
      if (lane == 0) M[p] = r;
      return r;
    }
    // ---- the scan: candidate c of the position (nearest first) as a distance; 0 = no such candidate.  Position 0 is never a match source (:467)
    const uint16_t *sprev = rp.S + ((seg << 15) - 32768);          // (32-bit offsets from the start of the previous segment in the sorted order)
    const uint32_t TT = n1 + n2, po = (uint32_t)p & 32767u;
    auto cand = [&](uint32_t c) -> uint32_t {
      const bool in1 = c < n1, in2 = !in1 && c - n1 < n2;
      const uint32_t a = in1 ? 32768u + i1 - 1u - c : (in2 ? i2 - (c - n1) : 32768u);
      const uint32_t raw = sprev[a];
      const bool q0 = raw == 0 && ((in1 && seg_first) || (in2 && prev_first));   // the candidate is position 0 of the stream
      return q0 ? 0u : (in1 ? po - raw : (in2 ? po + 32768u - raw : 0u));
    };
    uint32_t s_end = g32(p + (uint32_t)bst - 3);
    bool over = false;
    uint32_t dnext = cand((uint32_t)lane);
    for (uint32_t c0 = 0; c0 < TT && !over; c0 += 64) {
      const uint32_t d = dnext;
      if (c0 + 64 < TT) dnext = cand(c0 + 64 + (uint32_t)lane);
      const bool valid = d != 0;
      const bool inr = valid && d <= LF;
      bool pass = false;
      if (inr) pass = g32(p - d + (uint32_t)bst - 3) == s_end;    // (:754-757, four bytes bst-3 .. bst instead of two)
      unsigned long long pm = __ballot(pass);
      while (pm) {
        const int j = __ffsll((long long)pm) - 1;
        const uint32_t dj = RL(d, j);
        if (!hq && dj > LQ) { hq = true; rqq = bst >= 3 ? ((uint32_t)bst << 16) | bd : 0u; }     // the walk crosses the quarter limit (:733-735)
        const uint32_t o = 4u * (uint32_t)lane;
        const uint32_t x = g32(p - dj + o) ^ g32(p + o);
        const unsigned long long mm = __ballot(x != 0);
        int len;
        if (mm) { const int l0 = __ffsll((long long)mm) - 1; len = 4 * l0 + (int)(__builtin_ctz(RL(x, l0)) >> 3); }
        else { const uint32_t y = (g32(p - dj + 256) ^ g32(p + 256)) & 0xFFFFu; len = 256 + (y == 0 ? 2 : ((y & 0xFF) == 0 ? 1 : 0)); }
        len = len < la ? len : la;
        if (len > bst) {
          bst = len; bd = dj;
          if (len >= nice) { over = true; break; }                                               // :815
          s_end = g32(p + (uint32_t)bst - 3);
          pass = pass && lane > j && g32(p - d + (uint32_t)bst - 3) == s_end;
        } else pass = pass && lane > j;
        pm = __ballot(pass);
      }
      if (!over && !hq && __any(valid && d > LQ)) { hq = true; rqq = bst >= 3 ? ((uint32_t)bst << 16) | bd : 0u; }
      // beyond the limit, or position 0, or no more candidates: the chain ends (:819-822)
      if (__any(!inr)) over = true;
    }
    const uint32_t packed = bst >= 3 ? ((uint32_t)bst << 16) | bd : 0u;
    r.full = packed; r.quarter = hq ? rqq : packed;
    if (lane == 0) M[p] = r;
    return r;
  }
};

constexpr uint32_t PX_RECS = PCHUNK + 320;          // match records of a chunk kept in LDS (a parse runs over its chunk's end by a match or two; beyond: memory)
struct WaveFetch {
  const uint64_t *recs; uint32_t c0; const MatchPair *Mg; const WaveExact *wx;
  __device__ MatchPair operator()(uint32_t p) const {
    MatchPair r;
    if (p - c0 < PX_RECS) { const uint64_t v = recs[p - c0]; r.full = (uint32_t)v; r.quarter = (uint32_t)(v >> 32); }
    else r = Mg[p];
    if (r.full & M_GUESS) r = wx->search(p, r);
    return r;
  }
};
struct WaveSink {                                   // every lane counts, lane 0 writes
  uint32_t *dst; uint32_t n;
  __device__ void push(uint32_t t) { if ((threadIdx.x & 63) == 0) dst[n] = t; n++; }
};
constexpr int PX_WAVES = 4;
__global__ void __launch_bounds__(64 * PX_WAVES) k_parse_spec_exact(ParseIO io, uint32_t nchunks, uint32_t *__restrict__ spec_tok, uint32_t *__restrict__ spec_cnt,
                                                                     uint32_t *__restrict__ Fbits, uint32_t *__restrict__ Lbits, ExitState *__restrict__ exits,
                                                                     WaveExact wx, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n, uint32_t exact_max) {
  __shared__ uint64_t recs[PX_WAVES][PX_RECS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t nl = *list_n, stride = gridDim.x * PX_WAVES;
  if (nl > exact_max) return;                         // (a long list: the lane-per-chunk parse takes it)
  for (uint32_t i = blockIdx.x * PX_WAVES + w; i < nl; i += stride) {      // (a wave's loop: no workgroup barrier inside)
    const uint32_t k = list[i];
    if (k >= nchunks) continue;
    ParseIO cio = io;
    if (io.segend) cio.n = io.segend[((uint64_t)k * PCHUNK) >> 15] & 0x7FFFFFFFu;     // a batch: the input ends where the chunk's entry ends
    const uint32_t c0 = k * PCHUNK;
    {
      const uint64_t *src = (const uint64_t *)(io.M + c0);
      const uint32_t have = (uint64_t)c0 + PX_RECS <= io.n ? PX_RECS : (uint32_t)(io.n - c0);      // (the parse never looks beyond the input)
      for (uint32_t j = lane; j < have; j += 64) recs[w][j] = src[j];
    }
    __builtin_amdgcn_wave_barrier();
    WaveFetch wf; wf.recs = recs[w]; wf.c0 = c0; wf.Mg = io.M; wf.wx = &wx;
    WaveSink ts; ts.dst = spec_tok + (uint64_t)k * PTOK_STRIDE; ts.n = 0;
    ExitState ex;
    parse_spec_chunk_to(cio, k, PCHUNK, ts, Fbits, Lbits, ex, NoGuess(), wf);
    if (lane == 0) { spec_cnt[k] = ts.n; exits[k] = ex; }
    __builtin_amdgcn_wave_barrier();
  }
}

// Fix-up: chunk k re-parses from the true exit of chunk k-1 until it meets the speculative parse.
// dirty_in[k] != 0 : this chunk's entry changed since the last round and must be (re)done.
__global__ void k_parse_fix(ParseIO io, uint32_t nchunks, const uint32_t *__restrict__ spec_tok, const uint32_t *__restrict__ spec_cnt,
                            const uint32_t *__restrict__ Fbits, const uint32_t *__restrict__ Lbits,
                            const ExitState *__restrict__ spec_exits, ExitState *__restrict__ true_exits,
                            uint32_t *__restrict__ fix_tok, uint32_t *__restrict__ fix_cnt,
                            uint32_t *__restrict__ take_from, uint32_t *__restrict__ start_pos,
                            const uint8_t *__restrict__ dirty_in, uint8_t *__restrict__ dirty_out,
                            uint32_t *__restrict__ n_changed, DemandMarker dm, ExitState entry0, uint32_t fix_stride, uint32_t *__restrict__ fix_overflow) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nchunks) return;
  if (!dirty_in[k]) return;
  ExitState entry = entry0;                          // the state the parse of this buffer starts from
  if (k > 0) entry = true_exits[k - 1];
  if (io.segend) {                                   // a batch: every entry is parsed from its own first byte to its own end
    const uint32_t se = io.segend[((uint64_t)k * PCHUNK) >> 15];
    io.n = se & 0x7FFFFFFFu;
    if ((se >> 31) && ((k * PCHUNK) & 32767u) == 0) entry = ExitState{k * PCHUNK, SYNC_F};
  }
  const ExitState old_exit = true_exits[k];
  ExitState new_exit;
  uint32_t ntok = 0, take = 0, u0 = 0;
  parse_fix_chunk(io, k, PCHUNK, entry, spec_tok + (uint64_t)k * PTOK_STRIDE, spec_cnt[k], Fbits, Lbits, spec_exits[k],
                  fix_tok + (uint64_t)k * fix_stride, ntok, take, u0, new_exit, dm, DirectFetch{io.M}, fix_stride, fix_overflow);
  fix_cnt[k] = ntok;
  take_from[k] = take;
  start_pos[k] = u0;
  true_exits[k] = new_exit;
  if (k + 1 < nchunks && (new_exit.pos != old_exit.pos || new_exit.kind != old_exit.kind)) {
    dirty_out[k + 1] = 1;
    atomicAdd(n_changed, 1u);
    atomicMin(n_changed + 1, k);                    // the lowest chunk with a new exit: its entry was final, so is its exit
  }
}

// Runs of maximal matches.  In constant or exactly periodic data every position matches 258 bytes, the true parse is
// "match, match, match ..." from wherever it entered the run, and the speculative parses (started at the chunk
// boundaries) can never meet it: the splice would advance one chunk per round.  But a match of max_lazy_match or
// more is taken at once (:849, :875-899), so from a final F state q with an exact 258-byte match the next final
// states are q + 258, q + 516, ... for as long as every landing position also has one.  One wave follows the run
// from the exit of the lowest chunk that changed in this round, 64 landings per step, and writes the exit of every
// chunk it crosses; those chunks are then re-parsed, all in the same next round.
__global__ void __launch_bounds__(64) k_fix_forward(ParseIO io, uint32_t nchunks, ExitState *__restrict__ true_exits,
                                                    uint8_t *__restrict__ dirty_out, uint32_t *__restrict__ n_changed) {
  const uint32_t k = n_changed[1];
  if (k >= nchunks || 258 < io.cfg.lazy || io.segend) return;     // (batches: entries are small, the plain splice is enough)
  const ExitState e0 = true_exits[k];
  if (e0.kind != SYNC_F) return;
  const int lane = threadIdx.x;
  for (uint64_t q0 = e0.pos;; q0 += 258ull * 64) {
    const uint64_t q = q0 + 258ull * lane;
    bool good = false;
    if (q + 258 <= io.n) { const uint32_t f = io.M[q].full; good = !(f & M_GUESS) && ((f & M_VALUE) >> 16) == 258u; }
    const unsigned long long bad = ~__ballot(good);
    const int nvalid = bad ? __ffsll((long long)bad) - 1 : 64;      // landings 0 .. nvalid-1 are final F states with a 258 match
    // landing j is a final F state if all landings before it were good; it is the exit of the chunk ending at c1
    // if it is the first state at or beyond c1
    if (lane <= nvalid && (lane > 0 || q0 != e0.pos)) {
      const uint64_t c1 = (q / PCHUNK) * PCHUNK;
      if (q - c1 < 258 && c1 >= PCHUNK) {
        const uint32_t kk = (uint32_t)(c1 / PCHUNK) - 1;
        if (kk > k && kk < nchunks) {
          const ExitState old = true_exits[kk];
          if (old.pos != (uint32_t)q || old.kind != SYNC_F) {
            ExitState ne; ne.pos = (uint32_t)q; ne.kind = SYNC_F;
            true_exits[kk] = ne;
            if (kk + 1 < nchunks) dirty_out[kk + 1] = 1;
            atomicAdd(n_changed, 1u);
          }
        }
      }
    }
    if (nvalid < 64) break;
  }
}

// A parse that enters the buffer at `e` (the exit of the shard before): the chunks in front of e.pos emit nothing and
// hand the state on unchanged.
__global__ void k_seed_entry(ExitState *__restrict__ true_exits, uint32_t kE, ExitState e) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < kE) true_exits[k] = e;
}

// per-chunk true token count
__global__ void k_tok_count(uint32_t nchunks, const uint32_t *__restrict__ spec_cnt, const uint32_t *__restrict__ fix_cnt,
                            const uint32_t *__restrict__ take_from, uint32_t *__restrict__ counts) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nchunks) counts[k] = fix_cnt[k] + (spec_cnt[k] - take_from[k]);
}

// one wave per chunk: copy tokens, compute byte position of every atom
__global__ void __launch_bounds__(256) k_tok_compact(uint32_t nchunks, const uint32_t *__restrict__ spec_tok,
                                                     const uint32_t *__restrict__ spec_cnt, const uint32_t *__restrict__ fix_tok,
                                                     const uint32_t *__restrict__ fix_cnt, const uint32_t *__restrict__ take_from,
                                                     const uint32_t *__restrict__ start_pos, const uint32_t *__restrict__ offsets,
                                                     uint32_t *__restrict__ atoms, uint32_t *__restrict__ apos, uint32_t k0, uint32_t apos_bias, uint32_t fix_stride) {
  // chunks [k0, k0 + nchunks); offsets[] is the exclusive scan over exactly these chunks
  const uint32_t kk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (kk >= nchunks) return;
  const uint32_t k = k0 + kk;
  const uint32_t nf = fix_cnt[k], tf = take_from[k], ns = spec_cnt[k] - tf;
  const uint32_t *ft = fix_tok + (uint64_t)k * fix_stride;
  const uint32_t *st = spec_tok + (uint64_t)k * PTOK_STRIDE + tf;
  uint32_t out = offsets[kk];
  uint32_t pos = start_pos[k] + apos_bias;
  const uint32_t total = nf + ns;
  for (uint32_t b = 0; b < total; b += 64) {
    uint32_t i = b + lane;
    uint32_t t = 0, len = 0;
    if (i < total) { t = i < nf ? ft[i] : st[i - nf]; len = tok_len(t); }
    uint32_t incl = len;
    for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    if (i < total) { atoms[out + i] = t; apos[out + i] = pos + incl - len; }
    pos += __shfl(incl, 63);
  }
}

// ---- exclusive scan of uint32 counts (three small kernels) ----
__global__ void __launch_bounds__(1024) k_scan_block(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                     uint32_t *__restrict__ block_sums, uint32_t n) {
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint32_t i = blockIdx.x * 1024 + tid;
  uint32_t v = i < n ? in[i] : 0, incl = v;
  for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int k = 0; k < w; k++) base += wsum[k];
  if (i < n) out[i] = base + incl - v;
  if (tid == 1023) block_sums[blockIdx.x] = base + incl;
}
__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t *__restrict__ block_sums, uint32_t nb, uint32_t *__restrict__ total) {
  // single block; nb <= 2^20
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t b = 0; b < nb; b += 1024) {
    uint32_t i = b + tid;
    uint32_t v = i < nb ? block_sums[i] : 0, incl = v;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t base = carry;
    for (int k = 0; k < w; k++) base += wsum[k];
    if (i < nb) block_sums[i] = base + incl - v;
    __syncthreads();
    if (tid == 1023) carry = base + incl;
    __syncthreads();
  }
  if (tid == 0) *total = carry;
}
__global__ void k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ block_sums, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_sums[i >> 10];
}

// No-LZ77 front end (LZ77.No_LZ77, lz77.adb:2191-2194): every byte is a literal atom.
__global__ void k_literal_atoms(const uint8_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ atoms, uint32_t *__restrict__ apos, uint32_t bias) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { atoms[i] = in[i]; apos[i] = (uint32_t)i + bias; }
}

// the same for a batch of entries: per parse chunk, the bytes that belong to its entry (every byte a literal atom)
__global__ void k_literal_counts(uint32_t nchunks, Layout L, uint32_t *__restrict__ counts) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nchunks) return;
  const uint64_t c0 = (uint64_t)k * PCHUNK, e = lay_end(L, c0 >> 15);
  counts[k] = e > c0 ? (uint32_t)(e - c0 < PCHUNK ? e - c0 : PCHUNK) : 0u;
}
__global__ void __launch_bounds__(256) k_literal_atoms_batch(uint32_t nchunks, const uint8_t *__restrict__ in, const uint32_t *__restrict__ counts,
                                                             const uint32_t *__restrict__ offsets, uint32_t *__restrict__ atoms, uint32_t *__restrict__ apos) {
  const uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (k >= nchunks) return;
  const uint32_t cnt = counts[k], o = offsets[k], c0 = k * PCHUNK;
  for (uint32_t i = lane; i < cnt; i += 64) { atoms[o + i] = in[c0 + i]; apos[o + i] = c0 + i; }
}

// --------------------------------------------------------------------------------------------
// host side of the LZ stage
// --------------------------------------------------------------------------------------------
void exclusive_scan_u32(hipStream_t st, const uint32_t *d_in, uint32_t *d_out, uint32_t *d_sums, uint32_t *d_total, uint32_t n) {
  uint32_t nb = (n + 1023) / 1024;
  hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(1024), 0, st, d_in, d_out, d_sums, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, d_sums, nb, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3((n + 255) / 256), dim3(256), 0, st, d_out, d_sums, n);
}

int lz_shard(Ctx *c, int level, const ShardJob &job, ShardResult *res) {
  hipStream_t st = c->stream;
  Workspace &W = c->ws;
  const uint64_t n = job.nbuf;
  res->ntok = 0; res->exit = ExitState{(uint32_t)n, SYNC_F}; res->warm = ExitState{job.tok_lo, SYNC_F};
  if (n == 0) return 0;
  if (job.need && (level == 0 || n < 32768 + 2)) { if (int rn = job.need(n)) return rn; }      // (no segments to go by: everything first)
  if (level == 0 && job.segend) {                  // a batch: the chunks' byte counts stand for their token counts (zada_api.hip, batch_core)
    const uint32_t nch = (uint32_t)((n + PCHUNK - 1) / PCHUNK);
    const Layout L{job.segend, n};
    hipLaunchKernelGGL(k_literal_counts, dim3((nch + 255) / 256), dim3(256), 0, st, nch, L, W.counts);
    exclusive_scan_u32(st, W.counts, W.offsets, W.scan_sums, W.n_changed, nch);
    uint32_t total = 0;
    hipMemcpyAsync(&total, W.n_changed, 4, hipMemcpyDeviceToHost, st);
    if (hip_check(c, hipStreamSynchronize(st), "literal counts")) return ZADA_E_HIP_;
    if (total > job.cap_atoms) { c->err = "atom array overflow"; return -2; }
    hipLaunchKernelGGL(k_literal_atoms_batch, dim3((nch + 3) / 4), dim3(256), 0, st, nch, W.in, W.counts, W.offsets, job.dst_atoms, job.dst_apos);
    res->ntok = total;
    return hip_check(c, hipGetLastError(), "k_literal_atoms_batch");
  }
  if (level == 0) {
    const uint64_t hi = job.final ? n : job.tok_hi, cnt = hi - job.tok_lo;
    uint32_t *dst_atoms = job.dst_atoms, *dst_apos = job.dst_apos;
    if (cnt > job.cap_atoms) {                                       // (one atom per byte: more than the caller guessed -- it makes room, or it is an error)
      if (!job.grow_atoms) { c->err = "atom array overflow"; return -2; }
      if (int rg = job.grow_atoms(cnt, &dst_atoms, &dst_apos)) return rg;
    }
    if (cnt) hipLaunchKernelGGL(k_literal_atoms, dim3(2048), dim3(256), 0, st, W.in + job.tok_lo, cnt, dst_atoms, dst_apos, job.apos_bias + job.tok_lo);
    res->ntok = (uint32_t)cnt;
    res->exit = ExitState{(uint32_t)hi, SYNC_F};
    return hip_check(c, hipGetLastError(), "k_literal_atoms");
  }
  const LzConfig cfg = lz_config(level);
  const Layout L{job.segend, n};
  const uint64_t n_ins = job.segend ? n : (n >= 2 ? n - 2 : 0);
  const uint32_t nseg = (uint32_t)((n_ins + 32767) / 32768);
  c->tmark("lz:begin");
  if (!c->lz_attrs_set) {                            // per context: the attribute belongs to the function object of the current device
    hipFuncSetAttribute((const void *)k_prev_links<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PL_LDS_RUNS);
    hipFuncSetAttribute((const void *)k_prev_links<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PL_LDS);
    hipFuncSetAttribute((const void *)k_cross_links, hipFuncAttributeMaxDynamicSharedMemorySize, CL_LDS_PLANE);
    hipFuncSetAttribute((const void *)k_match, hipFuncAttributeMaxDynamicSharedMemorySize, MATCH_LDS);
    hipFuncSetAttribute((const void *)k_match_demand, hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS);
    c->lz_attrs_set = true;
  }
  LevelPtrs lv;
  DistPlanes dpl;
  RunPtrs rpt; rpt.S = W.SK; rpt.idx = W.idxK; rpt.cnt = W.cntK;
  for (int l = 0; l < NLEVELS; l++) dpl.d[l] = W.dplane[l];
  dpl.dlim = W.dlim; dpl.dlim_bits = W.dlim_bits;
  if (nseg > 0) {
    for (int l = 0; l < NLEVELS; l++) { lv.prev[l] = W.lprev[l]; lv.tails[l] = W.ltails[l]; }
#ifndef ZADA_OLD_INIT
    // "no chain-length limit" for every position (k_bucket_limits writes the few there are, and sets their bits): the bit map is cleared next to
    // k_prev_links (round 5: a bit per position instead of the 4-byte plane itself -- 4 bytes per input byte less written, and as many less read)
    hipEventRecord(c->ev_dlim, st);                                  // (the shard before has read the plane)
    hipStreamWaitEvent(c->stream2, c->ev_dlim, 0);
    hipMemsetAsync(W.dlim_bits, 0, (size_t)(n / 32 + 2) * 4, c->stream2);
    hipEventRecord(c->ev_dlim, c->stream2);
#endif
    // Runs of R segments per workgroup of k_prev_links (it then makes the cross links of all but a run's first segment itself; k_cross_links takes
    // those): as long as the launch still fills the chip several times over -- R = 16 from 512 MiB on, 1 (no runs) below 64 MiB; a piece of an
    // input that is still arriving (2 048 segments) goes as 256 workgroups of 8.  Knob "link_run" (0 = this rule).
    uint32_t R = 1;
    if (c->knob_link_run > 0) R = (uint32_t)c->knob_link_run;
    else {
      while (R < 16 && nseg / (2 * R) >= 1024) R *= 2;
      if (job.need && R > 8) R = 8;                                       // (a piece of 2 048 segments: one workgroup per CU)
    }
    // (zada_set_knob keeps R a power of two up to 64; the pieces below start at multiples of 2 048 segments, so every piece starts at a run)
    if ((R & (R - 1)) || R > 64) { c->err = "link_run is not a power of two up to 64"; return -1; }
    auto prev_links = [&](uint32_t s0, uint32_t s1) {                     // segments [s0, s1), s0 a multiple of R
      if (R > 1) hipLaunchKernelGGL(k_prev_links<true>, dim3((s1 - s0 + R - 1) / R), dim3(1024), PL_LDS_RUNS, st, W.in, L, cfg.chain, cfg.chain >> 2, lv,
                                    W.S3, W.T3, W.bsc3, dpl, rpt, (unsigned long long *)W.dbg, W.segmax, W.heavy, s0, R, s1);
      else hipLaunchKernelGGL(k_prev_links<false>, dim3(s1 - s0), dim3(1024), PL_LDS, st, W.in, L, cfg.chain, cfg.chain >> 2, lv,
                              W.S3, W.T3, W.bsc3, dpl, rpt, (unsigned long long *)W.dbg, W.segmax, W.heavy, s0, 1u, s1);
    };
    auto cross_links = [&](uint32_t from, uint32_t to) {                  // the segments of [from, to) that k_prev_links did not link itself
      const uint32_t first = (from + R - 1) / R * R;
      if (first < to) hipLaunchKernelGGL(k_cross_links, dim3((to - first + R - 1) / R, NLEVELS), dim3(1024), CL_LDS_PLANE, st, W.in, L, lv, dpl, first, R);
    };
    static_assert(BLOOM_WORDS == 4096, "Workspace::bloom4 is sized in zada_api.hip");
    auto cross_dist = [&]() {                                             // the positions from 32 768 on (everything is there: the last piece has come)
      const uint32_t nb = (uint32_t)((n_ins - 32768 + CD_THREADS - 1) / CD_THREADS);
      if (!c->knob_cd_filter) {
        hipLaunchKernelGGL(k_cross_dist<CD_ALL>, dim3(nb), dim3(CD_THREADS), 0, st, W.in, L, lv, W.S3, W.T3, W.bsc3, dpl, (uint64_t)32768, (uint64_t)n, CdTail{nullptr, nullptr, W.n_changed + 4, 0u, nullptr, W.n_changed + 5, 0u});
        return;
      }
      // the filters of the segments that are somebody's previous one; the sweep; the list of the walks it left open
      hipMemsetAsync(W.n_changed + 4, 0, 4, st);
      hipLaunchKernelGGL(k_bloom4, dim3(nseg - 1), dim3(256), 0, st, W.in, L, W.bloom4, 0u);
      hipMemsetAsync(W.n_changed + 5, 0, 4, st);
      const uint32_t lcap = c->knob_cd_list_cap > 0 && (uint64_t)c->knob_cd_list_cap < W.cd_cap ? (uint32_t)c->knob_cd_list_cap : (uint32_t)W.cd_cap;
      const CdTail tl{W.bloom4, W.cd_list, W.n_changed + 4, lcap, W.cd_list + W.cd_cap, W.n_changed + 5, lcap / 4};
      hipLaunchKernelGGL(k_cross_dist<CD_SWEEP>, dim3(nb), dim3(CD_THREADS), 0, st, W.in, L, lv, W.S3, W.T3, W.bsc3, dpl, (uint64_t)32768, (uint64_t)n, tl);
      hipLaunchKernelGGL(k_cross_dist<CD_LIST>, dim3((uint32_t)((W.cd_cap + CD_THREADS - 1) / CD_THREADS)), dim3(CD_THREADS), 0, st, W.in, L, lv, W.S3, W.T3, W.bsc3, dpl, (uint64_t)0, (uint64_t)n, tl);
      hipLaunchKernelGGL(k_cross_scan, dim3(2048), dim3(256), 0, st, W.in, L, dpl, tl);
    };
    if (job.need) {
      // The input is still arriving (host buffers): k_prev_links and k_bucket_limits on the segments of what has come, 64 MiB at a time --
      // a segment's workgroups read its 32 KiB and at most 31 bytes behind them, so a piece ends 64 bytes short of what is there.
      // k_cross_links (it only looks one segment back, which the piece before has finished on the same stream) joins them when the
      // pieces come slower than the chip takes them -- a piece of k_prev_links is 1.5 ms, with k_cross_links 2.2: on a host that
      // delivers 64 MiB every 1.7 ms it would only make the chip the slower side (measured on two boxes: 49.7 against 48.6 ms for the
      // link stage on a fast host, ten milliseconds the other way on a slow one), so it is launched over everything that is still open
      // whenever a piece took 2 ms and more to come, and at the end.  k_cross_dist always waits for the end: piece by piece it takes twice its
      // time (1.2 - 1.5 ms per 64 MiB against 0.68: the few long walks at the end of every launch, sixteen times instead of once --
      // tests/prof_trace_hostpath.sh).
      constexpr uint32_t PIECE = 2048;
      static_assert(PIECE % 64 == 0, "pieces start at multiples of every run length (a power of two up to 64)");
      constexpr double SLOW_PIECE_S = 2.0e-3;
      uint32_t cl_from = 1;                                              // first segment whose cross links are still to be made (segment 0 has nothing before it)
      auto cross_links_upto = [&](uint32_t s1) {
        if (s1 > cl_from) cross_links(cl_from, s1);
        if (s1 > cl_from) cl_from = s1;
      };
      auto t_prev = std::chrono::steady_clock::now();
      for (uint32_t s0 = 0; s0 < nseg;) {
        const uint32_t s1 = s0 + PIECE < nseg ? s0 + PIECE : nseg;
        const uint64_t upto = (uint64_t)s1 * 32768 + 64 < n ? (uint64_t)s1 * 32768 + 64 : n;
        if (int rn = job.need(upto)) return rn;
        const auto t_now = std::chrono::steady_clock::now();
        const bool slow = s0 > 0 && std::chrono::duration<double>(t_now - t_prev).count() >= SLOW_PIECE_S;
        t_prev = t_now;
        prev_links(s0, s1);
#ifndef ZADA_OLD_INIT
        if (s0 == 0) hipStreamWaitEvent(st, c->ev_dlim, 0);              // (k_bucket_limits writes into the plane the second stream has preset)
#endif
        if (slow) cross_links_upto(s1);
        hipLaunchKernelGGL(k_bucket_limits, dim3(s1 - s0), dim3(256), 0, st, L, cfg.chain, cfg.chain >> 2, W.S3, W.bsc3, W.dlim, W.dlim_bits, W.segmax, W.heavy, s0);
        s0 = s1;
      }
      if (int rn = job.need(n)) return rn;
      cross_links_upto(nseg);
      if (nseg > 1) cross_dist();
    } else
    prev_links(0, nseg);
#ifndef ZADA_OLD_INIT
    if (!job.need) hipStreamWaitEvent(st, c->ev_dlim, 0);
#endif
#ifdef ZADA_PL_STATS
    { unsigned long long h[32]; hipMemcpy(h, W.dbg, sizeof h, hipMemcpyDeviceToHost); fprintf(stderr, "[prev_links cycles/segment] init %.0f |", (double)h[8] / nseg); for (int q = 9; q < 9 + 4 * (NLEVELS + 1) - 1; q++) fprintf(stderr, " %.0f", (double)h[q] / nseg); fprintf(stderr, "  (per level 3..: sort, links, first candidates, queue rounds)\n"); fprintf(stderr, "[level 4 walks per segment] rounds %.2f  entries %.1f  entries of the last round %.1f  hops there %.1f (max %llu over all segments; walks over 64 hops %.3f)  first-queue overflow %.1f  cycles of the last round %.0f\n", (double)h[24] / nseg, (double)h[25] / nseg, (double)h[26] / nseg, (double)h[27] / nseg, h[28], (double)h[29] / nseg, (double)h[30] / nseg, (double)h[31] / nseg); hipMemset(W.dbg, 0, 256);
      unsigned long long sd[8]; hipMemcpyFromSymbol(sd, HIP_SYMBOL(g_sort_dbg), sizeof sd); fprintf(stderr, "[sort_pass cycles/segment, all six passes] clear %.0f  rank (LDS atomics) %.0f  scan %.0f  scatter %.0f\n", (double)sd[0] / nseg, (double)sd[1] / nseg, (double)sd[2] / nseg, (double)sd[3] / nseg); for (int q = 0; q < 8; q++) sd[q] = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_sort_dbg), sd, sizeof sd); }
#endif
    c->tmark("prev_links");
    if (!job.need) {
      if (nseg > 1) {
        cross_links(1, nseg);
        cross_dist();
      }
      hipLaunchKernelGGL(k_bucket_limits, dim3(nseg), dim3(256), 0, st, L, cfg.chain, cfg.chain >> 2, W.S3, W.bsc3, W.dlim, W.dlim_bits, W.segmax, W.heavy, 0u);
    }
  }
#ifdef ZADA_CD_STATS
  { unsigned long long h[16]; hipDeviceSynchronize(); hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cd_dbg), sizeof h);
    fprintf(stderr, "[k_cross_dist level-4 walks] %llu walks (%.1f %% of the positions), %llu steps (%.2f per walk), found %llu (%.1f %%), no step %llu, > 8 steps %llu, > 64 steps %llu, steps of the walks that found nothing %llu, gave-up restarts %llu\n",
            h[0], 100.0 * h[0] / n, h[1], (double)h[1] / (h[0] ? h[0] : 1), h[2], 100.0 * h[2] / (h[0] ? h[0] : 1), h[3], h[4], h[5], h[6], h[7]);
    fprintf(stderr, "[k_cross_dist level-4 walks] > 32 steps %llu, > 128 %llu, > 256 %llu, > 512 %llu, > 1024 %llu, longest %llu\n", h[8], h[9], h[10], h[11], h[12], h[13]);
    for (int q = 0; q < 16; q++) h[q] = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_cd_dbg), h, sizeof h); }
#endif
  c->tmark("cross_links");
  // ---- matches and parse, demand driven ----
  // The parser only ever looks at about a third of the positions, and hardly ever at the ones with the longest
  // chains (inside long matches): 9 % of all chain steps belong to positions it lands on.  So every position
  // first gets a bounded search (exact for 85 % of them); the parse runs on that, marking the guesses it lands
  // on; those are searched to the end; parses that used a guess which turned out different are redone; and so
  // on until a parse has used exact values only.  The result is the parse over exact values, whatever the budget.
  const uint32_t nbm = (uint32_t)((n + MB - 1) / MB);
  const uint32_t nch = (uint32_t)((n + PCHUNK - 1) / PCHUNK);
  const int max_rounds = c->knob_max_demand_rounds;     // demand rounds before everything that is still a guess is searched
  // (below 2 MiB the rounds cost more launches than the bounded search saves)
  // (round 6: six rounds from 512 MiB on -- the later demand rounds got cheaper, the flat optimum of 6 - 8 moved: 5 / 6 / 7 / 8 / 10 rounds give
  // 115.5 / 115.2 / 115.4 / 115.9 / 116.0 ms per GiB, k_match 27.6 ... 37.1 against parse 32.6 ... 23.7; at 256 MiB 31.7 against 31.5 ms for 6 / 8)
  int budget_env = c->knob_budget >= 0 ? c->knob_budget : (n < (2u << 20) ? 0 : n < (512u << 20) ? 8 : 6);
  if (budget_env < 1) budget_env = 1 << 20;
  const uint32_t nbd = (uint32_t)((n + DMB - 1) / DMB);
  hipMemsetAsync(W.blk_demand, 0, (size_t)nbd * 4, st);
  hipMemsetAsync(W.dbits, 0, (size_t)nbd * (DMB / 8), st);
  hipMemsetAsync(W.n_demand, 0, 4, st);
  hipLaunchKernelGGL(k_match, dim3(nbm), dim3(1024), MATCH_LDS, st, W.in, L, W.lprev[NLEVELS - 1], dpl, W.M, cfg.nice,
                     budget_env, (unsigned long long *)W.dbg, W.lprev[0], budget_env > c->knob_inner_budget ? c->knob_inner_budget : 0);
  c->tmark("match");
  ParseIO io; io.in = W.in; io.n = n; io.M = W.M; io.cfg = cfg; io.segend = job.segend;
  DemandMarker dm; dm.M = W.M; dm.blk_demand = W.blk_demand; dm.n_demand = W.n_demand; dm.by = M_BYSPEC; dm.dbits = W.dbits;
  DemandMarker dmf = dm; dmf.by = 0;
  int rounds = 0, demand_rounds = 0;
  bool valve_used = false;
  static const bool round_stats = getenv("ZADA_ROUND_STATS") != nullptr;
  // where the parse enters the buffer: at its first byte in the fresh state (the stream starts here, or a warm-up parse
  // through the halo), or in the state the shard before ended in
  const ExitState entry0 = job.entry_known ? job.entry : ExitState{0, SYNC_F};
  const uint32_t kE = job.entry_known ? (entry0.pos / PCHUNK < nch ? entry0.pos / PCHUNK : nch) : 0u;
  hipMemsetAsync(W.n_changed + 6, 0, 4, st);
  bool restart = false, first_again = false;
  for (bool first = true;; first = false) {
    if (first_again) { first = true; first_again = false; }
    const bool beat_valid = !first;                                  // the guesses this round's speculative parse lands on carry their length to beat (list parses only)
    // speculative parse: every chunk the first time, afterwards the chunks flagged by the demand pass
    if (first) hipLaunchKernelGGL(k_parse_spec, dim3((nch + 63) / 64), dim3(64), 0, st, io, nch,
                                  W.spec_tok, W.spec_cnt, W.Fbits, W.Lbits, W.spec_exits, dm, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u);
    else {
      // (the list's length stays on the device: the grid covers the most there can be -- the host knows how many chunks the last demand pass flagged
      // at most only after a round trip it does not need -- and the waves beyond the list end at once)
      hipMemsetAsync(W.n_changed + 2, 0, 4, st);
      hipLaunchKernelGGL(k_list_flagged, dim3((nch + 1023) / 1024), dim3(1024), 0, st, W.chg, nch, W.offsets, W.n_changed + 2);
      // Short lists (the last rounds: thousands of chunks, then a hundred, then one) go to k_parse_spec_exact -- one wave per chunk, the exact search inside
      // the parse, so that the chunk is never flagged again and the loop's tail of rounds, each as long as one lane's parse whatever the number of chunks,
      // is one launch.  Long lists (round 2 re-parses two thirds of all chunks on the benchmark stream: nearly every chunk has used one guess that turned out
      // different) stay with the lane-per-chunk parse: a wave per chunk for 1.4 M chunks took 28 ms.  The list's length stays on the device: both kernels
      // are launched, and the one the length is not for ends at once.
      const uint32_t exact_max = c->knob_exact_respec > 0 ? (uint32_t)c->knob_exact_respec : 0u;
      if (exact_max) {
        WaveExact wx; wx.in = W.in; wx.L = L; wx.dp = dpl; wx.rp = rpt; wx.tailsK = W.ltails[NLEVELS - 1]; wx.M = W.M; wx.nice_cfg = cfg.nice; wx.resume = W.lprev[0];
        hipLaunchKernelGGL(k_parse_spec_exact, dim3(256 * 5), dim3(64 * PX_WAVES), 0, st, io, nch, W.spec_tok, W.spec_cnt, W.Fbits, W.Lbits, W.spec_exits,
                           wx, (const uint32_t *)W.offsets, (const uint32_t *)(W.n_changed + 2), exact_max);
      }
      DemandMarker dmt = dm; dmt.track = true;
      hipLaunchKernelGGL(k_parse_spec, dim3((nch + 63) / 64), dim3(64), 0, st, io, nch,
                         W.spec_tok, W.spec_cnt, W.Fbits, W.Lbits, W.spec_exits, dmt, (const uint32_t *)W.offsets, (const uint32_t *)(W.n_changed + 2), exact_max);
    }
    // fixpoint of the splice, from scratch: round 0 handles every chunk with the speculative exits as entries
    hipMemcpyAsync(W.true_exits, W.spec_exits, (size_t)nch * sizeof(ExitState), hipMemcpyDeviceToDevice, st);
    if (kE > 0) hipLaunchKernelGGL(k_seed_entry, dim3((kE + 255) / 256), dim3(256), 0, st, W.true_exits, kE, entry0);
    hipMemsetAsync(W.dirty[0], 1, nch, st);
    int cur = 0, it = 0;
    uint32_t ndem = 0;                                             // a parse of this round landed on a guess
    bool slow = false;                                             // the splice is crawling: stop waiting for it
    for (;;) {
      hipMemsetAsync(W.dirty[cur ^ 1], 0, nch, st);
      hipMemsetAsync(W.n_changed, 0, 4, st);
      hipMemsetAsync(W.n_changed + 1, 0xFF, 4, st);
      hipLaunchKernelGGL(k_parse_fix, dim3((nch + 63) / 64), dim3(64), 0, st, io, nch,
                         W.spec_tok, W.spec_cnt, W.Fbits, W.Lbits, W.spec_exits, W.true_exits, W.fix_tok, W.fix_cnt,
                         W.take_from, W.start_pos, W.dirty[cur], W.dirty[cur ^ 1], W.n_changed, dmf, entry0, W.fix_stride, W.n_changed + 6);
      hipLaunchKernelGGL(k_fix_forward, dim3(1), dim3(64), 0, st, io, nch, W.true_exits, W.dirty[cur ^ 1], W.n_changed);
      uint32_t changed = 0, fix_ovf = 0;
      hipMemcpyAsync(&changed, W.n_changed, 4, hipMemcpyDeviceToHost, st);
      hipMemcpyAsync(&ndem, W.n_demand, 4, hipMemcpyDeviceToHost, st);     // (with the same round trip; the last one read counts)
      if (W.fix_stride < PTOK_STRIDE) hipMemcpyAsync(&fix_ovf, W.n_changed + 6, 4, hipMemcpyDeviceToHost, st);
      if (hip_check(c, hipStreamSynchronize(st), "parse_fix")) return ZADA_E_HIP_;
      rounds++;
      if (fix_ovf) {
        // A splice wrote more tokens than its chunk's small slot holds (it ran FIX_STRIDE_SMALL tokens without meeting the speculative parse): the
        // slots get their full size -- for the rest of the workspace's life -- and the parse starts again (what the rounds so far have made exact stays).
        uint32_t *bigger = nullptr;
        const uint64_t nch_cap = W.cap_n / PCHUNK + 2;
        if (hipMalloc(&bigger, nch_cap * (uint64_t)PTOK_STRIDE * 4) != hipSuccess) { (void)hipGetLastError(); c->err = "out of device memory (splice tokens)"; return -2; }
        W.allocs.push_back(bigger);                                  // (the small array stays booked until the workspace is rebuilt)
        W.fix_tok = bigger; W.fix_stride = PTOK_STRIDE;
        c->fix_grown++;
        hipMemsetAsync(W.n_changed + 6, 0, 4, st);
        restart = true;
        break;
      }
      if (changed == 0) break;
      // A splice that needs this many rounds is advancing chunk by chunk through data in which the speculative parses
      // never meet the true one; k_fix_forward would carry it through if it were not stopped by guesses.
      if (++it >= 32 && !valve_used) { slow = true; break; }
      cur ^= 1;
    }
    if (round_stats) {                                             // ZADA_ROUND_STATS=1: what every round of the loop did (stderr; costs a round trip per round)
      uint32_t nl = 0, nmark = 0;
      if (!first) hipMemcpy(&nl, W.n_changed + 2, 4, hipMemcpyDeviceToHost);
      { std::vector<uint32_t> hb(nbd); hipMemcpy(hb.data(), W.blk_demand, (size_t)nbd * 4, hipMemcpyDeviceToHost); for (uint32_t v : hb) nmark += v != 0; }
      fprintf(stderr, "[lz round %d] %s parse of %u chunks, splice iterations so far %d, blocks with demanded positions %u of %u, demand %u%s\n", demand_rounds, first || nl > (uint32_t)(c->knob_exact_respec > 0 ? c->knob_exact_respec : 0) ? "speculative" : "exact",
              first ? nch : nl, rounds, nmark, nbd, ndem, slow ? " (slow splice)" : "");
    }
#ifdef ZADA_RESPEC_STATS
    if (!first) { unsigned long long h[4]; hipDeviceSynchronize(); hipMemcpyFromSymbol(h, HIP_SYMBOL(g_respec_dbg), sizeof h); fprintf(stderr, "[respec round %d] %llu chunks parsed again, %llu of them with the same tokens and exit as before\n", demand_rounds, h[0], h[1]); h[0] = h[1] = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_respec_dbg), h, sizeof h); }
#endif
    if (restart) { restart = false; first_again = true; continue; }
    if (ndem == 0 && !slow) break;
    demand_rounds++;
    if (demand_rounds > 1000) { c->err = "demand loop did not converge"; return ZADA_E_HIP_; }
    hipMemsetAsync(W.n_demand, 0, 4, st);
    hipMemsetAsync(W.chg, 0, nch, st);
    if ((demand_rounds == max_rounds || slow) && !valve_used) {    // enough: search everything that is still a guess
      valve_used = true;
      hipLaunchKernelGGL(k_demand_all, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, L, W.M, W.blk_demand, W.dbits);
    }
    hipLaunchKernelGGL(k_match_demand, dim3(nbd), dim3(DM_THREADS), DM_LDS, st, W.in, L, dpl, rpt, W.ltails[NLEVELS - 1], W.M, cfg.nice,
                       W.blk_demand, W.dbits, W.chg, W.spec_exits, W.lprev[0], beat_valid);
    hipMemsetAsync(W.blk_demand, 0, (size_t)nbd * 4, st);
#ifdef ZADA_DM_STATS
    { unsigned long long h[8]; hipDeviceSynchronize(); hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dm_dbg), sizeof h);
      fprintf(stderr, "[k_match_demand round %d] %llu blocks, %.1f marked positions and %.1f scans per block; cycles per block: staging %.0f, list %.0f, phase A %.0f, phase B %.0f\n", demand_rounds, h[4],
              (double)h[5] / (h[4] ? h[4] : 1), (double)h[6] / (h[4] ? h[4] : 1), (double)h[0] / (h[4] ? h[4] : 1), (double)h[1] / (h[4] ? h[4] : 1), (double)h[2] / (h[4] ? h[4] : 1), (double)h[3] / (h[4] ? h[4] : 1));
      for (int q = 0; q < 8; q++) h[q] = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_dm_dbg), h, sizeof h); }
#endif
  }
  c->demand_rounds += demand_rounds;
  c->parse_rounds += rounds;
  c->tmark("parse");
  // the shard's tokens: those of the chunks [k0, k1)
  const uint32_t k0 = job.tok_lo / PCHUNK, k1 = job.final ? nch : job.tok_hi / PCHUNK, nk = k1 - k0;
  if (nk == 0) { c->err = "empty shard"; return -1; }
  hipLaunchKernelGGL(k_tok_count, dim3((nch + 255) / 256), dim3(256), 0, st, nch, W.spec_cnt, W.fix_cnt, W.take_from, W.counts);
  exclusive_scan_u32(st, W.counts + k0, W.offsets, W.scan_sums, W.n_changed, nk);
  struct { uint32_t total; ExitState ex, warm; } h;
  h.ex = ExitState{(uint32_t)n, SYNC_F}; h.warm = ExitState{0, SYNC_F};
  hipMemcpyAsync(&h.total, W.n_changed, 4, hipMemcpyDeviceToHost, st);
  if (!job.final) hipMemcpyAsync(&h.ex, W.true_exits + (k1 - 1), sizeof(ExitState), hipMemcpyDeviceToHost, st);
  if (k0 > 0) hipMemcpyAsync(&h.warm, W.true_exits + (k0 - 1), sizeof(ExitState), hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "tok_scan")) return ZADA_E_HIP_;
  uint32_t *dst_atoms = job.dst_atoms, *dst_apos = job.dst_apos;
  if (h.total > job.cap_atoms) {                                     // more atoms than the caller guessed: it makes room, or it is an error
    if (!job.grow_atoms) { c->err = "atom array overflow"; return -2; }
    if (int rg = job.grow_atoms(h.total, &dst_atoms, &dst_apos)) return rg;
  }
  hipLaunchKernelGGL(k_tok_compact, dim3((nk + 3) / 4), dim3(256), 0, st, nk, W.spec_tok, W.spec_cnt, W.fix_tok, W.fix_cnt,
                     W.take_from, W.start_pos, W.offsets, dst_atoms, dst_apos, k0, job.apos_bias, W.fix_stride);
  c->tmark("compact");
  res->ntok = h.total; res->exit = h.ex; res->warm = h.warm;
  return hip_check(c, hipGetLastError(), "lz_shard");
}

}  // namespace zada
