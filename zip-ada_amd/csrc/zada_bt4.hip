// zada_bt4.hip -- the BT4 match finder of LZMA Level_3 (zip_lib/lz77.adb:953-1827) as a PRODUCER that runs ahead of the coder: the match
// set Read_One_and_Get_Matches (:1234-1361) returns at every position of every entry of a call, written to HBM before k_lzma_encode
// starts.  zada_bt4.h says why the sets are a function of the input alone and holds the tree walk itself (host + device text, checked on
// the CPU against the oracle's sequential matcher by tests/test_bt4_producer.py).  Here: the data-parallel frame around it.
//
//   k_bt4_keys     one lane per position of the arena (all entries of the call side by side): its three hashes (calcHashes :1061-1069),
//                  or "not inserted" (padding; positions the window schedule never inserts: the last 162 of a stream, Move_Pos :1000-1017);
//   radix sort     (zada_sort.hip) of the positions by hash 2, by hash 3, by hash 4 -- stable, so a hash's positions stay in text order
//                  and, the arena being entry after entry, in entry order;
//   k_bt4_pred     hash 2 / hash 3: the previous position of the same hash AND entry = what hash2Table / hash3Table hold when the position
//                  is reached (:1247-1251);
//   k_bt4_jobkeys  + one more stable sort, by entry: the hash-4 order becomes (entry, hash 4, position) -- an entry's buckets side by side;
//   k_bt4_flags, k_bt4_heads, k_bt4_split
//                  the runs of equal (entry, hash 4) = the buckets, each one binary tree; long buckets and short ones apart;
//   k_bt4_walk_lds an entry of up to 16 KiB (one window fill) per workgroup, its text in LDS, one LANE per bucket (bt4_begin / bt4_step) -- a
//                  walk is a chain of dependent reads (node, then the candidate's bytes).  Two launches: the short buckets with their trees
//                  in LDS as well (16-bit nodes; a step is a tenth of what it is in HBM), the long buckets with their nodes in HBM and only
//                  the text in LDS (ten entries per CU: the long chains of thousands of entries side by side);
//   k_bt4_walk     the buckets of all other entries, one LANE per bucket with the nodes in HBM: persistent lanes, a lane that finishes a
//                  position takes its bucket's next one -- or the next bucket -- while its neighbours are still on their way down; long
//                  buckets are handed out one by one through a counter, the short ones by stride.
//
// Match sets in HBM: 8 slots per position (7 matches; most positions have one to three), the rest of a longer set (up to 50 matches: one
// per hash + one per tree level, Depth_Limit = 48) in a 43-slot block of an overflow pool booked through an atomic counter.  When the pool
// is too small the walk still counts what it would have needed, and the host runs it again with a pool of that size.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/zada.h"
#include "zada_internal.h"
#include "zada_bt4.h"

namespace zada {
namespace {

constexpr uint32_t NOJOB = 0xFFFFFFFFu;
constexpr uint32_t BT_LONG = 32;                    // a bucket of this many positions or more is handed out on its own

struct Tables { const uint32_t *tile_job; const Bt4Job *jobs; const Bt4Run *runs; };

__device__ __forceinline__ int32_t ord_of(const Tables &T, uint32_t j, uint32_t p) {
  const Bt4Job &J = T.jobs[j];
  const uint32_t q = p - (uint32_t)J.in_off;
  return (int32_t)(q - bt4_run_of(T.runs + J.run_off, J.run_cnt, q)->gap);
}

__global__ void __launch_bounds__(256) k_bt4_keys(const uint8_t *__restrict__ in, uint32_t P, Tables T, uint32_t hb4, uint32_t *__restrict__ k2, uint32_t *__restrict__ k3,
                                                  uint32_t *__restrict__ k4, uint32_t *__restrict__ val) {
  __shared__ uint32_t crc[256];
  crc[threadIdx.x] = bt4_crc(threadIdx.x);
  __syncthreads();
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  uint32_t a = 1u << 10, b = 1u << 16, c = 1u << hb4;              // "not inserted": one bit above the hash, sorts behind everything
  const uint32_t j = T.tile_job[p >> 6];
  if (j != NOJOB) {
    const Bt4Job &J = T.jobs[j];
    const uint32_t q = p - (uint32_t)J.in_off;
    if (q < J.n && bt4_run_of(T.runs + J.run_off, J.run_cnt, q)->cls != 2)
      bt4_hashes(crc[in[p]], in[p + 1], in[p + 2], crc[in[p + 3]], J.hash4_mask, a, b, c);      // (an inserted position has 162 bytes after it)
  }
  k2[p] = a; k3[p] = b; k4[p] = c; val[p] = p;
}

// pred [p] = ordinal (inside its entry) of the previous inserted position with p's hash, BT4_NONE when p is the first one
__global__ void __launch_bounds__(256) k_bt4_pred(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t P, uint32_t flag, Tables T, int32_t *__restrict__ pred) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t k = keys[i];
  if (k & flag) return;
  const uint32_t p = vals[i];
  int32_t o = BT4_NONE;
  if (i > 0 && keys[i - 1] == k) {
    const uint32_t pp = vals[i - 1], j = T.tile_job[p >> 6];
    if (T.tile_job[pp >> 6] == j) o = ord_of(T, j, pp);
  }
  pred[p] = o;
}

// after the hash-4 sort: the key of the second sort = the position's entry ("not inserted" behind all entries)
// (seg_shift < 32: ONE entry at the start of the arena, walked in segments of 2 ** seg_shift positions -- the key is the segment)
__global__ void __launch_bounds__(256) k_bt4_jobkeys(uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t P, uint32_t flag, Tables T, uint32_t E, uint32_t seg_shift) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  keys[i] = (keys[i] & flag) ? E : seg_shift < 32 ? vals[i] >> seg_shift : T.tile_job[vals[i] >> 6];
}

// cnts: [0] long buckets, [1] short buckets, [2] next long bucket to hand out, [3] overflow blocks booked, [4] first "not inserted" index
// vals: the positions in (entry, hash 4, position) order, the ones that are not inserted behind; k4: their hash-4 keys BY POSITION
__device__ __forceinline__ bool bt4_is_head(const uint32_t *__restrict__ k4, const uint32_t *__restrict__ vals, uint32_t i, const Tables &T, uint32_t seg_shift) {
  if (i == 0) return true;
  const uint32_t p = vals[i], pp = vals[i - 1];
  return k4[p] != k4[pp] || T.tile_job[p >> 6] != T.tile_job[pp >> 6] || (seg_shift < 32 && (p >> seg_shift) != (pp >> seg_shift));
}
__global__ void __launch_bounds__(256) k_bt4_flags(const uint32_t *__restrict__ k4, const uint32_t *__restrict__ vals, uint32_t P, uint32_t flag, Tables T,
                                                   uint32_t *__restrict__ flags, uint32_t *__restrict__ cnts, uint32_t seg_shift) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i > P) return;
  uint32_t f = 0;
  if (i < P) {
    if (k4[vals[i]] & flag) { if (i == 0 || !(k4[vals[i - 1]] & flag)) cnts[4] = i; }
    else f = bt4_is_head(k4, vals, i, T, seg_shift) ? 1u : 0u;
  }
  flags[i] = f;                                                      // (P + 1 entries: the scan's value at P is the number of buckets)
}
__global__ void __launch_bounds__(256) k_bt4_heads(const uint32_t *__restrict__ k4, const uint32_t *__restrict__ vals, uint32_t P, uint32_t flag, Tables T,
                                                   const uint32_t *__restrict__ rank, uint32_t *__restrict__ heads, uint32_t seg_shift) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  if (k4[vals[i]] & flag) return;
  if (bt4_is_head(k4, vals, i, T, seg_shift)) heads[rank[i]] = i;
}
// what a walk needs to start on a position, side by side in the order the walks take the positions in: (position, hash-2 predecessor,
// hash-3 predecessor) -- one 16-byte load, the next position's issued while the current one is on its way down the tree
__global__ void __launch_bounds__(256) k_bt4_records(const uint32_t *__restrict__ k4, const uint32_t *__restrict__ vals, uint32_t P, uint32_t flag, const int32_t *__restrict__ d2,
                                                     const int32_t *__restrict__ d3, uint4 *__restrict__ rec) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t p = vals[i];
  if (k4[p] & flag) return;
  rec[i] = make_uint4(p, (uint32_t)d2[p], (uint32_t)d3[p], 0u);
}
// (i0, i1: the sorted indices the launch takes the buckets of -- everything, or one segment of a stream walked in segments; rank [i] = buckets before index i)
__global__ void __launch_bounds__(256) k_bt4_split(const uint32_t *__restrict__ heads, const uint32_t *__restrict__ vals, Tables T, const uint32_t *__restrict__ nruns,
                                                   uint32_t *__restrict__ cnts, uint2 *__restrict__ longs, uint2 *__restrict__ shorts, const uint32_t *__restrict__ rank, uint32_t i0, uint32_t i1) {
  const uint32_t R = *nruns, r0 = rank[i0], r1 = rank[i1];
  const uint32_t r = r0 + blockIdx.x * 256 + threadIdx.x;
  bool live = r < r1;
  uint32_t s = 0, len = 0;
  if (live) {
    s = heads[r]; len = (r + 1 < R ? heads[r + 1] : cnts[4]) - s;
    if (T.jobs[T.tile_job[vals[s] >> 6]].small) live = false;       // (k_bt4_walk_lds takes such an entry: short buckets and long ones)
  }
  const int lane = threadIdx.x & 63;
  const uint64_t lt = (1ull << lane) - 1;
  for (int cls = 0; cls < 2; cls++) {                               // one atomic per wave and list
    const bool mine = live && (cls == 0 ? len >= BT_LONG : len < BT_LONG);
    const uint64_t m = __ballot(mine);
    if (m == 0) continue;
    uint32_t base = 0;
    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cnts[cls], (uint32_t)__popcll(m));
    base = __shfl(base, (int)__builtin_ctzll(m));
    if (mine) (cls == 0 ? longs : shorts)[base + (uint32_t)__popcll(m & lt)] = make_uint2(s, len);
  }
}

struct Sets { uint8_t *cnt; uint16_t *sl; uint32_t *sd; uint16_t *ol; uint32_t *od; uint32_t ovf_cap; };

// eight bytes at a time while eight remain below the limit (nothing at or beyond a + limit / b + limit is read)
__device__ __forceinline__ int extend8(const uint8_t *in, int64_t a, int64_t b, int len, int limit) {
  while (len + 8 <= limit) {
    unsigned long long x, y;
    __builtin_memcpy(&x, in + a + len, 8);
    __builtin_memcpy(&y, in + b + len, 8);
    x ^= y;
    if (x) return len + (__builtin_ctzll(x) >> 3);
    len += 8;
  }
  while (len < limit && in[a + len] == in[b + len]) len++;
  return len;
}

// htab (a stream walked in segments, else null): the ordinal of a bucket's last position in the segments before = the root its first position
// of this segment starts from, the reference's hash4Table (lz77.adb:1247-1251); k4: the positions' hash-4 keys.
__global__ void __launch_bounds__(256) k_bt4_walk(const uint8_t *__restrict__ arena, const uint4 *__restrict__ rec, const uint2 *__restrict__ longs, const uint2 *__restrict__ shorts,
                                                  uint32_t *__restrict__ cnts, Tables T, int32_t *__restrict__ tree, Sets S, int32_t *__restrict__ htab, const uint32_t *__restrict__ k4) {
  const uint32_t nlong = cnts[0], nshort = cnts[1], nthreads = gridDim.x * 256;
  uint32_t next_short = blockIdx.x * 256 + threadIdx.x;
  bool more_long = nlong > 0, walking = false;
  uint32_t i = 0, e = 0, p = 0, blk = 0, job = NOJOB, hcur = 0;
  int32_t prev_ord = BT4_NONE;
  uint4 nxt = make_uint4(0, 0, 0, 0);                                // the record of position i (loaded while position i - 1 was walked)
  // the bucket's entry (a bucket never leaves its entry)
  const uint8_t *jin = nullptr; int32_t *jtree = nullptr; const Bt4Run *jruns = nullptr;
  uint32_t jrun_cnt = 0, joff = 0; int32_t jmax = 0;
  Bt4Run cr{0, 0, 0, 0, 2, 0};                                       // the run of the last position (a bucket's positions mostly share it: no look-up in HBM per position)
  Bt4Walk w;
  bool rec0 = false;
  auto ext = [](const uint8_t *b, int64_t x, int64_t y, int l, int lim) { return extend8(b, x, y, l, lim); };
  auto put = [&](int k, int len, uint32_t dist) {
    const size_t base = (size_t)p * BT4_INLINE;
    if (k < BT4_INLINE - 1) { S.sl[base + k] = (uint16_t)len; S.sd[base + k] = dist; return; }
    if (k == BT4_INLINE - 1) { blk = atomicAdd(&cnts[3], 1u); S.sd[base + BT4_INLINE - 1] = blk; }
    if (blk < S.ovf_cap) { const size_t o = (size_t)blk * BT4_OVF + (uint32_t)(k - (BT4_INLINE - 1)); S.ol[o] = (uint16_t)len; S.od[o] = dist; }
  };
  for (;;) {
    if (!walking) {
      if (i >= e) {                                                  // the next bucket
        bool got = false;
        uint2 d = make_uint2(0, 0);
        if (more_long) {
          const uint32_t r = atomicAdd(&cnts[2], 1u);
          if (r < nlong) { d = longs[r]; got = true; } else more_long = false;
        }
        if (!got && next_short < nshort) { d = shorts[next_short]; next_short += nthreads; got = true; }
        if (!got) break;
        i = d.x; e = d.x + d.y; prev_ord = BT4_NONE;
        nxt = rec[i];
        job = T.tile_job[nxt.x >> 6];
        const Bt4Job &J = T.jobs[job];
        joff = (uint32_t)J.in_off; jin = arena + J.in_off; jtree = tree + 2 * (size_t)J.in_off;
        jruns = T.runs + J.run_off; jrun_cnt = J.run_cnt; jmax = (int32_t)J.max_dist;
        cr.start = cr.end = 0;
        if (htab) { hcur = k4[nxt.x] & J.hash4_mask; prev_ord = htab[hcur]; }
      }
      const uint4 rc = nxt;
      if (i + 1 < e) nxt = rec[i + 1];
      p = rc.x;
      const uint32_t q = p - joff;
      if (q < cr.start || q >= cr.end) cr = *bt4_run_of(jruns, jrun_cnt, q);
      const Bt4Run *r = &cr;
      const int avail = (int)(r->W - q - 1);
      rec0 = r->cls == 0;
      bt4_begin(w, jin, q, (int32_t)(q - r->gap), rec0, avail < BT4_LOOK ? avail : BT4_LOOK, jmax, prev_ord, rec0 ? (int32_t)rc.y : BT4_NONE, rec0 ? (int32_t)rc.z : BT4_NONE, ext, put);
      walking = true;
    }
    if (bt4_step(w, Bt4TreeI32{jtree}, ext, put)) {
      if (rec0) S.cnt[p] = (uint8_t)w.count;
      prev_ord = w.ordp;
      i++;
      walking = false;
      if (htab && i >= e) htab[hcur] = prev_ord;                      // (the bucket's last position of this segment)
    }
  }
}

// An entry of up to BT4_LDS_N bytes with one window fill (positions 0 .. n - 163 are read and inserted with n - q - 1 bytes available,
// the last 162 never: bt4_schedule gives two runs): one workgroup, the entry's text in LDS.  Its buckets are the entries
// rank [sorted_off] .. rank [sorted_end] of `heads`; the lanes take them through an LDS counter.  Two launches:
//   LONGS = false  512 lanes, the SHORT buckets (fewer than BT_LONG positions) with the nodes of their trees in LDS as well (16-bit
//                  nodes: 80 KB with the text, two entries per CU) -- a walk is a chain of dependent reads, in LDS a tenth of what it is in HBM;
//   LONGS = true   64 lanes, the LONG buckets with their nodes in HBM (a bucket's nodes are nobody else's): a long bucket is one lane's
//                  chain of thousands of dependent steps, during which a workgroup of the first kind would keep its 80 KB to itself; with
//                  the text alone (16 KB) ten entries share a CU, the long buckets of 2 560 entries walk side by side, and what a step
//                  waits for is one node from HBM -- the byte comparisons stay in LDS.
constexpr int BT4_LDS_THREADS = 512;
constexpr uint32_t BT4_LDS_BYTES = 4 * (BT4_LDS_N - BT4_NICE) + BT4_LDS_N + 16, BT4_TEXT_BYTES = BT4_LDS_N + 16;
template <bool LONGS>
__global__ void __launch_bounds__(LONGS ? 64 : BT4_LDS_THREADS) k_bt4_walk_lds(const uint8_t *__restrict__ arena, const uint4 *__restrict__ rec, const uint32_t *__restrict__ heads,
                                                                           const uint32_t *__restrict__ rank, const uint32_t *__restrict__ small_jobs, uint32_t *__restrict__ cnts, Tables T,
                                                                           int32_t *__restrict__ gtree, Sets S) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int NT = LONGS ? 64 : BT4_LDS_THREADS;
  const uint32_t job = small_jobs[blockIdx.x];
  const Bt4Job J = T.jobs[job];
  const uint32_t n = J.n, nins = n - BT4_NICE;                       // inserted positions (n > 162: the host lists no other entries)
  uint16_t *tree = (uint16_t *)lds;
  uint8_t *text = lds + (LONGS ? 0 : 4 * (size_t)(BT4_LDS_N - BT4_NICE));
  uint32_t *next = (uint32_t *)(text + BT4_LDS_N);
  const uint8_t *src = arena + J.in_off;
  for (uint32_t i = threadIdx.x * 16; i < n; i += NT * 16) {         // (entries start at multiples of 64)
    if (i + 16 <= n) *(uint4 *)(text + i) = *(const uint4 *)(src + i);
    else for (uint32_t k = i; k < n; k++) text[k] = src[k];          // (nothing is read behind the entry: the caller's buffer may end there)
  }
  if (threadIdx.x == 0) next[0] = 0;
  __syncthreads();
  const uint32_t s0 = J.sorted_off, s1 = s0 + nins, h0 = rank[s0], h1 = rank[s1], joff = (uint32_t)J.in_off;
  const int32_t jmax = (int32_t)J.max_dist;
  int32_t *jtree = gtree + 2 * (size_t)J.in_off;
  auto ext = [](const uint8_t *b, int64_t x, int64_t y, int l, int lim) { return bt4_extend(b, x, y, l, lim); };
  bool walking = false;
  uint32_t i = 0, e = 0, p = 0, blk = 0;
  int32_t prev_ord = BT4_NONE;
  uint4 nxt = make_uint4(0, 0, 0, 0);                                // the record of position i (loaded while position i - 1 was walked)
  Bt4Walk w;
  auto put = [&](int k, int len, uint32_t dist) {
    const size_t base = (size_t)p * BT4_INLINE;
    if (k < BT4_INLINE - 1) { S.sl[base + k] = (uint16_t)len; S.sd[base + k] = dist; return; }
    if (k == BT4_INLINE - 1) { blk = atomicAdd(&cnts[3], 1u); S.sd[base + BT4_INLINE - 1] = blk; }
    if (blk < S.ovf_cap) { const size_t o = (size_t)blk * BT4_OVF + (uint32_t)(k - (BT4_INLINE - 1)); S.ol[o] = (uint16_t)len; S.od[o] = dist; }
  };
  for (;;) {
    if (!walking) {
      while (i >= e) {                                                // the next bucket of this kind
        const uint32_t b = h0 + atomicAdd(&next[0], 1u);
        if (b >= h1) { i = 1; e = 0; break; }
        const uint32_t hs = heads[b], he = b + 1 < h1 ? heads[b + 1] : s1;
        if ((he - hs >= BT_LONG) == LONGS) { i = hs; e = he; prev_ord = BT4_NONE; nxt = rec[i]; }
      }
      if (i >= e) break;
      const uint4 r = nxt;
      if (i + 1 < e) nxt = rec[i + 1];
      p = r.x;
      const uint32_t q = p - joff;
      const int avail = (int)(n - q - 1);
      bt4_begin(w, text, q, (int32_t)q, true, avail < BT4_LOOK ? avail : BT4_LOOK, jmax, prev_ord, (int32_t)r.y, (int32_t)r.z, ext, put);
      walking = true;
    }
    bool done;
    if constexpr (LONGS) done = bt4_step(w, Bt4TreeI32{jtree}, ext, put);
    else done = bt4_step(w, Bt4TreeU16{tree}, ext, put);
    if (done) {
      S.cnt[p] = (uint8_t)w.count;
      prev_ord = w.ordp;
      i++;
      walking = false;
    }
  }
}

// How much the coder will have to do for an entry, roughly: the matches the producer found in it (every one is a candidate that
// Get_Next_Symbol and the estimates look at) on top of its length.  The coder starts the heaviest entries first.
__global__ void __launch_bounds__(64) k_bt4_weight(const Bt4Job *__restrict__ jobs, const uint8_t *__restrict__ cnt, uint32_t *__restrict__ weight) {
  const Bt4Job J = jobs[blockIdx.x];
  uint32_t s = 0;
  for (uint32_t q = threadIdx.x * 4; q + 4 <= J.n; q += 256) {
    const uint32_t w = *(const uint32_t *)(cnt + J.in_off + q);       // (entries start at multiples of 64)
    s += (w & 255u) + ((w >> 8) & 255u) + ((w >> 16) & 255u) + (w >> 24);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (threadIdx.x == 0) weight[blockIdx.x] = s + J.n / 4;
}

struct Buf {
  void *p = nullptr; size_t cap = 0;
  template <typename T> T *as() const { return (T *)p; }
};
// what bt4_produce books per position of the arena, grow()'s slack of 1/16 included: six key / value planes of the sorts (24) and their scratch (8), the two
// predecessor planes (8), the tree (8), the inline match sets (1 + 16 + 32), flags / heads / shorts (16), the 16-byte records, the overflow pool (16)
constexpr unsigned BT4_BYTES_PER_POSITION = 156;
struct State {
  Buf tile_job, jobs, runs, k2, k3, k4, val, ks, vs, tmp, d2, d3, tree, cnt, sl, sd, ol, od, flags, heads, longs, shorts, cnts, scan, small, rec, weight;
  Buf *all[27] = {&tile_job, &jobs, &runs, &k2, &k3, &k4, &val, &ks, &vs, &tmp, &d2, &d3, &tree, &cnt, &sl, &sd, &ol, &od, &flags, &heads, &longs, &shorts, &cnts, &scan, &small, &rec, &weight};
  bool lds_attr = false;
  // a stream walked in segments (bt4_produce with `seg`, then bt4_walk_segment per segment)
  Buf htab;
  uint32_t seg_shift = 32, P = 0, hb4 = 16;
  std::vector<uint32_t> seg_start;                                   // sorted index at which segment k begins (+ the end)
  const uint32_t *order = nullptr;
  const uint8_t *arena = nullptr;
  uint32_t ovf_cap = 0;
  uint32_t seg_blocks_max = 0;                                       // the most overflow blocks one segment's walks have booked (this stream)
};
int grow(Ctx *c, Buf &b, size_t bytes) {
  if (b.p && b.cap >= bytes) return 0;
  hipStreamSynchronize(c->stream);
  if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }
  const size_t want = bytes + bytes / 16 + 4096;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { (void)hipGetLastError(); hip_check(c, e, "hipMalloc (BT4 producer)"); b.p = nullptr; return ZADA_E_NOMEM; }
  b.cap = want;
  return 0;
}

}  // namespace

void bt4_destroy(Ctx *c) {
  State *B = (State *)c->bt4;
  if (!B) return;
  for (Buf *b : B->all) if (b->p) hipFree(b->p);
  if (B->htab.p) hipFree(B->htab.p);
  delete B;
  c->bt4 = nullptr;
}

// The match sets of all Level_3 entries among `jobs` (arena: the device buffer the entries' in_off count from; P = bytes of it that
// hold entries, every entry at a multiple of 64).  On return `out` points at the sets (device memory owned by the context, valid until
// the next call).  Returns 0, ZADA_E_NOMEM, ZADA_E_INVALID (a window schedule the producer does not take) or ZADA_E_HIP.
// seg_shift < 32 (ONE entry at the start of the arena): everything but the walks -- the entry is then walked segment by segment (2 ** seg_shift
// positions each, bt4_walk_segment), so that the coder of a segment runs next to the walks of the segment after it; *nseg receives their number.
int bt4_produce(Ctx *c, const std::vector<LzmaJob> &jobs, const uint8_t *d_arena, uint64_t arena_bytes, Bt4Sets *out, std::vector<uint32_t> *weights, uint32_t seg_shift, uint32_t *nseg) {
  if (!c->bt4) c->bt4 = new State();
  State *B = (State *)c->bt4;
  hipStream_t st = c->stream;
  const uint64_t P64 = (arena_bytes + 63) & ~63ull;
  if (P64 >= (1ull << 32)) { c->err = "LZMA: a batch of 4 GiB and more is not taken at once"; return ZADA_E_TOO_LARGE; }
  const uint32_t P = (uint32_t)P64;
  {
    // The producer books everything for the whole arena at once -- BT4_BYTES_PER_POSITION per position with grow()'s slack -- and keeps it in the
    // context: an arena that cannot fit the device's free memory is refused with the limit in the message instead of failing half way with NOMEM.
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
      size_t held = 0;
      for (Buf *b : B->all) held += b->cap;
      const double need = (double)BT4_BYTES_PER_POSITION * (double)P;
      if (need > (double)fr + (double)held) {
        static thread_local char msg[200];
        snprintf(msg, sizeof msg, "LZMA_3: the match producer needs about %u bytes per input byte; %.1f GiB of entries do not fit the %.1f GiB that are free on this device (limit here: %.2f GiB per call)",
                 BT4_BYTES_PER_POSITION, P / 1073741824.0, ((double)fr + (double)held) / 1073741824.0, ((double)fr + (double)held) / BT4_BYTES_PER_POSITION / 1073741824.0);
        c->err = msg;
        return ZADA_E_TOO_LARGE;
      }
    }
  }
  std::vector<Bt4Job> hj(jobs.size());
  std::vector<Bt4Run> hr, one;
  std::vector<uint32_t> tj(P / 64, NOJOB);
  uint32_t hmax = 1u << 16, sorted = 0;
  std::vector<uint32_t> small_jobs;
  for (size_t e = 0; e < jobs.size(); e++) {
    const LzmaJob &J = jobs[e];
    Bt4Job &b = hj[e];
    memset(&b, 0, sizeof b);
    b.in_off = J.in_off; b.n = J.level == 3 ? (uint32_t)J.n : 0; b.sbs = J.sbs; b.hash4_mask = J.hash4_size - 1; b.max_dist = J.sbs - (BT4_LOOK + 2);
    b.run_off = (uint32_t)hr.size(); b.sorted_off = sorted;
    if (J.level != 3 || J.n == 0) continue;
    if ((J.in_off & 63) || J.in_off + J.n > arena_bytes) { c->err = "LZMA: entry outside the arena"; return ZADA_E_INVALID; }
    if (!bt4_schedule(J.n, J.sbs, one)) { c->err = "LZMA: window schedule not taken by the match producer"; return ZADA_E_INVALID; }
    hr.insert(hr.end(), one.begin(), one.end());
    b.run_cnt = (uint32_t)one.size();
    for (const Bt4Run &r : one) if (r.cls != 2) sorted += r.end - r.start;
    // (an entry walked in LDS belongs to the unsegmented walks: the segmented path returns before they are launched and its split drops
    // the buckets of small entries -- a 16 384-byte stream coded with lzma_segment = 13 got no match sets that way)
    if (seg_shift >= 32 && J.n <= BT4_LDS_N && J.n > (uint64_t)BT4_NICE && one.size() == 2 && one[0].cls == 0 && one[1].cls == 2 && one[0].end == J.n - BT4_NICE) {
      b.small = 1; small_jobs.push_back((uint32_t)e);
    }
    if (J.hash4_size > hmax) hmax = J.hash4_size;
    for (uint64_t t = J.in_off >> 6; t < (J.in_off + J.n + 63) >> 6; t++) tj[t] = (uint32_t)e;
  }
  if (hr.empty()) hr.push_back(Bt4Run{0, 0, 0, 0, 2, 0});
  uint32_t hb4 = 16;
  while ((1u << hb4) < hmax) hb4++;
  const size_t tmp_bytes = radix_sort_tmp_bytes(P, 4);
  const bool segmented = seg_shift < 32;
  if (segmented && (jobs.size() != 1 || jobs[0].in_off != 0 || jobs[0].level != 3)) { c->err = "LZMA: segments are for one entry"; return ZADA_E_INVALID; }
  if (c->knob_lzma_pool > 0) B->ovf_cap = (uint32_t)c->knob_lzma_pool;        // (test knob: a pool that is too small, to walk twice -- or, segmented, to grow)
  else if (B->ovf_cap < P / 16 + 1024) B->ovf_cap = P / 16 + 1024;
  B->seg_shift = seg_shift; B->P = P; B->arena = d_arena; B->seg_blocks_max = 0; c->bt4_overflow = 0;
  if (segmented) {                                                   // where every segment begins in the (segment, hash 4, position) order: the inserted positions before it
    const uint32_t ns = (uint32_t)((jobs[0].n + (1ull << seg_shift) - 1) >> seg_shift);
    B->seg_start.assign(ns + 1, 0);
    for (uint32_t k = 0; k <= ns; k++) {
      const uint64_t x = (uint64_t)k << seg_shift;
      uint64_t ins = 0;
      for (const Bt4Run &r : hr) if (r.cls != 2 && x > r.start) ins += (x < r.end ? x : r.end) - r.start;
      B->seg_start[k] = (uint32_t)ins;
    }
    if (nseg) *nseg = ns;
  }
  int rc;
  if ((rc = grow(c, B->tile_job, 4ull * (P / 64) + 64)) || (rc = grow(c, B->jobs, sizeof(Bt4Job) * hj.size() + 64)) || (rc = grow(c, B->runs, sizeof(Bt4Run) * hr.size())) ||
      (rc = grow(c, B->k2, 4ull * P)) || (rc = grow(c, B->k3, 4ull * P)) || (rc = grow(c, B->k4, 4ull * P)) || (rc = grow(c, B->val, 4ull * P)) || (rc = grow(c, B->ks, 4ull * P)) ||
      (rc = grow(c, B->vs, 4ull * P)) || (rc = grow(c, B->tmp, tmp_bytes)) || (rc = grow(c, B->d2, 4ull * P)) || (rc = grow(c, B->d3, 4ull * P)) || (rc = grow(c, B->tree, 8ull * P)) ||
      (rc = grow(c, B->cnt, P)) || (rc = grow(c, B->sl, 2ull * BT4_INLINE * P)) || (rc = grow(c, B->sd, 4ull * BT4_INLINE * P)) || (rc = grow(c, B->flags, 4ull * P + 64)) || (rc = grow(c, B->small, 4ull * small_jobs.size() + 64)) || (rc = grow(c, B->rec, 16ull * P)) ||
      (rc = grow(c, B->heads, 4ull * P)) || (rc = grow(c, B->longs, 8ull * (P / BT_LONG + 64))) || (rc = grow(c, B->shorts, 8ull * P)) || (rc = grow(c, B->cnts, 256)) ||
      (rc = grow(c, B->scan, 4ull * ((P + 1) / 1024 + 64))))
    return rc;
  if (hipMemcpyAsync(B->tile_job.p, tj.data(), 4ull * tj.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(B->jobs.p, hj.data(), sizeof(Bt4Job) * hj.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(B->runs.p, hr.data(), sizeof(Bt4Run) * hr.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
      (!small_jobs.empty() && hipMemcpyAsync(B->small.p, small_jobs.data(), 4ull * small_jobs.size(), hipMemcpyHostToDevice, st) != hipSuccess)) { hip_check(c, hipGetLastError(), "BT4 tables"); return ZADA_E_HIP; }
  hipStreamSynchronize(st);                                         // (the host vectors go out of scope)
  uint32_t *cnts = B->cnts.as<uint32_t>();
  const uint32_t cnts0[8] = {0, 0, 0, 0, P, 0, 0, 0};
  hipMemcpyAsync(cnts, cnts0, sizeof cnts0, hipMemcpyHostToDevice, st);
  hipMemsetAsync(B->cnt.p, 0, P, st);
  const Tables T{B->tile_job.as<uint32_t>(), B->jobs.as<Bt4Job>(), B->runs.as<Bt4Run>()};
  const dim3 gp((P + 255) / 256), b256(256);
  uint32_t *ks = B->ks.as<uint32_t>(), *vs = B->vs.as<uint32_t>(), *val = B->val.as<uint32_t>();
  hipLaunchKernelGGL(k_bt4_keys, gp, b256, 0, st, d_arena, P, T, hb4, B->k2.as<uint32_t>(), B->k3.as<uint32_t>(), B->k4.as<uint32_t>(), val);
  if ((rc = radix_sort_pairs(c, st, B->tmp.p, B->tmp.cap, B->k2.as<uint32_t>(), ks, val, vs, 4, P, 0, 11))) return rc;
  hipLaunchKernelGGL(k_bt4_pred, gp, b256, 0, st, ks, vs, P, 1u << 10, T, B->d2.as<int32_t>());
  if ((rc = radix_sort_pairs(c, st, B->tmp.p, B->tmp.cap, B->k3.as<uint32_t>(), ks, val, vs, 4, P, 0, 17))) return rc;
  hipLaunchKernelGGL(k_bt4_pred, gp, b256, 0, st, ks, vs, P, 1u << 16, T, B->d3.as<int32_t>());
  if ((rc = radix_sort_pairs(c, st, B->tmp.p, B->tmp.cap, B->k4.as<uint32_t>(), ks, val, vs, 4, P, 0, hb4 + 1))) return rc;
  const uint32_t E = (uint32_t)jobs.size();
  const uint32_t *order = vs;                                        // positions in (entry, hash 4, position) order
  const uint32_t nkeys = segmented ? (uint32_t)B->seg_start.size() - 1 : E;      // entries, or segments of the one entry
  if (nkeys > 1) {
    uint32_t eb = 1;
    while ((1u << eb) <= nkeys) eb++;
    hipLaunchKernelGGL(k_bt4_jobkeys, gp, b256, 0, st, ks, vs, P, 1u << hb4, T, nkeys, seg_shift);
    if ((rc = radix_sort_pairs(c, st, B->tmp.p, B->tmp.cap, ks, B->k2.as<uint32_t>(), vs, val, 4, P, 0, eb))) return rc;   // (k2 and the identity are free by now)
    order = val;
  }
  uint32_t *flags = B->flags.as<uint32_t>(), *heads = B->heads.as<uint32_t>();
  const uint32_t *k4 = B->k4.as<uint32_t>();
  hipLaunchKernelGGL(k_bt4_flags, dim3(P / 256 + 1), b256, 0, st, k4, order, P, 1u << hb4, T, flags, cnts, seg_shift);
  exclusive_scan_u32(st, flags, flags, B->scan.as<uint32_t>(), cnts + 5, P + 1);             // cnts [5] = number of buckets
  hipLaunchKernelGGL(k_bt4_heads, gp, b256, 0, st, k4, order, P, 1u << hb4, T, flags, heads, seg_shift);
  hipLaunchKernelGGL(k_bt4_records, gp, b256, 0, st, k4, order, P, 1u << hb4, B->d2.as<int32_t>(), B->d3.as<int32_t>(), B->rec.as<uint4>());
  B->order = order; B->hb4 = hb4;
  if (segmented) {
    // the walks come segment by segment (bt4_walk_segment); the buckets' roots travel from segment to segment in a table like the reference's
    if ((rc = grow(c, B->htab, 4ull * jobs[0].hash4_size)) || (rc = grow(c, B->ol, 2ull * BT4_OVF * B->ovf_cap)) || (rc = grow(c, B->od, 4ull * BT4_OVF * B->ovf_cap))) return rc;
    hipMemsetAsync(B->htab.p, 0xFF, 4ull * jobs[0].hash4_size, st);
    if (hip_check(c, hipStreamSynchronize(st), "BT4 producer (sorts)")) return ZADA_E_HIP;
    out->cnt = B->cnt.as<uint8_t>(); out->sl = B->sl.as<uint16_t>(); out->sd = B->sd.as<uint32_t>(); out->ol = B->ol.as<uint16_t>(); out->od = B->od.as<uint32_t>();
    return 0;
  }
  hipLaunchKernelGGL(k_bt4_split, gp, b256, 0, st, heads, order, T, cnts + 5, cnts, B->longs.as<uint2>(), B->shorts.as<uint2>(), flags, 0u, P);
  if (hip_check(c, hipGetLastError(), "BT4 producer (sorts)")) return ZADA_E_HIP;
  if (!small_jobs.empty() && !B->lds_attr) {
    if (hip_check(c, hipFuncSetAttribute((const void *)k_bt4_walk_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BT4_LDS_BYTES), "k_bt4_walk_lds (LDS size)")) return ZADA_E_HIP;
    B->lds_attr = true;
  }
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = grow(c, B->ol, 2ull * BT4_OVF * B->ovf_cap)) || (rc = grow(c, B->od, 4ull * BT4_OVF * B->ovf_cap))) return rc;
    const Sets S{B->cnt.as<uint8_t>(), B->sl.as<uint16_t>(), B->sd.as<uint32_t>(), B->ol.as<uint16_t>(), B->od.as<uint32_t>(), B->ovf_cap};
    const uint32_t nblk = (P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048;
    if (!small_jobs.empty()) {                                         // (the long buckets first: their chains are the critical path)
      hipLaunchKernelGGL(k_bt4_walk_lds<true>, dim3((uint32_t)small_jobs.size()), dim3(64), BT4_TEXT_BYTES, st, d_arena, B->rec.as<uint4>(), heads, flags, B->small.as<uint32_t>(), cnts, T,
                         B->tree.as<int32_t>(), S);
      hipLaunchKernelGGL(k_bt4_walk_lds<false>, dim3((uint32_t)small_jobs.size()), dim3(BT4_LDS_THREADS), BT4_LDS_BYTES, st, d_arena, B->rec.as<uint4>(), heads, flags, B->small.as<uint32_t>(), cnts, T,
                         B->tree.as<int32_t>(), S);
    }
    hipLaunchKernelGGL(k_bt4_walk, dim3(nblk), b256, 0, st, d_arena, B->rec.as<uint4>(), B->longs.as<uint2>(), B->shorts.as<uint2>(), cnts, T, B->tree.as<int32_t>(), S, (int32_t *)nullptr, k4);
    uint32_t h[8];
    hipMemcpyAsync(h, cnts, sizeof h, hipMemcpyDeviceToHost, st);
    if (hip_check(c, hipStreamSynchronize(st), "k_bt4_walk")) return ZADA_E_HIP;
    c->bt4_buckets = h[5]; c->bt4_long = h[0]; c->bt4_overflow = h[3];
    if (h[3] <= B->ovf_cap) break;
    if (attempt == 1) { c->err = "LZMA: overflow pool of the match sets"; return ZADA_E_HIP; }
    B->ovf_cap = h[3] + 1024;                                        // (the trees are rebuilt from nothing: every walk writes its nodes before anyone reads them)
    c->bt4_reruns++;
    const uint32_t again[4] = {h[0], h[1], 0, 0};
    hipMemcpyAsync(cnts, again, sizeof again, hipMemcpyHostToDevice, st);
    hipMemsetAsync(B->cnt.p, 0, P, st);
  }
  if (weights && jobs.size() > 1) {                                  // (one entry: nothing to order)
    if ((rc = grow(c, B->weight, 4ull * jobs.size() + 64))) return rc;
    weights->resize(jobs.size());
    hipLaunchKernelGGL(k_bt4_weight, dim3((uint32_t)jobs.size()), dim3(64), 0, st, B->jobs.as<Bt4Job>(), B->cnt.as<uint8_t>(), B->weight.as<uint32_t>());
    hipMemcpyAsync(weights->data(), B->weight.p, 4ull * jobs.size(), hipMemcpyDeviceToHost, st);
    if (hip_check(c, hipStreamSynchronize(st), "k_bt4_weight")) return ZADA_E_HIP;
  }
  out->cnt = B->cnt.as<uint8_t>(); out->sl = B->sl.as<uint16_t>(); out->sd = B->sd.as<uint32_t>(); out->ol = B->ol.as<uint16_t>(); out->od = B->od.as<uint32_t>();
  return 0;
}

#ifndef ZADA_SEG_NBLK
#define ZADA_SEG_NBLK 2048
#endif
// The walks of segment k of the stream bt4_produce prepared in segments, on `st` (no wait: the caller orders the coder behind them).
int bt4_walk_segment(Ctx *c, uint32_t k, hipStream_t st) {
  State *B = (State *)c->bt4;
  if (!B || B->seg_shift >= 32 || k + 1 >= B->seg_start.size()) { c->err = "LZMA: no such segment"; return ZADA_E_INVALID; }
  const uint32_t i0 = B->seg_start[k], i1 = B->seg_start[k + 1];
  if (i1 <= i0) return 0;
  uint32_t *cnts = B->cnts.as<uint32_t>();
  hipMemsetAsync(cnts, 0, 12, st);                                   // long / short buckets and the long ones' hand-out counter: this segment's
  const Tables T{B->tile_job.as<uint32_t>(), B->jobs.as<Bt4Job>(), B->runs.as<Bt4Run>()};
  hipLaunchKernelGGL(k_bt4_split, dim3((i1 - i0 + 255) / 256), dim3(256), 0, st, B->heads.as<uint32_t>(), B->order, T, cnts + 5, cnts, B->longs.as<uint2>(), B->shorts.as<uint2>(),
                     B->flags.as<uint32_t>(), i0, i1);
  const Sets S{B->cnt.as<uint8_t>(), B->sl.as<uint16_t>(), B->sd.as<uint32_t>(), B->ol.as<uint16_t>(), B->od.as<uint32_t>(), B->ovf_cap};
  const uint32_t nblk = (i1 - i0 + 255) / 256 < ZADA_SEG_NBLK ? (i1 - i0 + 255) / 256 : ZADA_SEG_NBLK;
  hipLaunchKernelGGL(k_bt4_walk, dim3(nblk), dim3(256), 0, st, B->arena, B->rec.as<uint4>(), B->longs.as<uint2>(), B->shorts.as<uint2>(), cnts, T, B->tree.as<int32_t>(), S,
                     B->htab.as<int32_t>(), B->k4.as<uint32_t>());
  return hip_check(c, hipGetLastError(), "k_bt4_walk (segment)") ? ZADA_E_HIP : 0;
}
// After the walks enqueued on `st` so far: 1 when the overflow pool of the match sets was too small for them (the sets of the last
// segment are not all there: the caller goes back to the unsegmented way), 0 when all is well.
// Round 5: the pool GROWS between the segments.  The segments book their blocks one after the other, so what the walks so far have used says what
// the next ones will (P / 16 blocks to start with was enough for the whole of a stream on silesia_mix_v1, whose long repeats have short match sets;
// on v2 a 512 MiB stream ran out at 70 % and was coded a second time from its first byte, all match sets first).  The caller comes here with no
// kernel of the stream under way on either of the context's streams (the walks of `st` are waited for here, the coder's launches are bounded and
// waited for): when less than three segments' worth of room is left, a pool of twice the size takes the blocks over; *out is pointed at it.
int bt4_segments_overflowed(Ctx *c, hipStream_t st, Bt4Sets *out) {
  State *B = (State *)c->bt4;
  uint32_t h = 0;
  hipMemcpyAsync(&h, B->cnts.as<uint32_t>() + 3, 4, hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "k_bt4_walk (segment)")) return ZADA_E_HIP;
  const uint32_t used_before = c->bt4_overflow;
  c->bt4_overflow = h;
  if (h > B->ovf_cap) return 1;
  const uint32_t this_seg = h >= used_before ? h - used_before : h;
  if (this_seg > B->seg_blocks_max) B->seg_blocks_max = this_seg;
  const uint64_t room = (uint64_t)B->ovf_cap - h, want = 3ull * B->seg_blocks_max + 1024;
  if (out && room < want && c->knob_lzma_pool_fixed == 0) {
    uint64_t cap2 = 2ull * B->ovf_cap > (uint64_t)h + 2 * want ? 2ull * B->ovf_cap : (uint64_t)h + 2 * want;
    if (cap2 > B->P) cap2 = B->P;                                    // (a position books at most one block)
    if (cap2 > B->ovf_cap) {
      hipStreamSynchronize(c->stream);
      void *nl = nullptr, *nd = nullptr;
      const size_t bl = 2ull * BT4_OVF * cap2 + 4096, bd = 4ull * BT4_OVF * cap2 + 4096;
      if (hipMalloc(&nl, bl) == hipSuccess && hipMalloc(&nd, bd) == hipSuccess) {
        hipMemcpy(nl, B->ol.p, 2ull * BT4_OVF * (size_t)(h < B->ovf_cap ? h : B->ovf_cap), hipMemcpyDeviceToDevice);
        hipMemcpy(nd, B->od.p, 4ull * BT4_OVF * (size_t)(h < B->ovf_cap ? h : B->ovf_cap), hipMemcpyDeviceToDevice);
        hipFree(B->ol.p); hipFree(B->od.p);
        B->ol.p = nl; B->ol.cap = bl; B->od.p = nd; B->od.cap = bd;
        B->ovf_cap = (uint32_t)cap2;
        out->ol = B->ol.as<uint16_t>(); out->od = B->od.as<uint32_t>();
        c->bt4_pool_grown++;
      } else {                                                       // no memory for a larger pool: go on with the one there is (it may still do)
        (void)hipGetLastError();
        if (nl) hipFree(nl);
      }
    }
  }
  return 0;
}

}  // namespace zada
