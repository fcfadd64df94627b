// zada_api.hip -- C ABI (include/zada.h), context / workspace management, CRC-32 kernel and the
// Compress_Data wrapper (Store fallback).  Host code only orchestrates: every byte of LZ77,
// Huffman and CRC arithmetic runs in the HIP kernels of zada_lz.hip / zada_huff.hip / here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <thread>
#include "../../include/zada.h"
#include "zada_internal.h"
#include "zada_bt4.h"

struct zada_ctx { zada::Ctx c; };

namespace zada {

// A worker thread of a call (the BZip2 path's second batch in flight) sets tls_err: its error text goes there, and the main thread
// copies it to the context when it joins the worker -- two threads never write the same string.
thread_local std::string *tls_err = nullptr;
int hip_check(Ctx *c, hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  if (tls_err) *tls_err = buf; else c->err = buf;
  return ZADA_E_HIP_;
}

void Ctx::tbegin() {
  marks.clear();
  timing.clear();
}
void Ctx::tmark(const char *name) {
  if (!timing_on) return;
  size_t k = marks.size();
  if (k >= ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); ev_pool.push_back(e); }
  hipEventRecord(ev_pool[k], stream);
  marks.push_back({name, ev_pool[k]});
}
void Ctx::tend() {
  if (!timing_on || marks.empty()) return;
  hipEventSynchronize(marks.back().second);
  for (size_t i = 1; i < marks.size(); i++) {
    float ms = 0;
    hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second);
    timing.push_back({marks[i].first, ms});
  }
  // counters (names start with '#'): rounds of the demand loop and of the parse splice
  timing.push_back({"#demand_rounds", (float)demand_rounds});
  timing.push_back({"#splice_rounds", (float)parse_rounds});
  timing.push_back({"#bt4_reruns", (float)bt4_reruns});
  timing.push_back({"#lzma_launches", (float)lzma_launches});
  timing.push_back({"#bt4_pool_grown", (float)bt4_pool_grown});
  timing.push_back({"#atoms_grown", (float)atoms_grown});
  timing.push_back({"#fix_grown", (float)fix_grown});
}

#ifndef ZADA_COPY_LANES
#define ZADA_COPY_LANES 4
#endif
constexpr int ARR_LANES = ZADA_COPY_LANES;                      // copy lanes (= COPY_LANES below: the same staging buffers)
// An input that arrives while the LZ stage has begun (zada_deflate on host buffers).  The copy lanes of copy_in run in the background,
// each on a stream of its own; range_lz / lz_shard ask for the bytes they are about to read (arrival_order): the caller waits until
// the lanes have SENT them, and the stream that will read them waits for the lanes' events.  The first kernel of the LZ stage
// (k_prev_links, a fifth of the step) works segment by segment, so it runs on what has come while the rest is on the link.
struct Arrival {
  Ctx *c; const uint8_t *src; uint8_t *dst; uint64_t n; int T;
  std::vector<std::thread> th;
  std::mutex m; std::condition_variable cv;
  uint64_t sent[ARR_LANES] = {};                       // pieces a lane has put on its stream
  bool failed = false;
  void lane(int t) {
    hipSetDevice(c->device);
    uint64_t i = 0;
    for (uint64_t o = (uint64_t)t * STAGE_BYTES; o < n; o += (uint64_t)T * STAGE_BYTES, i++) {
      const int b = 2 * t + (int)(i & 1);
      if (i >= 2) hipEventSynchronize(c->ev_stage[b]);             // the copy that last used this buffer has left it
      const uint64_t k = n - o < STAGE_BYTES ? n - o : STAGE_BYTES;
      memcpy(c->stage[b], src + o, k);
      const bool ok = hipMemcpyAsync(dst + o, c->stage[b], k, hipMemcpyHostToDevice, c->stream_in[t]) == hipSuccess &&
                      hipEventRecord(c->ev_stage[b], c->stream_in[t]) == hipSuccess;
      std::lock_guard<std::mutex> g(m);
      if (!ok || hipEventRecord(c->ev_in[t], c->stream_in[t]) != hipSuccess) failed = true;
      sent[t]++;
      cv.notify_all();
    }
  }
  void start() { for (int t = 0; t < T; t++) th.emplace_back([this, t] { lane(t); }); }
  void join() { for (auto &x : th) if (x.joinable()) x.join(); th.clear(); }
  ~Arrival() { join(); }
};
static bool ensure_arrival_streams(Ctx *c) {
  for (int t = 0; t < ARR_LANES; t++) {
    if (c->stream_in[t]) continue;
    if (hipStreamCreateWithFlags(&c->stream_in[t], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_in[t], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
  }
  return true;
}
// the first `upto` bytes of the arriving input: sent (the caller waits for that) and waited for by `st` (in stream order)
static int arrival_order(Ctx *c, hipStream_t st, uint64_t upto) {
  Arrival *A = (Arrival *)c->arrival;
  if (!A) return 0;
  if (upto > A->n) upto = A->n;
  const uint64_t P = (upto + STAGE_BYTES - 1) / STAGE_BYTES;      // pieces that hold them
  std::unique_lock<std::mutex> g(A->m);
  for (int t = 0; t < A->T; t++) {
    const uint64_t want = P > (uint64_t)t ? (P - 1 - t) / A->T + 1 : 0;
    if (want == 0) continue;
    A->cv.wait(g, [&] { return A->sent[t] >= want || A->failed; });
    if (A->failed) { c->err = "host to device copy of the input"; return ZADA_E_HIP; }
    if (hipStreamWaitEvent(st, c->ev_in[t], 0) != hipSuccess) { c->err = "hipStreamWaitEvent (input arrival)"; return ZADA_E_HIP; }
  }
  return 0;
}


template <typename T>
static int dalloc(Ctx *c, std::vector<void *> &group, T **p, uint64_t count) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
  if (e != hipSuccess) { hip_check(c, e, "hipMalloc"); (void)hipGetLastError(); return ZADA_E_NOMEM; }
  group.push_back(q);
  *p = (T *)q;
  return 0;
}

static void free_group(std::vector<void *> &group) {
  for (void *p : group) hipFree(p);
  group.clear();
}

// LZ stage: per byte of shard buffer 2 x 2 (links) + 8/32768 x 65536 x 2 (tails) + 2 + 1 + 4 (15-bit order) + 2 x 2 + 4 (planes) +
// 8 (match records) + 2 + 2 + 2 (last-level order) + 9 (fix-up tokens; the speculative ones share the link stage's tables' block) + ... = about 46 bytes; the buffer is one shard of a range
// (ZADA_SHARD_KIB, default 1 GiB) with its halo and tail.
int ensure_lz_workspace(Ctx *c, uint64_t nbuf) {
  Workspace &W = c->ws;
  if (W.cap_n >= nbuf && W.cap_n > 0) return 0;
  hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2);
  free_group(W.allocs);
  W.cap_n = 0;
  uint64_t cap = nbuf < (1u << 20) ? (1u << 20) : nbuf;
  cap = (cap + 65535) & ~65535ull;
  const uint64_t nch = cap / PCHUNK + 2, nseg32 = cap / 32768 + 2;
  int rc = 0;
#define A(ptr, cnt) if (!rc) rc = dalloc(c, W.allocs, &W.ptr, (cnt))
#define A2(ptr, cnt) if (!rc) rc = dalloc(c, W.allocs, &ptr, (cnt))
  A(in, cap + IN_PAD + 64);
  // What only the LINK stage reads -- the tails tables of the first hashed level (k_cross_links and the runs of k_prev_links; the last level's are read
  // by the demand pass), the 15-bit hash's sorted order, tags and bucket records (k_cross_dist, k_bucket_limits) -- is dead when the first parse starts,
  // and the speculative tokens are born there and dead again when the shard's tokens have been compacted: the two share ONE block (11 against 9 bytes
  // per input byte).  The tables' readers take whatever the block holds (uninitialised tables are checked entry by entry, the others are written whole).
  // (Who may touch the link tables: k_prev_links, k_cross_links, k_cross_dist, k_bucket_limits -- everything in front of tmark("cross_links") in lz_shard.
  // From k_match on the block holds speculative tokens; deflate_spans also uses it as scratch between spans.  With one hashed level ltails[0] would be
  // the table k_match_demand reads: the aliasing needs two.)
  static_assert(NLEVELS >= 2, "spec_tok shares the block of ltails[0], S3, T3, bsc3: the first level's tails table must be dead when the first parse starts");
  {
    const uint64_t b_tails0 = nseg32 * 65536 * 2, b_s3 = nseg32 * 32768 * 2, b_t3 = nseg32 * 32768, b_bsc = nseg32 * 32768 * 4, b_spec = nch * (uint64_t)PTOK_STRIDE * 4;
    const uint64_t links = b_tails0 + b_s3 + b_t3 + b_bsc + 1024, bytes = links > b_spec ? links : b_spec;
    uint8_t *blk = nullptr;
    A2(blk, bytes);
    if (!rc) {
      uint8_t *q = blk;
      W.ltails[0] = (uint16_t *)q; q += b_tails0 + 256;
      W.S3 = (uint16_t *)q; q += b_s3 + 256;
      W.bsc3 = (uint32_t *)q; q += b_bsc + 256;
      W.T3 = q;
      W.spec_tok = (uint32_t *)blk;
    }
  }
  for (int l = 0; l < NLEVELS; l++) { A(lprev[l], cap + IN_PAD); if (l > 0) A(ltails[l], nseg32 * 65536); }
  A(segmax, nseg32 + 16); A(heavy, nseg32 * 2048);
  A(bloom4, nseg32 * 4096);                                   // (BLOOM_WORDS of zada_lz.hip: 2^17 bits per segment)
  W.cd_cap = cap / 256 + 4096; A(cd_list, W.cd_cap + W.cd_cap / 4);    // (the list of the sweep, and behind it the list of its second pass)
  for (int l = 0; l < NLEVELS; l++) A(dplane[l], cap + 64);
  A(dlim, cap + 64); A(dlim_bits, cap / 32 + 64);
  A(M, cap + 64);
  A(SK, nseg32 * 32768); A(idxK, cap + 64); A(cntK, cap + 64);
  W.fix_stride = c->knob_fix_stride > 0 ? (uint32_t)c->knob_fix_stride : FIX_STRIDE_SMALL;
  if (W.fix_stride > PTOK_STRIDE) W.fix_stride = PTOK_STRIDE;
  A(fix_tok, nch * W.fix_stride);
  A(spec_cnt, nch); A(fix_cnt, nch); A(take_from, nch); A(start_pos, nch); A(counts, nch); A(offsets, nch);
  A(scan_sums, nch / 1024 + 1024);
  A(Fbits, cap / 32 + 64); A(Lbits, cap / 32 + 64);
  A(spec_exits, nch); A(true_exits, nch);
  A(dirty[0], nch + 64); A(dirty[1], nch + 64);
  A(n_changed, 16);
  A(blk_demand, cap / 4096 + 64); A(dbits, cap / 32 + 4096); A(n_demand, 16); A(chg, nch + 64);
  A(dbg, 128);
#undef A
#undef A2
  if (rc) { free_group(W.allocs); return rc; }
  W.cap_n = cap;
  hipMemsetAsync(W.dbg, 0, 128 * 8, c->stream);
  return hip_check(c, hipStreamSynchronize(c->stream), "workspace init");
}

// Entropy stage: sized for the atoms of one range (worst case one atom per byte) and its output.
int ensure_entropy_workspace(Ctx *c, uint64_t atoms, uint64_t flushes, uint64_t out_bytes) {
  Workspace &W = c->ws;
  if (out_bytes < atoms) out_bytes = atoms;
  const uint64_t out_need = out_bytes + out_bytes / 8 + (1u << 20);
  if (W.cap_atoms >= atoms && W.cap_flush >= flushes && W.cap_atoms > 0 && W.cap_out >= out_need) return 0;
  hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2);
  free_group(W.en_allocs);
  W.cap_atoms = 0; W.cap_flush = 0;
  uint64_t cap = atoms < (1u << 20) ? (1u << 20) : atoms;
  cap = (cap + 65535) & ~65535ull;
  // flushes: one per 65 536 atoms of a stream; a batch of small entries has (at least) one per entry
  const uint64_t nflush = (cap / FLUSH > flushes ? cap / FLUSH : flushes + flushes / 4) + 3;
  int rc = 0;
#define A(ptr, cnt) if (!rc) rc = dalloc(c, W.en_allocs, &W.ptr, (cnt))
  A(ea_atoms, LB_CAP + cap + LA_CAP + 64); A(ea_apos, LB_CAP + cap + LA_CAP + 64);
  A(descr, nflush * SLOTS * 320);
  A(seg_nblk, nflush); A(seg_cut, nflush * MAXBLK_PER_SEG); A(seg_blk_off, nflush);
  A(cut_trace, nflush * SLOTS * 2);
  W.cap_blocks = nflush * MAXBLK_PER_SEG;
  A(blocks, W.cap_blocks);
  A(binfo, W.cap_blocks);
  A(emit, W.cap_blocks);
  A(chrec, W.cap_blocks * 16 + 16);
  A(chw, W.cap_blocks * 4 + 16); A(piece_base, W.cap_blocks + 16);
  A(codes, (W.cap_blocks + 2) * 320);
  W.cap_pieces = cap / 2048 + W.cap_blocks + 64;
  A(pieces, W.cap_pieces);
  W.cap_tiles = cap / TILE + W.cap_blocks + 64;
  A(tile_block, W.cap_tiles); A(tile_bitpos, W.cap_tiles); A(tile_bits, W.cap_tiles);
  A(blk_entry, W.cap_blocks + 16);
  A(chooser, 1); A(carry, 2);
  A(scan2, nflush / 1024 + 1024); A(total2, 16);
  // the largest stream the encoder can produce for `cap` bytes: every literal in nine bits (fixed code) + block overheads
  {
    const uint64_t ob = ((out_bytes < (1u << 20) ? (1u << 20) : out_bytes) + 65535) & ~65535ull;
    W.cap_out = ob + ob / 8 + (1u << 20);
  }
  A(out, W.cap_out);
#undef A
  if (!rc) rc = ensure_crc_workspace(c, out_bytes > cap ? out_bytes : cap);
  if (rc) { free_group(W.en_allocs); return rc; }
  W.cap_atoms = cap; W.cap_flush = nflush - 3;
  return 0;
}

int ensure_crc_workspace(Ctx *c, uint64_t n) {
  Workspace &W = c->ws;
  if (W.cap_crc >= n && W.cap_crc > 0) return 0;
  hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2);
  free_group(W.crc_allocs);
  W.cap_crc = 0;
  const uint64_t cap = ((n < (1u << 20) ? (1u << 20) : n) + 65535) & ~65535ull;
  int rc = 0;
  for (int l = 0; l < 4 && !rc; l++) rc = dalloc(c, W.crc_allocs, &W.crc_lvl[l], (cap >> (4 * l)) / CRC_SUB + 64);
  if (!rc) rc = dalloc(c, W.crc_allocs, &W.crc_mat, 128);
  W.crc_mat_ready = false;
  if (rc) { free_group(W.crc_allocs); return rc; }
  W.cap_crc = cap;
  return 0;
}

int ensure_batch_workspace(Ctx *c, uint64_t entries, uint64_t fslots, uint64_t segs) {
  Workspace &W = c->ws;
  if (W.cap_ent >= entries && W.cap_fslots >= fslots && W.cap_bseg >= segs && W.cap_ent > 0) return 0;
  hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2);
  free_group(W.bt_allocs);
  W.cap_ent = W.cap_fslots = W.cap_bseg = 0;
  const uint64_t ce = entries + entries / 4 + 1024, cf = fslots + fslots / 4 + 1024, cs = segs + segs / 4 + 1024;
  int rc = 0;
#define A(ptr, cnt) if (!rc) rc = dalloc(c, W.bt_allocs, &W.ptr, (cnt))
  A(ftab, cf); A(ent_out, ce);
  A(ent_chunk0, ce + 1); A(ent_fl0, ce + 1); A(ent_start, ce + 1); A(ent_len, ce + 1); A(ent_bytes, ce + 1); A(ent_base, ce + 1); A(ent_crc, ce + 1);
  A(segend, cs + 1);
#undef A
  if (rc) { free_group(W.bt_allocs); return rc; }
  W.cap_ent = ce; W.cap_fslots = cf; W.cap_bseg = cs;
  return 0;
}

static int ensure_rin(Ctx *c, uint64_t n) {
  Workspace &W = c->ws;
  if (W.cap_rin >= n && W.rin_own) return 0;
  hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2);
  if (W.rin_own) hipFree(W.rin_own);
  W.rin_own = nullptr; W.cap_rin = 0;
  const uint64_t cap = ((n < (1u << 20) ? (1u << 20) : n) + 65535) & ~65535ull;
  if (hipMalloc((void **)&W.rin_own, cap + 256) != hipSuccess) { (void)hipGetLastError(); c->err = "hipMalloc (input)"; return ZADA_E_NOMEM; }
  W.cap_rin = cap;
  return 0;
}

static void free_workspace(Ctx *c) {
  free_group(c->ws.allocs); free_group(c->ws.en_allocs); free_group(c->ws.bt_allocs); free_group(c->ws.crc_allocs);
  if (c->ws.rin_own) hipFree(c->ws.rin_own);
  c->ws = Workspace();
}

// --------------------------------------------------------------------------------------------
// CRC-32 (zip-crc_crypto.adb:31-76): per-chunk raw registers on the GPU, GF(2) combine on the host
// --------------------------------------------------------------------------------------------
// One lane per CRC_SUB bytes (raw register started at 0: the linear part), then the 16 sub-results of
// each CRC_CHUNK are folded on the device with the fixed "advance by CRC_SUB zero bytes" operator
// (mat, 32 words), leaving one value per CRC_CHUNK for the host to chain.
// Round 5: the workgroup's 16 KiB of input are staged in LDS with coalesced 16-byte loads and every lane then reads its own 256 bytes from there
// (rows padded to 272 bytes: lanes 68 words apart, no bank conflicts beyond the four a 16-byte read has anyway).  With every lane loading its own
// 256-byte stretch from memory, 16 bytes at a time and 256 bytes apart from its neighbours', the kernel fetched 3.5 times the input (PMC FETCH_SIZE).
constexpr int CRC_WG = 64, CRC_ROW = CRC_SUB + 16;
__global__ void __launch_bounds__(CRC_WG) k_crc_chunks(const uint8_t *__restrict__ in, uint64_t n, uint32_t nsub, uint32_t *__restrict__ sub) {   // in: 16-byte aligned
  __shared__ uint32_t tab[256];
  __shared__ __attribute__((aligned(16))) uint8_t rows[CRC_WG * CRC_ROW];
  for (uint32_t t = threadIdx.x; t < 256; t += CRC_WG) {
    uint32_t l = t;
    for (int b = 0; b < 8; b++) l = (l & 1) ? (l >> 1) ^ 0xEDB88320u : l >> 1;      // Prepare_table :31-47
    tab[t] = l;
  }
  const uint64_t tile0 = (uint64_t)blockIdx.x * CRC_WG * CRC_SUB;
  {
    // (whole 16-byte words inside the input only -- a caller's device buffer ends at n --, the last few bytes one by one)
    const uint4 *src = (const uint4 *)(in + tile0);
    const uint64_t left = tile0 < n ? n - tile0 : 0, words = left / 16;
#pragma unroll
    for (int j = 0; j < CRC_SUB / 16; j++) {
      const uint32_t wq = j * CRC_WG + threadIdx.x;                               // word of the tile: row wq / 16, word wq % 16 of the row
      if (wq < words) *(uint4 *)(rows + (wq >> 4) * CRC_ROW + (wq & 15) * 16) = src[wq];
    }
    if (words < (uint64_t)CRC_WG * CRC_SUB / 16 && threadIdx.x < (left & 15)) {
      const uint64_t o = words * 16 + threadIdx.x;                                // byte of the tile
      rows[(o / CRC_SUB) * CRC_ROW + (o % CRC_SUB)] = in[tile0 + o];
    }
  }
  __syncthreads();
  uint32_t k = blockIdx.x * CRC_WG + threadIdx.x;
  if (k >= nsub) return;
  uint64_t p0 = (uint64_t)k * CRC_SUB, p1 = p0 + CRC_SUB < n ? p0 + CRC_SUB : n;
  uint32_t r = 0;
  const uint8_t *row = rows + threadIdx.x * CRC_ROW;
  const uint4 *w = (const uint4 *)row;
  uint64_t len = p1 - p0, i = 0;
  for (; i + 16 <= len; i += 16) {
    const uint4 v = w[i >> 4];
    const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint32_t x = xs[q];
      r = tab[(r ^ x) & 0xFF] ^ (r >> 8); x >>= 8;      // Update :49-60, one byte at a time
      r = tab[(r ^ x) & 0xFF] ^ (r >> 8); x >>= 8;
      r = tab[(r ^ x) & 0xFF] ^ (r >> 8); x >>= 8;
      r = tab[(r ^ x) & 0xFF] ^ (r >> 8);
    }
  }
  for (; i < len; i++) r = tab[(r ^ row[i]) & 0xFF] ^ (r >> 8);
  sub[k] = r;
}

// out[k] = fold of src[16k .. 16k+15] with the operator "advance by the length one src value covers"
__global__ void __launch_bounds__(256) k_crc_fold(const uint32_t *__restrict__ src, uint32_t ngroups, const uint32_t *__restrict__ mat,
                                                  uint32_t *__restrict__ out) {
  __shared__ uint32_t m[32];
  if (threadIdx.x < 32) m[threadIdx.x] = mat[threadIdx.x];
  __syncthreads();
  uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ngroups) return;
  uint32_t r = 0;
  for (uint32_t j = 0; j < 16; j++) {
    uint32_t s = 0, v = r;
    for (int b = 0; v; b++, v >>= 1) if (v & 1) s ^= m[b];
    r = s ^ src[k * 16 + j];
  }
  out[k] = r;
}

// operator "advance the raw register over `len` zero bytes" as a 32x32 GF(2) matrix
static void gf2_square(uint32_t *sq, const uint32_t *m) {
  for (int i = 0; i < 32; i++) { uint32_t v = m[i], s = 0; for (int j = 0; v; j++, v >>= 1) if (v & 1) s ^= m[j]; sq[i] = s; }
}
static uint32_t gf2_apply(const uint32_t *m, uint32_t v) { uint32_t s = 0; for (int j = 0; v; j++, v >>= 1) if (v & 1) s ^= m[j]; return s; }
static void zero_advance_matrix(uint64_t len, uint32_t *out) {
  uint32_t op[32], tmp[32], acc[32];
  // one zero BIT: r' = (r >> 1) ^ (r & 1 ? poly : 0)
  op[0] = 0xEDB88320u;
  for (int i = 1; i < 32; i++) op[i] = 1u << (i - 1);
  for (int i = 0; i < 32; i++) acc[i] = 1u << i;      // identity
  uint64_t bits = len * 8;
  while (bits) {
    if (bits & 1) { for (int i = 0; i < 32; i++) tmp[i] = gf2_apply(op, acc[i]); memcpy(acc, tmp, sizeof acc); }
    gf2_square(tmp, op); memcpy(op, tmp, sizeof op);
    bits >>= 1;
  }
  memcpy(out, acc, sizeof acc);
}

// CRC-32 runs on a second stream, next to the LZ stage (its kernels have no LDS and fit beside the one-workgroup-
// per-CU kernels): crc_launch queues the kernels and the copies of the few values the host has to chain,
// crc_finish waits for them and chains.
constexpr int CRC_NLEV = 4;
static uint32_t crc_M[CRC_NLEV][32];
int crc_launch(Ctx *c, const uint8_t *d_in, uint64_t n) {
  if (n == 0) return 0;
  Workspace &W = c->ws;
  CrcPending &P = c->crc;
  // level 0: one value per CRC_SUB (256 B); levels 1..3 fold 16:1 (4 KiB, 64 KiB, 1 MiB); only FULL
  // groups are folded on the device, the host chains the few leftovers of every level
  static std::once_flag mats;
  std::call_once(mats, [] { for (int l = 0; l < CRC_NLEV; l++) zero_advance_matrix((uint64_t)CRC_SUB << (4 * l), crc_M[l]); });
  hipStream_t s2 = c->stream2;
  hipEventRecord(c->ev_input, c->stream);                            // the input (and its zero pad) is in place
  hipStreamWaitEvent(s2, c->ev_input, 0);
  P.nsub = (uint32_t)((n + CRC_SUB - 1) / CRC_SUB);
  P.nfull0 = (uint32_t)(n / CRC_SUB);                                // full level-0 values
  P.cnt[0] = P.nfull0;
  for (int l = 1; l < CRC_NLEV; l++) P.cnt[l] = P.cnt[l - 1] / 16;
  if (!W.crc_mat_ready) { hipMemcpyAsync(W.crc_mat, crc_M, sizeof(uint32_t) * 32 * (CRC_NLEV - 1), hipMemcpyHostToDevice, s2); W.crc_mat_ready = true; }
  hipLaunchKernelGGL(k_crc_chunks, dim3((P.nsub + CRC_WG - 1) / CRC_WG), dim3(CRC_WG), 0, s2, d_in, n, P.nsub, W.crc_lvl[0]);
  for (int l = 1; l < CRC_NLEV; l++)
    if (P.cnt[l]) hipLaunchKernelGGL(k_crc_fold, dim3((P.cnt[l] + 255) / 256), dim3(256), 0, s2, W.crc_lvl[l - 1], P.cnt[l], W.crc_mat + 32 * (l - 1), W.crc_lvl[l]);
  // values the host needs: all of the top level, and per lower level the < 16 values after the last full group
  // (pinned host memory: a device-to-host copy into pageable memory would block this thread until the kernels are done)
  if (P.cnt[CRC_NLEV - 1] > CRC_HOST_TOP) { c->err = "crc: too many top-level values"; return ZADA_E_TOO_LARGE; }
  if (P.cnt[CRC_NLEV - 1]) hipMemcpyAsync(c->crc_host, W.crc_lvl[CRC_NLEV - 1], (size_t)P.cnt[CRC_NLEV - 1] * 4, hipMemcpyDeviceToHost, s2);
  for (int l = 0; l < CRC_NLEV - 1; l++) {
    const uint32_t first = P.cnt[l + 1] * 16;
    P.nrest[l] = (l == 0 ? P.nsub : P.cnt[l]) - first;               // level 0 includes the final short value
    if (P.nrest[l]) hipMemcpyAsync(c->crc_host + CRC_HOST_TOP + 16 * l, W.crc_lvl[l] + first, (size_t)P.nrest[l] * 4, hipMemcpyDeviceToHost, s2);
  }
  return hip_check(c, hipGetLastError(), "crc launch");
}

int crc_finish(Ctx *c, uint64_t n, uint32_t *crc_inout) {
  if (n == 0) return 0;
  constexpr int NLEV = CRC_NLEV;
  CrcPending &P = c->crc;
  const uint32_t *cnt = P.cnt, *nrest = P.nrest;
  const uint32_t nfull0 = P.nfull0;
  const uint32_t *top = c->crc_host; const uint32_t (*rest)[16] = (const uint32_t (*)[16])(c->crc_host + CRC_HOST_TOP); auto &M = crc_M;
  if (hip_check(c, hipStreamSynchronize(c->stream2), "crc")) return ZADA_E_HIP_;
  uint32_t r = *crc_inout;
  for (uint32_t k = 0; k < cnt[NLEV - 1]; k++) r = gf2_apply(M[NLEV - 1], r) ^ top[k];
  for (int l = NLEV - 2; l >= 0; l--) {
    for (uint32_t j = 0; j < nrest[l]; j++) {
      if (l == 0 && cnt[1] * 16 + j == nfull0) {                          // the final, short level-0 value
        uint32_t Ml[32];
        zero_advance_matrix(n - (uint64_t)nfull0 * CRC_SUB, Ml);
        r = gf2_apply(Ml, r) ^ rest[0][j];
      } else r = gf2_apply(M[l], r) ^ rest[l][j];
    }
  }
  *crc_inout = r;
  return 0;
}

static int method_level(int method) {
  switch (method) {
    case ZADA_DEFLATE_FIXED: return 4;       // LZ77_choice, zip-compress-deflate.adb:1573-1579
    case ZADA_DEFLATE_0: return 0;
    case ZADA_DEFLATE_1: return 6;
    case ZADA_DEFLATE_2: return 8;
    case ZADA_DEFLATE_3: return 10;
    default: return -1;
  }
}

uint32_t crc32_advance(uint32_t reg, uint64_t len) {          // the register after `len` more zero bytes (the linear part of Update)
  if (len == 0) return reg;
  uint32_t m[32];
  zero_advance_matrix(len, m);
  return gf2_apply(m, reg);
}

// --------------------------------------------------------------------------------------------
// Ranges.  A stream is compressed as one or more RANGES (one per GPU when several share a stream; one for the whole
// stream otherwise), a range in SHARDS (what the LZ workspace holds at a time).  Nothing of this changes a byte:
//   * the match finder at position p needs the bytes [p - 32 506, p + 258) only (lz77.adb:495-500), so a shard is
//     searched in a buffer that starts 32 KiB before it and ends 4 KiB behind it;
//   * the parser (lz77.adb:827-933) enters a shard in the history-free state the shard before ended in (first state
//     at or beyond the boundary: the parse runs over the boundary by less than 520 bytes, inside the tail).  Where that
//     state is not known yet (a range on another GPU) the parse is started 32 KiB earlier in the fresh state: two parses
//     that reach the same history-free state at the same position are identical from there on, which is checked when
//     the neighbour's state arrives (zada_range_lz is run again with it otherwise);
//   * the LZ buffer is flushed every 65 536 atoms counted from the start of the stream (zip-compress-deflate.adb:
//     1424-1432): a range owns the flushes whose first atom is one of its own and gets the atoms it lacks (2 048 behind,
//     up to 65 535 ahead) from its neighbours;
//   * Send_as_block's state (curr_descr, last_block_type, block_to_finish, last_block_marked, bit position:
//     zip-compress-deflate.adb:722, 993-997) is handed from range to range (ChooserCarry).
// --------------------------------------------------------------------------------------------
constexpr uint32_t SHARD_HALO = 32768, SHARD_TAIL = 4096;
constexpr uint64_t RANGE_POST = 1u << 20;    // bytes to keep resident behind a range: the look-ahead atoms of a block that may be
                                             // stored are at most 65 535 x 14 bytes (:1093-1095, 1222)

// the shard's bytes into the LZ buffer: 16 bytes per lane, four loads in flight (the runtime's copy kernel reaches a
// third of this)
__global__ void __launch_bounds__(256) k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

struct PadArgs { uint8_t *in_end; uint32_t n_in; uint16_t *link_end[NLEVELS]; };
__global__ void k_pad_init(PadArgs a) {
  for (uint32_t i = threadIdx.x; i < a.n_in; i += blockDim.x) a.in_end[i] = 0;
  for (int l = 0; l < NLEVELS; l++)
    if (threadIdx.x < 64) a.link_end[l][threadIdx.x] = 0;
}

int range_open(Ctx *c, int method, const uint8_t *rin, uint64_t stream_size, uint64_t lo, uint64_t n, uint64_t pre, uint64_t post, uint64_t carry_atoms = 0) {
  const int level = method_level(method);
  if (level < 0) { c->err = "unsupported method"; return ZADA_E_INVALID; }
  if (lo + n > stream_size || n > stream_size) { c->err = "range outside the stream"; return ZADA_E_INVALID; }
  const uint64_t behind = stream_size - (lo + n);
  // (spans of one stream on one context keep more than the halo before the range resident: the bytes of the atoms carried over)
  const bool pre_ok = pre == (lo > 0 ? SHARD_HALO : 0u) || (carry_atoms > 0 && pre <= lo && pre >= SHARD_HALO && (pre % 32768) == 0) || (lo > 0 && pre == 65536);
  if (!pre_ok || post != (behind < RANGE_POST ? behind : RANGE_POST) ||
      (lo % 65536) != 0 || (behind > 0 && (n % 65536) != 0) || (behind > 0 && n == 0)) {
    c->err = "range: boundaries must be multiples of 64 KiB, with 32 KiB before and min(1 MiB, rest of the stream) behind it resident";
    return ZADA_E_INVALID;
  }
  if (pre + n + post >= (1ull << 32) - (1ull << 26)) { c->err = "range too large for one context (4 GiB - 64 MiB): split the stream into ranges"; return ZADA_E_TOO_LARGE; }
  if (((uintptr_t)rin & 15) != 0) { c->err = "range: input must be 16-byte aligned"; return ZADA_E_INVALID; }
  const uint64_t shard = (uint64_t)c->knob_shard_kib << 10;
  // The atom arrays start with room for "atoms_pct" atoms per 100 bytes (one atom per byte is the worst case, the benchmark stream has 0.3) and grow
  // when a shard has more (range_lz: grow_atoms); everything the entropy stage sizes by atoms is the smaller for it.  A workspace that is large enough
  // already (an earlier, larger call) stays as it is.
  const uint64_t pct = c->knob_atoms_pct < 1 ? 1 : c->knob_atoms_pct > 100 ? 100 : (uint64_t)c->knob_atoms_pct;
  const uint64_t guess = n < (4u << 20) ? n : n / 100 * pct + (1u << 20);
  int rc = ensure_entropy_workspace(c, (guess < n ? guess : n) + carry_atoms, 0, n);
  if (!rc) rc = ensure_lz_workspace(c, (n < shard ? n : shard) + SHARD_HALO + SHARD_TAIL);
  if (rc) return rc;
  Range &R = c->rg;
  R = Range();
  R.T_carried = carry_atoms;
  R.open = true; R.rin = rin; R.lo = lo; R.pre = pre; R.n = n; R.post = post;
  R.first = lo == 0; R.last = behind == 0; R.method = method; R.level = level;
  c->last_nblocks = 0;
  c->demand_rounds = 0; c->parse_rounds = 0; c->atoms_grown = 0; c->fix_grown = 0;
  // the output bit stream is OR-ed together: zero it meanwhile, on the second stream
  Workspace &W = c->ws;
  const uint64_t zbytes = n + n / 8 + (1u << 20) < W.cap_out ? n + n / 8 + (1u << 20) : W.cap_out;
  hipMemsetAsync(W.out, 0, zbytes, c->stream2);
  hipEventRecord(c->ev_out, c->stream2);
  return 0;
}

// LZ stage of the whole range, shard by shard; entry = the state the range before ended in (nullptr: not known yet).
int range_lz(Ctx *c, const GlobalState *entry, zada_feedback_fn fb, void *user) {
  Range &R = c->rg;
  Workspace &W = c->ws;
  hipStream_t st = c->stream;
  if (!R.open) { c->err = "no range open"; return ZADA_E_INVALID; }
  const uint64_t shard = (uint64_t)c->knob_shard_kib << 10;
  if (shard == 0 || (shard % 65536) != 0) { c->err = "shard size must be a multiple of 64 KiB"; return ZADA_E_INVALID; }
  R.T = R.T_carried; R.placed = R.analyzed = R.chosen = false;
  R.entry_known = R.first || entry != nullptr;
  GlobalState cur{R.lo, SYNC_F, 0};
  if (!R.first && entry) cur = *entry;
  R.warm = cur;
  bool known = R.entry_known;
  // CRC-32 of the range's own bytes, on the second stream (register started from 0: the linear part) -- at once when the input is
  // resident, behind the LZ stage's first kernel when it is still arriving (Arrival: that kernel starts on what has come)
  Arrival *arr = (Arrival *)c->arrival;
  int rc = 0;
  if (!arr) {
    hipEventRecord(c->ev_input, st);
    hipStreamWaitEvent(c->stream2, c->ev_input, 0);
    if ((rc = crc_launch(c, R.rin + R.pre, R.n))) return rc;
  }
  for (uint64_t s_lo = 0; s_lo < R.n; s_lo += shard) {
    const uint64_t s_hi = s_lo + shard < R.n ? s_lo + shard : R.n;
    const uint32_t H = (R.pre + s_lo > 0) ? SHARD_HALO : 0u;
    const uint64_t after = R.n + R.post - s_hi;
    const uint32_t tail = after < SHARD_TAIL ? (uint32_t)after : SHARD_TAIL;
    const uint64_t nbuf = H + (s_hi - s_lo) + tail;
    const uint64_t boff = R.pre + s_lo - H;                     // the buffer's first byte in rin
    const uint64_t gbuf = R.lo + s_lo - H;                      // ... and in the stream
    // (boff is a multiple of 32 KiB and rin 16-byte aligned; the last 16-byte piece may read up to 15 bytes of the resident
    // tail / allocation slack behind the buffer, which the pad kernel below overwrites with zeros)
    uint64_t copied = 0;                                        // bytes of the buffer that are in W.in (or on their way, in stream order)
    if (arr) {}                                                 // (piece by piece, as lz_shard asks for them: job.need below)
    else if (nbuf >= (1u << 20) && (R.pre + R.n + R.post) - boff >= ((nbuf + 15) & ~15ull))
      hipLaunchKernelGGL(k_copy16, dim3(4096), dim3(256), 0, st, (const uint4 *)(R.rin + boff), (uint4 *)W.in, (nbuf + 15) / 16);
    else hipMemcpyAsync(W.in, R.rin + boff, nbuf, hipMemcpyDeviceToDevice, st);
    // zero pad behind the buffer and behind the link planes: one small launch
    PadArgs pa; pa.in_end = W.in + nbuf; pa.n_in = IN_PAD;
    for (int l = 0; l < NLEVELS; l++) pa.link_end[l] = W.lprev[l] + (nbuf >= 2 ? nbuf - 2 : 0);
    hipLaunchKernelGGL(k_pad_init, dim3(1), dim3(256), 0, st, pa);
    ShardJob job;
    job.nbuf = nbuf; job.tok_lo = H; job.tok_hi = (uint32_t)(H + (s_hi - s_lo));
    job.final = R.last && s_hi == R.n;
    job.entry_known = known;
    job.entry = ExitState{(uint32_t)(cur.pos - gbuf), cur.kind};
    job.dst_atoms = W.ea_atoms + LB_CAP + R.T; job.dst_apos = W.ea_apos + LB_CAP + R.T;
    job.apos_bias = (uint32_t)boff;
    job.cap_atoms = W.cap_atoms - R.T;
    if (arr) job.need = [&, boff, nbuf, pa](uint64_t upto) -> int {
      if (upto > nbuf) upto = nbuf;
      if (upto <= copied) return 0;
      if (int ra = arrival_order(c, st, boff + upto)) return ra;       // the bytes have been sent and `st` waits for them
      const uint64_t a = copied & ~15ull, b = upto == nbuf ? nbuf : upto & ~15ull;      // (whole 16-byte pieces; the last one as above)
      if (b > a) hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st, (const uint4 *)(R.rin + boff + a), (uint4 *)(W.in + a), (b - a + 15) / 16);
      if (b == nbuf && (nbuf & 15)) hipLaunchKernelGGL(k_pad_init, dim3(1), dim3(256), 0, st, pa);      // (the last 16-byte piece wrote up to 15 bytes behind the buffer: zeros again)
      copied = b;
      return 0;
    };
    job.grow_atoms = [&, s_hi](uint64_t total, uint32_t **da, uint32_t **dp) -> int {
      // room for what is there, this shard's atoms and one atom per byte of what follows: it grows once
      const uint64_t keep = LB_CAP + R.T, want = R.T + total + (R.n - s_hi) + 4096;
      uint32_t *sa = nullptr, *sp = nullptr;
      if (keep) {
        if (hipMalloc(&sa, keep * 4) != hipSuccess || hipMalloc(&sp, keep * 4) != hipSuccess) { (void)hipGetLastError(); if (sa) hipFree(sa); c->err = "out of device memory (atom arrays)"; return ZADA_E_NOMEM; }
        hipMemcpyAsync(sa, W.ea_atoms, keep * 4, hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(sp, W.ea_apos, keep * 4, hipMemcpyDeviceToDevice, st);
      }
      int rg = ensure_entropy_workspace(c, want, 0, R.n + R.T_carried);      // (waits for both streams, frees, books anew)
      if (!rg && keep) {
        hipMemcpyAsync(W.ea_atoms, sa, keep * 4, hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(W.ea_apos, sp, keep * 4, hipMemcpyDeviceToDevice, st);
      }
      if (!rg) {                                                          // the new output buffer: zeros again, as range_open left the old one
        const uint64_t zb = R.n + R.n / 8 + (1u << 20) < W.cap_out ? R.n + R.n / 8 + (1u << 20) : W.cap_out;
        hipMemsetAsync(W.out, 0, zb, c->stream2);
        hipEventRecord(c->ev_out, c->stream2);
      }
      hipStreamSynchronize(st);
      if (sa) hipFree(sa);
      if (sp) hipFree(sp);
      if (rg) return rg;
      c->atoms_grown++;
      *da = W.ea_atoms + LB_CAP + R.T; *dp = W.ea_apos + LB_CAP + R.T;
      return 0;
    };
    ShardResult res;
    rc = lz_shard(c, R.level, job, &res);
    if (rc) return rc == -2 ? ZADA_E_NOMEM : rc;
    if (s_lo == 0 && !known) R.warm = GlobalState{gbuf + res.warm.pos, res.warm.kind, 0};
    R.T += res.ntok;
    cur = GlobalState{gbuf + res.exit.pos, res.exit.kind, 0};
    known = true;
    if (fb && fb(5 + (int)(65 * s_hi / R.n), user)) return ZADA_ABORTED;
  }
  R.exit = cur;
  if (arr) {                                                    // (every byte has been asked for by now: the shards' last need)
    if ((rc = arrival_order(c, c->stream2, R.pre + R.n)) || (rc = crc_launch(c, R.rin + R.pre, R.n))) return rc;
  }
  uint32_t raw = 0;
  rc = crc_finish(c, R.n, &raw);
  if (rc) return rc;
  R.crc_raw = raw;
  return 0;
}

// Where the range lies in the stream's atom sequence, and the neighbours' atoms it needs (already in the local array).
int range_place(Ctx *c, uint64_t G, uint64_t T_total, uint32_t n_lb, uint32_t n_la) {
  Range &R = c->rg;
  if (!R.open) { c->err = "no range open"; return ZADA_E_INVALID; }
  if (G + R.T > T_total || n_lb > LB_CAP || n_la > LA_CAP) { c->err = "range_place: inconsistent atom counts"; return ZADA_E_INVALID; }
  R.G = G; R.T_total = T_total; R.n_lb = n_lb; R.n_la = n_la;
  if (R.method == ZADA_DEFLATE_FIXED) {               // one fixed block: every range codes its own atoms
    if (n_lb || n_la) { c->err = "range_place: Deflate_Fixed needs no neighbours"; return ZADA_E_INVALID; }
    R.nflush = (uint32_t)((R.T + FLUSH - 1) / FLUSH); R.foff = 0; R.j0 = 0;
    // (the chooser tells the end of the stream by stream_final = G + T == T_total)
  } else {
    const uint64_t gF0 = (G + FLUSH - 1) / FLUSH * FLUSH;
    R.nflush = G + R.T > gF0 ? (uint32_t)((G + R.T - gF0 + FLUSH - 1) / FLUSH) : 0u;
    R.foff = (uint32_t)(n_lb + (gF0 - G));
    R.j0 = gF0 / FLUSH;
    const uint64_t last_end = gF0 + (uint64_t)R.nflush * FLUSH < T_total ? gF0 + (uint64_t)R.nflush * FLUSH : T_total;
    const uint64_t need_la = R.nflush > 0 && last_end > G + R.T ? last_end - (G + R.T) : 0;
    const uint64_t need_lb = R.nflush > 0 && R.j0 > 0 && gF0 - G < HALF_SLIDER ? HALF_SLIDER - (gF0 - G) : 0;
    if (n_la < need_la || n_lb < need_lb) { c->err = "range_place: neighbours' atoms missing"; return ZADA_E_INVALID; }
    R.n_la = (uint32_t)need_la;                        // (no more than the last owned flush needs)
    if (R.nflush == 0) R.n_la = 0;
  }
  R.placed = true; R.analyzed = R.chosen = false;
  return 0;
}

// Zip.Compress.Deflate on a whole stream resident at d_in (one range): result in W.out
static int deflate_core(Ctx *c, int method, const uint8_t *d_in, uint64_t n, uint64_t *out_len, uint32_t *crc_inout, zada_feedback_fn fb, void *user) {
  if (fb && fb(0, user)) return ZADA_ABORTED;
  c->tbegin();
  c->tmark("begin");
  int rc = range_open(c, method, d_in, n, 0, n, 0, 0);
  if (rc) return rc;
  Range &R = c->rg;
  if (fb && fb(5, user)) return ZADA_ABORTED;
  rc = range_lz(c, nullptr, fb, user);
  if (rc) return rc;
  if (fb && fb(70, user)) return ZADA_ABORTED;
  rc = range_place(c, 0, R.T, 0, 0);
  if (!rc) rc = entropy_analyze(c);
  if (rc) return rc;
  hipStreamWaitEvent(c->stream, c->ev_out, 0);
  R.carry_in = ChooserCarry();
  R.carry_in.last_type = BT_RESERVED; R.carry_in.cur_eob = 7u << 16;
  rc = entropy_choose(c);
  if (rc) return rc;
  *out_len = (R.co.total_bits + 7) / 8;
  // Compression_inefficient (zip-compress.adb:479-486): the stream is not smaller than the input.  The reference stops
  // writing at the first 1 MiB flush that says so; nothing is emitted here.
  const bool inefficient = *out_len >= n;
  if (!inefficient) {
    rc = entropy_emit(c, nullptr);
    if (rc) return rc;
  }
  if (hip_check(c, hipStreamSynchronize(c->stream), "deflate")) return ZADA_E_HIP_;
  c->tmark("end");
  c->tend();
  if (crc_inout) *crc_inout = crc32_advance(*crc_inout, n) ^ R.crc_raw;
  if (fb && fb(100, user)) return ZADA_ABORTED;
  return inefficient ? ZADA_INEFFICIENT : ZADA_OK;
}

// --------------------------------------------------------------------------------------------
// Spans: a stream longer than one pass takes ("span_mib", default 2 GiB; a range is limited to 4 GiB - 64 MiB) on ONE
// context, span after span.  What a range would get from its neighbours on other GPUs is here simply kept: the parser
// state at the end of a span, the atoms behind its last complete 65 536-atom flush (they open the next span's atom array,
// with 2 048 atoms of look-behind in front) together with the bytes they stand for (the next span's input window starts at
// the first of them), Send_as_block's state and the last, partly filled byte of the output.
// --------------------------------------------------------------------------------------------
}  // namespace zada
static void copy_in(zada::Ctx *c, void *d_dst, const uint8_t *src, uint64_t n);
static int copy_out(zada::Ctx *c, uint8_t *dst, const void *d_src, uint64_t n);
namespace zada {
__global__ void k_add_u32(uint32_t *__restrict__ p, uint32_t n, uint32_t delta) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += delta;
}

static int deflate_spans(Ctx *c, int method, const uint8_t *d_src, const uint8_t *h_src, uint64_t n, uint8_t *d_dst, uint8_t *h_dst, uint64_t cap,
                         uint64_t *out_len, uint32_t *crc_inout, zada_feedback_fn fb, void *user) {
  Workspace &W = c->ws;
  hipStream_t st = c->stream;
  const uint64_t span = (uint64_t)c->knob_span_mib << 20;
  if (fb && fb(0, user)) return ZADA_ABORTED;
  c->tbegin(); c->tmark("begin");
  // (The entropy workspace is booked once, for the largest request a span can make -- range_open asks for its guess plus the carried atoms, fewer than a
  // flush --, so that it is never booked anew between spans: the carried atoms live in it.  A span with more atoms than the guess enlarges it through
  // range_lz's grow_atoms, which keeps them.)
  const uint64_t pct = c->knob_atoms_pct < 1 ? 1 : c->knob_atoms_pct > 100 ? 100 : (uint64_t)c->knob_atoms_pct;
  const uint64_t sguess = span / 100 * pct + (1u << 20);
  int rc = ensure_entropy_workspace(c, (sguess < span ? sguess : span) + FLUSH + 4096, 0, span + FLUSH + 4096);
  if (rc) return rc;
  GlobalState entry{0, SYNC_F, 0};
  uint64_t G = 0, ws_prev = 0, carry_first_byte = 0, base_bytes = 0;
  uint32_t Tc = 0, nlb = 0, crc = crc_inout ? *crc_inout : 0xFFFFFFFFu;
  uint8_t shared = 0;
  ChooserCarry cc = ChooserCarry();
  cc.last_type = BT_RESERVED; cc.cur_eob = 7u << 16;
  bool inefficient = false;
  int total_demand = 0, total_splice = 0;
  for (uint64_t lo = 0; lo < n; lo += span) {
    const uint64_t hi = lo + span < n ? lo + span : n;
    const bool last = hi == n;
    uint64_t ws = lo >= 65536 ? lo - 65536 : 0;                         // input window: 64 KiB before the span (its halo and then some) ...
    if (Tc > 0 && (carry_first_byte & ~65535ull) < ws) ws = carry_first_byte & ~65535ull;   // ... or from the first carried atom's byte
    const uint64_t post = n - hi < RANGE_POST ? n - hi : RANGE_POST;
    const uint8_t *rin;
    if (d_src) rin = d_src + ws;
    else {
      rc = ensure_rin(c, hi + post - ws);
      if (rc) return rc;
      copy_in(c, W.rin_own, h_src + ws, hi + post - ws);
      rin = W.rin_own;
    }
    rc = range_open(c, method, rin, n, lo, hi - lo, lo - ws, post, lo > 0 ? (uint64_t)Tc + 1 : 0);
    if (rc) return rc;
    Range &R = c->rg;
    R.T_carried = Tc; R.n_lb = nlb;
    // the carried atoms' positions were relative to the window before
    if (nlb + Tc > 0 && ws != ws_prev)
      hipLaunchKernelGGL(k_add_u32, dim3((nlb + Tc + 255) / 256), dim3(256), 0, st, W.ea_apos + LB_CAP - nlb, nlb + Tc, (uint32_t)(ws_prev - ws));
    rc = range_lz(c, lo > 0 ? &entry : nullptr, nullptr, nullptr);
    if (rc) return rc;
    total_demand += c->demand_rounds; total_splice += c->parse_rounds;
    // what the entropy stage takes now: whole flushes (the stream's position on the flush grid is G, a multiple of 65 536)
    const uint64_t tview = last ? R.T : R.T / FLUSH * FLUSH;
    R.G = G; R.n_la = 0; R.T_view = tview; R.T_total = last ? G + R.T : ~0ull >> 2;
    R.nflush = (uint32_t)((tview + FLUSH - 1) / FLUSH); R.foff = nlb; R.j0 = G / FLUSH; R.placed = true;
    if (R.nflush > 0 || last) {
      rc = entropy_analyze(c);
      if (rc) return rc;
      hipStreamWaitEvent(st, c->ev_out, 0);
      if (cc.pos & 7) hipMemcpyAsync(W.out, &shared, 1, hipMemcpyHostToDevice, st);       // the byte this span shares with the one before
      R.carry_in = cc;
      rc = entropy_choose(c);
      if (rc) return rc;
      cc = R.carry_out;
      const uint64_t nb = (R.co.total_bits + 7) / 8;
      if (base_bytes + nb >= n) {                       // zip-compress.adb:479-486: the reference stops here as well
        // ... but the caller Stores the entry with the CRC-32 of ALL its bytes (zip-compress.adb:224-237): this span's
        // part is known, the bytes behind it go through the CRC kernels alone, a span at a time
        inefficient = true; base_bytes += nb;
        crc = crc32_advance(crc, hi - lo) ^ R.crc_raw;
        for (uint64_t o = hi; o < n && crc_inout; o += span) {
          const uint64_t k = n - o < span ? n - o : span;
          const uint8_t *p = d_src ? d_src + o : W.rin_own;
          if (!d_src) copy_in(c, W.rin_own, h_src + o, k);            // (rin_own holds more than a span)
          if ((rc = crc_launch(c, p, k)) || (rc = crc_finish(c, k, &crc))) return rc;
        }
        break;
      }
      rc = entropy_emit(c, nullptr);
      if (rc) return rc;
      if (base_bytes + nb > cap) { c->err = "output buffer too small"; return ZADA_E_INVALID; }
      if (d_dst) hipMemcpyAsync(d_dst + base_bytes, W.out, nb, hipMemcpyDeviceToDevice, st);
      else if (copy_out(c, h_dst + base_bytes, W.out, nb)) return ZADA_E_HIP_;
      if ((cc.pos & 7) && nb) hipMemcpyAsync(&shared, W.out + nb - 1, 1, hipMemcpyDeviceToHost, st);
      if (hip_check(c, hipStreamSynchronize(st), "span out")) return ZADA_E_HIP_;
      base_bytes = cc.pos / 8;                                           // (the next span rewrites the shared byte with its bits added)
    }
    if (!last) {
      // carry: the atoms behind the last whole flush, with up to 2 048 atoms of look-behind in front of them
      const uint32_t tc2 = (uint32_t)(R.T - tview), avail = nlb + (uint32_t)tview, nlb2 = avail < LB_CAP ? avail : LB_CAP;
      const uint32_t cnt = nlb2 + tc2;
      const uint64_t src = LB_CAP + tview - nlb2, dst = LB_CAP - nlb2;
      if (src != dst && cnt) {                                           // (through a scratch array: the two places may overlap)
        hipMemcpyAsync(W.spec_tok, W.ea_atoms + src, (size_t)cnt * 4, hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(W.spec_tok + cnt, W.ea_apos + src, (size_t)cnt * 4, hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(W.ea_atoms + dst, W.spec_tok, (size_t)cnt * 4, hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(W.ea_apos + dst, W.spec_tok + cnt, (size_t)cnt * 4, hipMemcpyDeviceToDevice, st);
      }
      uint32_t first_pos = 0;
      if (tc2) hipMemcpyAsync(&first_pos, W.ea_apos + LB_CAP, 4, hipMemcpyDeviceToHost, st);
      if (hip_check(c, hipStreamSynchronize(st), "span carry")) return ZADA_E_HIP_;
      carry_first_byte = ws + first_pos;
      G += tview; Tc = tc2; nlb = nlb2;
    }
    entry = R.exit; ws_prev = ws;
    crc = crc32_advance(crc, hi - lo) ^ R.crc_raw;
    if (fb && fb((int)(100 * hi / n), user)) return ZADA_ABORTED;
  }
  c->demand_rounds = total_demand; c->parse_rounds = total_splice;
  c->tmark("end"); c->tend();
  *out_len = inefficient ? base_bytes : (cc.pos + 7) / 8;
  if (crc_inout) *crc_inout = crc;
  return (inefficient || *out_len >= n) ? ZADA_INEFFICIENT : ZADA_OK;
}

// --------------------------------------------------------------------------------------------
// Batches of small entries: ONE launch sequence for many independent streams (Zip.Create.Add_Stream is per entry,
// zip-create.adb:194-297; zipada's usual workload is many small files, tools/zipada.adb:126-134).  Every entry takes whole
// 32 KiB segments of the LZ buffer (Layout, zada_lz.hip), has its own flush grid, chooser state and place in the output
// (FlushGeom / EntOut, zada_huff.hip).  The bytes are those of one zada_deflate call per entry.
// --------------------------------------------------------------------------------------------
constexpr uint64_t BATCH_ENTRY_MAX = 4ull << 20;      // larger entries fill the GPU well enough by themselves
// (slots of one batch: knob "batch_mib", default 512; the LZ workspace takes 55 bytes per byte)

// CRC-32 of every entry (zip-crc_crypto.adb:49-60), one wave per entry: lane j takes the j-th 1/64 of the entry byte by
// byte from register 0; the pieces are chained with the GF(2) operator "advance by L zero bytes", L being the piece length,
// built by the wave itself (32 lanes hold its columns; squaring = 32 shuffles).
__device__ __forceinline__ uint32_t gf2_apply_wave(uint32_t col, uint32_t v, int lane) {       // sum of the columns j with bit j of v set
  uint32_t x = (lane < 32 && ((v >> lane) & 1u)) ? col : 0u;
  for (int off = 32; off >= 1; off >>= 1) x ^= __shfl_xor(x, off);
  return x;
}
__device__ __forceinline__ uint32_t gf2_compose_wave(uint32_t a, uint32_t b, int lane) {      // column `lane` of a o b
  uint32_t r = 0;
  for (int j = 0; j < 32; j++) { const uint32_t aj = __shfl(a, j); if ((b >> j) & 1u) r ^= aj; }
  return r;
}
__device__ uint32_t gf2_advance_cols(const uint32_t *tab, uint64_t len, int lane) {           // column `lane` of "advance by len zero bytes"
  uint32_t op = 1u << (lane & 31);
  op = tab[op & 0xFF] ^ (op >> 8);                                    // one zero byte
  uint32_t acc = 1u << (lane & 31);                                   // identity
  while (len) {
    if (len & 1) acc = gf2_compose_wave(op, acc, lane);
    op = gf2_compose_wave(op, op, lane);
    len >>= 1;
  }
  return acc;
}
__global__ void __launch_bounds__(64) k_batch_crc(uint32_t E, const uint8_t *__restrict__ in, const uint32_t *__restrict__ ent_start, const uint32_t *__restrict__ ent_len,
                                                  uint32_t *__restrict__ crc_inout) {
  __shared__ uint32_t tab[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) { uint32_t l = (uint32_t)i; for (int b = 0; b < 8; b++) l = (l & 1) ? (l >> 1) ^ 0xEDB88320u : l >> 1; tab[i] = l; }
  __syncthreads();
  const uint32_t e = blockIdx.x;
  if (e >= E) return;
  const uint32_t len = ent_len[e];
  uint32_t reg = crc_inout[e];
  if (len == 0) return;
  const uint32_t L = (len + 63) / 64, ns = (len + L - 1) / L, Llast = len - (ns - 1) * L;      // ns pieces of L bytes, the last of Llast
  const uint8_t *p = in + ent_start[e];
  uint32_t raw = 0;
  if ((uint32_t)lane < ns) {
    const uint32_t o = (uint32_t)lane * L, k = (uint32_t)lane + 1 == ns ? Llast : L;
    for (uint32_t i = 0; i < k; i++) raw = tab[(raw ^ p[o + i]) & 0xFF] ^ (raw >> 8);
  }
  const uint32_t mL = gf2_advance_cols(tab, L, lane), mLast = Llast == L ? mL : gf2_advance_cols(tab, Llast, lane);
  for (uint32_t j = 0; j < ns; j++) reg = gf2_apply_wave(j + 1 == ns ? mLast : mL, reg, lane) ^ __shfl(raw, (int)j);
  if (lane == 0) crc_inout[e] = reg;
}

static int grow_pinned(void **p, uint64_t *cap, uint64_t need) {
  if (*cap >= need && *p) return 0;
  if (*p) hipHostFree(*p);
  *p = nullptr; *cap = 0;
  const uint64_t c = need + need / 4 + (1u << 20);
  if (hipHostMalloc(p, c, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return ZADA_E_NOMEM; }
  *cap = c;
  return 0;
}

// fn(e) for e in [0, E) on a few host threads (packing and unpacking a batch is memcpy-bound: one thread moves ~8 GB/s)
template <typename F>
static void parallel_entries(uint32_t E, uint64_t bytes, F &&fn) {
  const unsigned T = bytes < (16u << 20) ? 1u : 4u;
  if (T == 1) { for (uint32_t e = 0; e < E; e++) fn(e); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; t++) th.emplace_back([&, t] { for (uint32_t e = (uint32_t)((uint64_t)E * t / T); e < (uint32_t)((uint64_t)E * (t + 1) / T); e++) fn(e); });
  for (auto &x : th) x.join();
}

// entries idx[0 .. E) of the caller's arrays through one launch sequence
static int batch_core(Ctx *c, int method, const int *idx, uint32_t E, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out,
                      const uint64_t *cap, uint64_t *out_len, uint32_t *crc, int *rc_out) {
  const int level = method_level(method);
  Workspace &W = c->ws;
  hipStream_t st = c->stream;
  // ---- layout
  std::vector<uint32_t> start(E + 1), fl0(E + 1), chunk0(E + 1), len(E + 1), crc_in(E + 1);
  uint64_t total = 0, nfs = 0;
  for (uint32_t e = 0; e < E; e++) {
    const uint64_t ln = n[idx[e]], slot = ((ln ? ln : 1) + 32767) & ~32767ull;
    start[e] = (uint32_t)total; len[e] = (uint32_t)ln; chunk0[e] = (uint32_t)(total / PCHUNK); fl0[e] = (uint32_t)nfs;
    crc_in[e] = crc ? crc[idx[e]] : 0xFFFFFFFFu;
    total += slot; nfs += ln ? (ln + FLUSH - 1) / FLUSH : 1;
  }
  start[E] = (uint32_t)total; fl0[E] = (uint32_t)nfs; chunk0[E] = (uint32_t)(total / PCHUNK); len[E] = 0;
  const uint32_t nseg = (uint32_t)(total >> 15);
  int rc = ensure_lz_workspace(c, total + 4096);
  if (!rc) rc = ensure_entropy_workspace(c, total, nfs);
  if (!rc) rc = ensure_batch_workspace(c, E, nfs, nseg);
  if (!rc) rc = grow_pinned((void **)&c->bstage, &c->cap_bstage, total + total / 8 + (1u << 20));
  const uint64_t tabw = (uint64_t)nseg + 6ull * (E + 1) + 64;
  uint64_t capw = c->cap_btab * 4;
  if (!rc) { rc = grow_pinned((void **)&c->btab, &capw, tabw * 4); c->cap_btab = capw / 4; }
  if (rc) return rc;
  // ---- pack the entries and the tables (pinned), one copy each
  uint32_t *t_seg = c->btab, *t_ent = c->btab + nseg;
  parallel_entries(E, total, [&](uint32_t e) {
    if (len[e]) memcpy(c->bstage + start[e], in[idx[e]], len[e]);
    for (uint32_t s = start[e] >> 15; s < (start[e + 1] >> 15); s++) t_seg[s] = (start[e] + len[e]) | (s == (start[e] >> 15) ? 0x80000000u : 0u);
  });
  memcpy(t_ent, chunk0.data(), (E + 1) * 4); memcpy(t_ent + (E + 1), fl0.data(), (E + 1) * 4); memcpy(t_ent + 2 * (E + 1), start.data(), (E + 1) * 4);
  memcpy(t_ent + 3 * (E + 1), len.data(), (E + 1) * 4); memcpy(t_ent + 4 * (E + 1), crc_in.data(), (E + 1) * 4);
  c->tbegin(); c->tmark("begin");
  hipMemcpyAsync(W.in, c->bstage, total, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.segend, t_seg, (size_t)nseg * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_chunk0, t_ent, (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_fl0, t_ent + (E + 1), (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_start, t_ent + 2 * (E + 1), (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_len, t_ent + 3 * (E + 1), (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_crc, t_ent + 4 * (E + 1), (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  {
    PadArgs pa; pa.in_end = W.in + total; pa.n_in = IN_PAD;
    for (int l = 0; l < NLEVELS; l++) pa.link_end[l] = W.lprev[l] + (total >= 2 ? total - 2 : 0);
    hipLaunchKernelGGL(k_pad_init, dim3(1), dim3(256), 0, st, pa);
  }
  const uint64_t zbytes = total + total / 8 + (1u << 20) < W.cap_out ? total + total / 8 + (1u << 20) : W.cap_out;
  hipMemsetAsync(W.out, 0, zbytes, c->stream2);
  hipEventRecord(c->ev_out, c->stream2);
  // ---- LZ stage of all entries
  Range &R = c->rg;
  R = Range();
  R.open = true; R.batch = true; R.n_entries = E; R.rin = W.in; R.n = total; R.method = method; R.level = level;
  c->last_nblocks = 0; c->demand_rounds = 0; c->parse_rounds = 0;
  hipLaunchKernelGGL(k_batch_crc, dim3(E), dim3(64), 0, st, E, W.in, W.ent_start, W.ent_len, W.ent_crc);
  ShardJob job;
  job.nbuf = total; job.tok_lo = 0; job.tok_hi = (uint32_t)total; job.final = true; job.entry_known = true; job.entry = ExitState{0, SYNC_F};
  job.dst_atoms = W.ea_atoms + LB_CAP; job.dst_apos = W.ea_apos + LB_CAP; job.apos_bias = 0; job.cap_atoms = W.cap_atoms; job.segend = W.segend;
  ShardResult res;
  rc = lz_shard(c, level, job, &res);
  if (rc) return rc == -2 ? ZADA_E_NOMEM : rc;
  R.T = res.ntok; R.T_total = res.ntok; R.G = 0; R.n_lb = R.n_la = 0; R.nflush = (uint32_t)nfs; R.foff = 0; R.j0 = 0; R.placed = true;
  // ---- entropy stage, every entry a stream of its own
  rc = batch_geometry(c, E, W.n_changed);            // (lz_shard left the total atom count there)
  if (!rc) rc = entropy_analyze(c);
  if (rc) return rc;
  hipStreamWaitEvent(st, c->ev_out, 0);
  rc = entropy_choose(c);
  if (!rc) rc = entropy_emit(c, nullptr);
  if (rc) return rc;
  // ---- results: sizes, places, CRCs; then the streams
  const uint64_t obytes = (R.co.total_bits + 7) / 8;
  uint32_t *h_bytes = c->btab, *h_base = c->btab + (E + 1), *h_crc = c->btab + 2 * (E + 1);
  hipMemcpyAsync(h_bytes, W.ent_bytes, (size_t)E * 4, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(h_base, W.ent_base, (size_t)E * 4, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(h_crc, W.ent_crc, (size_t)E * 4, hipMemcpyDeviceToHost, st);
  if (obytes) hipMemcpyAsync(c->bstage, W.out, obytes, hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "batch out")) return ZADA_E_HIP_;
  c->tmark("end"); c->tend();
  std::atomic<int> too_small(0);
  parallel_entries(E, obytes, [&](uint32_t e) {
    const int i = idx[e];
    out_len[i] = h_bytes[e];
    if (crc) crc[i] = h_crc[e];
    if (h_bytes[e] >= n[i]) { rc_out[i] = ZADA_INEFFICIENT; return; }         // zip-compress.adb:479-486
    if (h_bytes[e] > cap[i]) { rc_out[i] = ZADA_E_INVALID; too_small.store(1); return; }
    memcpy(out[i], c->bstage + h_base[e], h_bytes[e]);
    rc_out[i] = ZADA_OK;
  });
  if (too_small.load()) c->err = "output buffer too small";
  return 0;
}

}  // namespace zada

using namespace zada;

extern "C" {

const char *zada_version(void) { return "zada-hip 0.1 (gfx950)"; }

static void lzma_free(zada::Ctx *c);
// releases whatever a context holds (also a partly built one: every handle is tested)
static void ctx_release(zada_ctx *z) {
  if (z->c.stream) hipStreamSynchronize(z->c.stream);
  if (z->c.stream2) hipStreamSynchronize(z->c.stream2);
  bz2_destroy(&z->c);
  lzma_free(&z->c);
  free_workspace(&z->c);
  for (hipEvent_t e : z->c.ev_pool) hipEventDestroy(e);
  if (z->c.stream2) hipStreamDestroy(z->c.stream2);
  if (z->c.crc_host) hipHostFree(z->c.crc_host);
  for (int b = 0; b < 2 * MAX_COPY_LANES; b++) { if (z->c.stage[b]) hipHostFree(z->c.stage[b]); if (z->c.ev_stage[b]) hipEventDestroy(z->c.ev_stage[b]); }
  for (int t = 0; t < MAX_COPY_LANES; t++) { if (z->c.stream_in[t]) hipStreamDestroy(z->c.stream_in[t]); if (z->c.ev_in[t]) hipEventDestroy(z->c.ev_in[t]); }
  if (z->c.bstage) hipHostFree(z->c.bstage);
  if (z->c.btab) hipHostFree(z->c.btab);
  if (z->c.ev_input) hipEventDestroy(z->c.ev_input);
  if (z->c.ev_out) hipEventDestroy(z->c.ev_out);
  if (z->c.ev_dlim) hipEventDestroy(z->c.ev_dlim);
  if (z->c.stream) hipStreamDestroy(z->c.stream);
  delete z;
}

zada_ctx *zada_create(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  zada_ctx *z = new (std::nothrow) zada_ctx();
  if (!z) return nullptr;
  z->c.device = device;
  if (hipStreamCreate(&z->c.stream) != hipSuccess ||
      hipHostMalloc((void **)&z->c.crc_host, (CRC_HOST_TOP + 64) * 4, hipHostMallocDefault) != hipSuccess ||
      hipStreamCreate(&z->c.stream2) != hipSuccess || hipEventCreateWithFlags(&z->c.ev_input, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&z->c.ev_out, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&z->c.ev_dlim, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx_release(z); return nullptr; }
  // tuning knobs: read once per context
  if (const char *e = getenv("ZADA_BUDGET")) z->c.knob_budget = atoi(e);
  if (const char *e = getenv("ZADA_INNER_BUDGET")) z->c.knob_inner_budget = atoi(e);
  if (const char *e = getenv("ZADA_LINK_RUN")) { const int v = atoi(e); if (v >= 0 && v <= 64 && !(v & (v - 1))) z->c.knob_link_run = v; }
  if (const char *e = getenv("ZADA_ATOMS_PCT")) { if (atoi(e) >= 1 && atoi(e) <= 100) z->c.knob_atoms_pct = atoi(e); }
  if (const char *e = getenv("ZADA_CD_FILTER")) z->c.knob_cd_filter = atoi(e) != 0;
  if (const char *e = getenv("ZADA_EXACT_RESPEC")) { if (atoi(e) >= 0) z->c.knob_exact_respec = atoi(e); }
  if (const char *e = getenv("ZADA_MAX_DEMAND_ROUNDS")) { if (atoi(e) > 0) z->c.knob_max_demand_rounds = atoi(e); }
  if (const char *e = getenv("ZADA_SHARD_KIB")) { if (atoi(e) >= 64 && atoi(e) % 64 == 0) z->c.knob_shard_kib = atoi(e); }
  return z;
}

void zada_destroy(zada_ctx *z) {
  if (!z) return;
  hipSetDevice(z->c.device);
  ctx_release(z);
}

const char *zada_last_error(const zada_ctx *z) { return z ? z->c.err.c_str() : "no context"; }

int zada_set_knob(zada_ctx *z, const char *name, int value) {
  if (!z || !name) return ZADA_E_INVALID;
  if (!strcmp(name, "budget")) z->c.knob_budget = value;
  else if (!strcmp(name, "inner_budget")) z->c.knob_inner_budget = value;
  else if (!strcmp(name, "link_run")) { if (value < 0 || value > 64 || (value & (value - 1))) return ZADA_E_INVALID; z->c.knob_link_run = value; }
  else if (!strcmp(name, "span_mib")) { if (value < 1 || value > 3968) return ZADA_E_INVALID; z->c.knob_span_mib = value; }
  else if (!strcmp(name, "bz_batch_mib")) { if (value < 1 || value > 2048) return ZADA_E_INVALID; z->c.knob_bz_batch_mib = value; }
  else if (!strcmp(name, "bz_span_mib")) { if (value < 24 || value > 3072) return ZADA_E_INVALID; z->c.knob_bz_span_mib = value; }
  else if (!strcmp(name, "bz_batch_melems")) { if (value < 1 || value > 1536) return ZADA_E_INVALID; z->c.knob_bz_batch_melems = value; }
  else if (!strcmp(name, "bz_tail_pct")) { if (value < 0 || value > 50) return ZADA_E_INVALID; z->c.knob_bz_tail_pct = value; }
  else if (!strcmp(name, "bz_text_order")) { if (value < 0 || value > 1) return ZADA_E_INVALID; z->c.knob_bz_text_order = value; }
  else if (!strcmp(name, "bz_pipeline")) { if (value < 0 || value > 1) return ZADA_E_INVALID; z->c.knob_bz_pipeline = value; }
  else if (!strcmp(name, "bz_pipe_prio")) { if (value < 0 || value > 1) return ZADA_E_INVALID; z->c.knob_bz_pipe_prio = value; }
  else if (!strcmp(name, "bz_list_rows")) { if (value < 0 || value > 8192) return ZADA_E_INVALID; z->c.knob_bz_list_rows = value; }
  else if (!strcmp(name, "bz_split")) { if (value < 0 || value > 1) return ZADA_E_INVALID; z->c.knob_bz_split = value; }
  else if (!strcmp(name, "bz_small_wg")) { if (value < 0 || value > 1) return ZADA_E_INVALID; z->c.knob_bz_small_wg = value; }
  else if (!strcmp(name, "bz_lists")) { if (value < -1) return ZADA_E_INVALID; z->c.knob_bz_lists = value; }
  else if (!strcmp(name, "batch_mib")) { if (value < 1 || value > 1024) return ZADA_E_INVALID; z->c.knob_batch_mib = value; }
  else if (!strcmp(name, "atoms_pct")) { if (value < 1 || value > 100) return ZADA_E_INVALID; z->c.knob_atoms_pct = value; z->c.ws.cap_atoms = 0; }   // (the next call books the entropy workspace anew)
  else if (!strcmp(name, "fix_stride")) { if (value < 0 || value > (int)PTOK_STRIDE) return ZADA_E_INVALID; z->c.knob_fix_stride = value; z->c.ws.cap_n = 0; }
  else if (!strcmp(name, "cd_list_cap")) { if (value < 0) return ZADA_E_INVALID; z->c.knob_cd_list_cap = value; }
  else if (!strcmp(name, "cd_filter")) { if (value < 0 || value > 1) return ZADA_E_INVALID; z->c.knob_cd_filter = value; }
  else if (!strcmp(name, "exact_respec")) { if (value < 0) return ZADA_E_INVALID; z->c.knob_exact_respec = value; }
  else if (!strcmp(name, "max_demand_rounds")) z->c.knob_max_demand_rounds = value > 0 ? value : 12;
  else if (!strcmp(name, "shard_kib")) { if (value < 64 || value % 64) return ZADA_E_INVALID; z->c.knob_shard_kib = value; }
  else if (!strcmp(name, "lzma_dict")) { if (value < 0) return ZADA_E_INVALID; z->c.knob_lzma_dict = value; }
  // (positions per launch: -1 = the whole stream in one launch, 0 = by level; a launch of fewer than 256 positions is a host round trip
  // per handful of bytes -- the tests go down to 777)
  else if (!strcmp(name, "lzma_chunk")) { if (value < -1 || (value > 0 && value < 256)) return ZADA_E_INVALID; z->c.knob_lzma_chunk = value; }
  else if (!strcmp(name, "lzma_pool")) { if (value < 0) return ZADA_E_INVALID; z->c.knob_lzma_pool = value; }
  else if (!strcmp(name, "lzma_pool_fixed")) { if (value < 0 || value > 1) return ZADA_E_INVALID; z->c.knob_lzma_pool_fixed = value; }
  else if (!strcmp(name, "lzma_waves")) { if (value != 0 && value != 1 && value != 4) return ZADA_E_INVALID; z->c.knob_lzma_waves = value; }
  else if (!strcmp(name, "lzma_segment")) { if (value < -1 || (value > 0 && (value < 13 || value > 30))) return ZADA_E_INVALID; z->c.knob_lzma_segment = value; }
  else return ZADA_E_INVALID;
  return ZADA_OK;
}

// Large host buffers travel through two pinned staging buffers of the context (memcpy into one while the other is on
// its way): measured on the MI355X box for 1 GiB, 33 ms against 67-80 ms for hipMemcpy from pageable memory and 90 ms for
// hipHostRegister + copy (tests/probes/h2d_paths.hip).
// Host buffers travel through pinned staging buffers, 8 MiB at a time.  One thread moves ~12-25 GB/s through memcpy, the link
// takes 55 GB/s: large copies go over COPY_LANES lanes, a host thread and two staging buffers each, taking every fourth piece
// (the pieces go to their own places, so their order on the stream does not matter).
constexpr int COPY_LANES = ZADA_COPY_LANES;
static_assert(COPY_LANES <= MAX_COPY_LANES, "staging buffers");
static_assert(COPY_LANES == ARR_LANES, "the arriving input's lanes use the staging buffers of copy_in's");
constexpr uint64_t COPY_MT_MIN = 64ull << 20;
static bool ensure_staging(Ctx *c, int lanes) {
  for (int b = 0; b < 2 * lanes; b++) {
    if (c->stage[b]) continue;
    if (hipHostMalloc((void **)&c->stage[b], STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stage[b], hipEventDisableTiming) != hipSuccess) {
      if (c->stage[b]) { hipHostFree(c->stage[b]); c->stage[b] = nullptr; }
      (void)hipGetLastError();
      return false;
    }
  }
  return true;
}

// lane `t` of `T`: the pieces t, t + T, ... of the copy
static void copy_in_lane(Ctx *c, void *d_dst, const uint8_t *src, uint64_t n, int t, int T) {
  uint64_t i = 0;
  for (uint64_t o = (uint64_t)t * STAGE_BYTES; o < n; o += (uint64_t)T * STAGE_BYTES, i++) {
    const int b = 2 * t + (int)(i & 1);
    if (i >= 2) hipEventSynchronize(c->ev_stage[b]);             // the copy that last used this buffer has left it
    const uint64_t k = n - o < STAGE_BYTES ? n - o : STAGE_BYTES;
    memcpy(c->stage[b], src + o, k);
    hipMemcpyAsync((uint8_t *)d_dst + o, c->stage[b], k, hipMemcpyHostToDevice, c->stream);
    hipEventRecord(c->ev_stage[b], c->stream);
  }
}
static void copy_in(Ctx *c, void *d_dst, const uint8_t *src, uint64_t n) {
  const int T = n >= COPY_MT_MIN ? COPY_LANES : 1;
  if (n < 2 * STAGE_BYTES || !ensure_staging(c, T)) { if (n) hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, c->stream); return; }
  if (T == 1) { copy_in_lane(c, d_dst, src, n, 0, 1); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back([=] { hipSetDevice(c->device); copy_in_lane(c, d_dst, src, n, t, T); });
  copy_in_lane(c, d_dst, src, n, 0, T);
  for (auto &x : th) x.join();
}

static int copy_out_lane(Ctx *c, uint8_t *dst, const void *d_src, uint64_t n, int t, int T) {
  uint64_t i = 0, prev_o = 0, prev_k = 0;
  for (uint64_t o = (uint64_t)t * STAGE_BYTES; o < n; o += (uint64_t)T * STAGE_BYTES, i++) {
    const int b = 2 * t + (int)(i & 1);
    const uint64_t k = n - o < STAGE_BYTES ? n - o : STAGE_BYTES;
    hipMemcpyAsync(c->stage[b], (const uint8_t *)d_src + o, k, hipMemcpyDeviceToHost, c->stream);
    hipEventRecord(c->ev_stage[b], c->stream);
    if (i >= 1) {                                                 // meanwhile: the previous piece goes to the caller's buffer
      if (hipEventSynchronize(c->ev_stage[b ^ 1]) != hipSuccess) return ZADA_E_HIP_;
      memcpy(dst + prev_o, c->stage[b ^ 1], prev_k);
    }
    prev_o = o; prev_k = k;
  }
  if (i == 0) return 0;
  const int last = 2 * t + (int)((i - 1) & 1);
  if (hipEventSynchronize(c->ev_stage[last]) != hipSuccess) return ZADA_E_HIP_;
  memcpy(dst + prev_o, c->stage[last], prev_k);
  return 0;
}
static int copy_out(Ctx *c, uint8_t *dst, const void *d_src, uint64_t n) {
  const int T = n >= COPY_MT_MIN ? COPY_LANES : 1;
  if (n < 2 * STAGE_BYTES || !ensure_staging(c, T)) {
    if (n) hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, c->stream);
    return hip_check(c, hipStreamSynchronize(c->stream), "copy out");
  }
  // (the staging buffers are free: every copy_in of this call was consumed before the kernels ran)
  int rcs[COPY_LANES] = {};
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back([=, &rcs] { hipSetDevice(c->device); rcs[t] = copy_out_lane(c, dst, d_src, n, t, T); });
  rcs[0] = copy_out_lane(c, dst, d_src, n, 0, T);
  for (auto &x : th) x.join();
  for (int t = 0; t < T; t++) if (rcs[t]) { hip_check(c, hipGetLastError(), "copy out"); return ZADA_E_HIP_; }
  return 0;
}

static int prepare(zada_ctx *z) {
  if (!z) return ZADA_E_INVALID;
  if (hipSetDevice(z->c.device) != hipSuccess) return ZADA_E_HIP;
  return 0;
}

// a call that ends early (abort, error) must not leave work in flight on the context's streams
static int finish_call(Ctx *c, int rc) {
  if (rc != ZADA_OK && rc != ZADA_INEFFICIENT) { hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2); (void)hipGetLastError(); c->rg.open = false; }
  return rc;
}

int zada_deflate(zada_ctx *z, int method, const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *out_len,
                 uint32_t *crc_inout, zada_feedback_fn fb, void *user) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  if (n > ((uint64_t)c->knob_span_mib << 20)) {                   // longer than one pass takes: span after span
    uint64_t ol = 0;
    rc = finish_call(c, deflate_spans(c, method, nullptr, in, n, nullptr, out, cap, &ol, crc_inout, fb, user));
    if (out_len && rc >= 0 && rc != ZADA_ABORTED) *out_len = ol;
    return rc;
  }
  rc = ensure_rin(c, n);
  if (rc) return rc;
  uint64_t ol = 0;
  if (n >= COPY_MT_MIN && method != ZADA_DEFLATE_0 && ensure_staging(c, COPY_LANES) && ensure_arrival_streams(c)) {
    // the input goes to the device while the LZ stage's first kernel already works on what has come
    hipStreamSynchronize(c->stream);                                 // (the staging buffers and rin_own: nothing of an earlier call is in flight)
    Arrival A; A.c = c; A.src = in; A.dst = c->ws.rin_own; A.n = n; A.T = ARR_LANES;
    c->arrival = &A;
    A.start();
    rc = finish_call(c, deflate_core(c, method, c->ws.rin_own, n, &ol, crc_inout, fb, user));
    A.join();
    c->arrival = nullptr;
    for (int t = 0; t < ARR_LANES; t++) hipStreamSynchronize(c->stream_in[t]);
  } else {
    copy_in(c, c->ws.rin_own, in, n);
    rc = finish_call(c, deflate_core(c, method, c->ws.rin_own, n, &ol, crc_inout, fb, user));
  }
  if (rc < 0 || rc == ZADA_ABORTED) return rc;
  if (out_len) *out_len = ol;
  if (rc == ZADA_OK) {
    if (ol > cap) { c->err = "output buffer too small"; return ZADA_E_INVALID; }
    if (copy_out(c, out, c->ws.out, ol)) return ZADA_E_HIP;
  }
  return rc;
}

// ---- BZip2: Zip.Compress.BZip2_E, zip-compress-bzip2_e.adb:44-157 ----
static int bzip2_core(Ctx *c, int method, const uint8_t *d_in, uint64_t n, uint8_t *d_out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
                      zada_feedback_fn fb, void *user) {
  if (method < ZADA_BZIP2_1 || method > ZADA_BZIP2_3) { c->err = "not a BZip2 method"; return ZADA_E_INVALID; }
  int rc;
  c->tbegin();
  c->tmark("bz:begin");
  if (crc_inout && n) {                                            // the Zip CRC-32 of the input, 2 GiB at a time (next to nothing)
    const uint64_t piece = 2ull << 30;
    if ((rc = ensure_crc_workspace(c, n < piece ? n : piece))) return rc;
    for (uint64_t o = 0; o < n; o += piece) {
      const uint64_t k = n - o < piece ? n - o : piece;
      if ((rc = crc_launch(c, d_in + o, k)) || (rc = crc_finish(c, k, crc_inout))) return rc;
    }
  }
  rc = bz2_encode_device(c, method - ZADA_BZIP2_1, d_in, n, (int64_t)n, d_out, cap, out_len, fb, user);
  c->tmark("bz:end");
  c->tend();
  return rc;
}
int zada_bzip2_device(zada_ctx *z, int method, const void *d_in, uint64_t n, void *d_out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  const uint8_t *src = (const uint8_t *)d_in;
  if (((uintptr_t)d_in & 15) != 0 && n) {
    if ((rc = ensure_rin(c, n))) return rc;
    hipMemcpyAsync(c->ws.rin_own, d_in, n, hipMemcpyDeviceToDevice, c->stream);
    src = c->ws.rin_own;
  }
  uint64_t ol = 0;
  rc = finish_call(c, bzip2_core(c, method, src, n, (uint8_t *)d_out, cap, &ol, crc_inout, nullptr, nullptr));
  if (out_len && rc >= 0) *out_len = ol;
  return rc;
}
int zada_bzip2(zada_ctx *z, int method, const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
               zada_feedback_fn fb, void *user) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  if ((rc = ensure_rin(c, n + cap + 64))) return rc;           // input, then the stream, in one device buffer
  copy_in(c, c->ws.rin_own, in, n);
  uint8_t *d_out = c->ws.rin_own + ((n + 63) & ~63ull);
  uint64_t ol = 0;
  rc = finish_call(c, bzip2_core(c, method, c->ws.rin_own, n, d_out, cap, &ol, crc_inout, fb, user));
  if (rc < 0 || rc == ZADA_ABORTED) return rc;
  if (out_len) *out_len = ol;
  if (ol <= cap && copy_out(c, out, d_out, ol)) return ZADA_E_HIP;
  return rc;
}
// entries idx[0 .. E) of the caller's arrays, each short enough for one block, through one launch sequence
static int bzip2_batch_core(Ctx *c, int method, const int *idx, uint32_t E, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out, const uint64_t *cap,
                            uint64_t *out_len, uint32_t *crc, int *rc_out) {
  hipStream_t st = c->stream;
  std::vector<uint64_t> start(E); std::vector<uint32_t> start32(E + 1), len(E + 1), crc_in(E + 1);
  uint64_t total = 0;
  for (uint32_t e = 0; e < E; e++) {
    start[e] = total; start32[e] = (uint32_t)total; len[e] = (uint32_t)n[idx[e]]; crc_in[e] = crc ? crc[idx[e]] : 0xFFFFFFFFu;
    total += (n[idx[e]] + 63) & ~63ull;
  }
  if (total >= (1ull << 32)) return ZADA_E_TOO_LARGE;
  const uint64_t out_cap = total + total / 4 + 128ull * E + (1u << 20);
  int rc = ensure_rin(c, total + 64 + 16ull * (E + 1));
  if (!rc) rc = grow_pinned((void **)&c->bstage, &c->cap_bstage, (total > out_cap ? total : out_cap) + 64);
  if (rc) return rc;
  parallel_entries(E, total, [&](uint32_t e) { if (len[e]) memcpy(c->bstage + start[e], in[idx[e]], len[e]); });
  c->tbegin(); c->tmark("bz:begin");
  uint8_t *d_arena = c->ws.rin_own;
  uint32_t *d_tab = (uint32_t *)(d_arena + ((total + 63) & ~63ull));          // starts, lengths, CRC registers behind the entries
  hipMemcpyAsync(d_arena, c->bstage, total, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_tab, start32.data(), 4ull * E, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_tab + E, len.data(), 4ull * E, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_tab + 2 * E, crc_in.data(), 4ull * E, hipMemcpyHostToDevice, st);
  hipLaunchKernelGGL(k_batch_crc, dim3(E), dim3(64), 0, st, E, d_arena, d_tab, d_tab + E, d_tab + 2 * E);
  hipMemcpyAsync(crc_in.data(), d_tab + 2 * E, 4ull * E, hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "bzip2 batch in")) return ZADA_E_HIP;
  std::vector<uint64_t> off(E), bytes(E);
  rc = bz2_batch_encode(c, method - ZADA_BZIP2_1, d_arena, E, start.data(), len.data(), c->bstage, out_cap, off.data(), bytes.data());
  c->tmark("bz:end"); c->tend();
  if (rc) return rc;
  std::atomic<int> bad(0);
  parallel_entries(E, out_cap, [&](uint32_t e) {
    const int i = idx[e];
    out_len[i] = bytes[e];
    if (crc) crc[i] = crc_in[e];
    rc_out[i] = bytes[e] >= n[i] ? ZADA_INEFFICIENT : ZADA_OK;                    // zip-compress.adb:479-486
    if (bytes[e] <= cap[i]) memcpy(out[i], c->bstage + off[e], bytes[e]);
    else if (rc_out[i] == ZADA_OK) { rc_out[i] = ZADA_E_INVALID; bad = 1; }
  });
  if (bad) c->err = "output buffer too small";
  return 0;
}

int zada_bzip2_batch(zada_ctx *z, int method, int count, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out, const uint64_t *cap, uint64_t *out_len,
                     uint32_t *crc, int *rc) {
  if (!z || count < 0 || method < ZADA_BZIP2_1 || method > ZADA_BZIP2_3) return ZADA_E_INVALID;
  int prc = prepare(z);
  if (prc) return prc;
  Ctx *c = &z->c;
  // an entry whose RLE_1 form cannot reach the block capacity is one block (bzip2-encoding.adb:1161-1209: bytes are taken while
  // the simulated size + 5 stays below the capacity; RLE_1 grows data by a quarter at most); it also stays below the sizes
  // at which the last two blocks are balanced (:1406-1423)
  const uint64_t capacity = method == ZADA_BZIP2_1 ? 100000 : method == ZADA_BZIP2_2 ? 400000 : 900000, one_block = (capacity - 16) * 4 / 5;
  int worst = 0;
  std::vector<int> group;
  uint64_t gbytes = 0;
  auto flush_group = [&]() {
    if (group.empty()) return;
    int r = group.size() == 1 ? 1 : finish_call(c, bzip2_batch_core(c, method, group.data(), (uint32_t)group.size(), in, n, out, cap, out_len, crc, rc));
    if (group.size() == 1 || r == ZADA_E_NOMEM || r == ZADA_E_TOO_LARGE) {
      for (int i : group) { rc[i] = zada_bzip2(z, method, in[i], n[i], out[i], cap[i], &out_len[i], crc ? &crc[i] : nullptr, nullptr, nullptr); if (rc[i] < 0) worst = rc[i]; }
    } else if (r < 0) { for (int i : group) rc[i] = r; worst = r; }
    else { for (int i : group) if (rc[i] < 0) worst = rc[i]; }
    group.clear(); gbytes = 0;
  };
  for (int i = 0; i < count; i++) {
    if (n[i] <= one_block) {
      const uint64_t slot = (n[i] + 63) & ~63ull;
      if (gbytes + slot > ((uint64_t)c->knob_bz_batch_mib << 20)) flush_group();
      group.push_back(i); gbytes += slot;
    } else {
      rc[i] = zada_bzip2(z, method, in[i], n[i], out[i], cap[i], &out_len[i], crc ? &crc[i] : nullptr, nullptr, nullptr);
      if (rc[i] < 0) worst = rc[i];
    }
  }
  flush_group();
  return worst;
}
// --------------------------------------------------------------------------------------------
// LZMA: Zip.Compress.LZMA_E (zip-compress-lzma_e.adb:121-172) for LZMA_0 .. LZMA_3.  One stream per
// workgroup of k_lzma_encode (zada_lzma.hip); Level_1 / Level_2 take their tokens from the LZ stage of the Deflate path.
// --------------------------------------------------------------------------------------------
static int lz_grow(Ctx *c, void **p, size_t *cap, size_t bytes) {
  if (*p && *cap >= bytes) return 0;
  hipStreamSynchronize(c->stream);
  if (*p) { hipFree(*p); *p = nullptr; *cap = 0; }
  const size_t want = bytes + bytes / 8 + 4096;
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) { (void)hipGetLastError(); hip_check(c, e, "hipMalloc (LZMA workspace)"); *p = nullptr; return ZADA_E_NOMEM; }
  *cap = want;
  return 0;
}
static void lzma_free(Ctx *c) {
  if (c->lz_tab) hipFree(c->lz_tab);
  if (c->lz_save) hipFree(c->lz_save);
  c->lz_tab = c->lz_save = nullptr; c->cap_lz_tab = c->cap_lz_save = 0;
  bt4_destroy(c);
}
// One LZMA_3 stream in launches (budget > 0): log2 of the positions per segment of the BT4 producer, 32 = no segments.  The match sets of
// segment k + 1 are found (stream2) while the coder -- one wave -- codes segment k: of all the producer's time, only segment 0's is waited for.
static uint32_t lzma_segment_shift(const Ctx *c, uint64_t n) {
  if (c->knob_lzma_segment < 0) return 32;
  // (by size: 2 ** 20 positions from 2 MiB on, 2 ** 18 from 512 KiB on -- the walks of a small segment are launch overhead and short buckets' tails)
  const uint32_t sh = c->knob_lzma_segment > 0 ? (uint32_t)c->knob_lzma_segment : n >= (2ull << 20) ? 20 : 18;
  return n >= (2ull << sh) ? sh : 32;                                // (fewer than two segments: nothing to overlap)
}
// jobs: sbs / hash4_size are filled here.  res: 2 per job (stream bytes, input bytes coded).  arena_bytes: the bytes at d_in that hold
// the entries (Level_3: the BT4 producer of zada_bt4.hip writes the match sets of all entries before the coder starts).
// budget > 0: launches of `budget` positions per stream (the coder's state waits in HBM in between, zada_lzma.hip "A stream in several
// launches"), with feedback (pct_lo .. pct_hi by positions coded) and the abort test between them.
static int lzma_run(Ctx *c, std::vector<LzmaJob> &jobs, const uint8_t *d_in, uint64_t arena_bytes, const uint32_t *d_tok, uint8_t *d_out, std::vector<uint64_t> &res,
                    const uint32_t *d_apos = nullptr, uint32_t T = 0, const uint32_t *d_ent_start = nullptr,
                    uint64_t budget = 0, zada_feedback_fn fb = nullptr, void *user = nullptr, int pct_lo = 0, int pct_hi = 100) {
  const uint32_t E = (uint32_t)jobs.size();
  bool bt4 = false;
  for (LzmaJob &j : jobs) {
    j.sbs = lzma_string_buffer_size(j.level, c->knob_lzma_dict > 0 ? (uint64_t)c->knob_lzma_dict : j.n);   // dictionary_size = the entry's size, zip-compress-lzma_e.adb:165
    j.hash4_size = j.level == 3 ? lzma_hash4_size(j.sbs) : 0;
    j.verify = 0;
    if (j.level == 3 && j.n > j.sbs) {                                 // (n <= String_buffer_size: the whole entry in the first fill, nothing is read behind a gap)
      std::vector<Bt4Run> runs;
      if (bt4_schedule(j.n, j.sbs, runs) && bt4_reads_behind_a_gap(runs)) j.verify = 1;
    }
    bt4 = bt4 || (j.level == 3 && j.n > 0);
  }
  // (a state from zada_lzma_import_state is for the next stream coded, whatever comes of it: it has to be the state of THIS stream -- one LZMA_3 stream in
  // launches, same length, dictionary and place -- or the call is refused; the kernel would take its counters as they are)
  std::vector<uint8_t> resume_blob;
  resume_blob.swap(c->lz_resume);
  if (!resume_blob.empty() && !(E == 1 && budget > 0 && resume_blob.size() == lzma_save_stride() && lzma_save_fits(resume_blob.data(), jobs[0]))) {
    c->err = "LZMA: the imported state is not one of this stream (ONE LZMA_3 stream in launches, of the same length and dictionary)";
    return ZADA_E_INVALID;
  }
  int rc = lz_grow(c, &c->lz_tab, &c->cap_lz_tab, (sizeof(LzmaJob) + 16 + 4) * (size_t)E + 192);
  if (rc) return rc;
  Bt4Sets sets{nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<uint32_t> weight;                                    // Level_3: what the producer found in each entry
  uint32_t seg_shift = bt4 && E == 1 && budget > 0 && jobs[0].in_off == 0 ? lzma_segment_shift(c, jobs[0].n) : 32, nseg = 0;
  if (bt4) {
    if ((rc = bt4_produce(c, jobs, d_in, arena_bytes, &sets, &weight, seg_shift, &nseg))) return rc;
    c->tmark("lzma:bt4");
  }
  LzmaJob *d_jobs = (LzmaJob *)c->lz_tab;
  uint64_t *d_res = (uint64_t *)((uint8_t *)c->lz_tab + ((sizeof(LzmaJob) * (size_t)E + 63) & ~63ull));
  uint32_t *d_order = (uint32_t *)((uint8_t *)d_res + ((16 * (size_t)E + 63) & ~63ull));
  std::vector<uint32_t> order(E);                                  // a stream's time goes with its length -- and, at Level_3, with the matches in it: the heaviest first
  for (uint32_t e = 0; e < E; e++) order[e] = e;
  if (weight.size() == E) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
  else std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return jobs[a].n > jobs[b].n; });
  hipMemcpyAsync(d_jobs, jobs.data(), sizeof(LzmaJob) * (size_t)E, hipMemcpyHostToDevice, c->stream);
  hipMemcpyAsync(d_order, order.data(), 4 * (size_t)E, hipMemcpyHostToDevice, c->stream);
  if (d_apos && (rc = lzma_token_ranges(c, E, d_apos, T, d_ent_start, d_jobs))) return rc;   // token ranges of a batch, found on the device
  res.resize(2 * (size_t)E);
  if (budget == 0) {
    if ((rc = lzma_launch(c, d_jobs, d_order, E, d_in, d_tok, d_out, sets, d_res))) return rc;
    hipMemcpyAsync(res.data(), d_res, 16 * (size_t)E, hipMemcpyDeviceToHost, c->stream);
    if (hip_check(c, hipStreamSynchronize(c->stream), "k_lzma_encode")) return ZADA_E_HIP;
  } else {
    const size_t save_bytes = (size_t)lzma_save_stride() * E;
    if ((rc = lz_grow(c, &c->lz_save, &c->cap_lz_save, save_bytes))) return rc;
    // (a state from zada_lzma_import_state: ONE stream goes on from where an earlier call -- of another context, of another process -- stopped; the match
    // sets are a function of the input alone, so the producer simply makes them again, and the launches up to the state's position code nothing)
    const bool resume = !resume_blob.empty();
    auto fresh_state = [&]() {
      if (resume) hipMemcpyAsync(c->lz_save, resume_blob.data(), save_bytes, hipMemcpyHostToDevice, c->stream);
      else hipMemsetAsync(c->lz_save, 0, save_bytes, c->stream);
    };
    fresh_state();
    uint64_t total = 0;
    for (const LzmaJob &j : jobs) total += j.n;
    c->lzma_launches = 0;
    // (segments: the coder stops LZ_SEG_MARGIN positions short of a segment whose sets are still being written -- a step of its loop looks at
    // the sets of fewer than 2 x 273 positions beyond the one it codes)
    constexpr uint64_t LZ_SEG_MARGIN = 4096;
    // one LZMA_3 stream alone: its workgroup gets four waves, the chain's and three helpers for its forks (zada_lzma.hip "one stream on four waves")
    const int waves = E == 1 && bt4 && (c->knob_lzma_waves == 0 || c->knob_lzma_waves == 4) ? 4 : 1;
    uint32_t k = 0;                                                  // segments whose sets are there
    uint64_t cap = ~0ull;
    if (seg_shift < 32 && (rc = bt4_walk_segment(c, 0, c->stream2))) return rc;
    for (;;) {
      if (seg_shift < 32 && k < nseg && (k == 0 || cap != ~0ull)) {
        // segment k's sets are complete when stream2 is idle; the next segment's walks start before the coder goes on
        const int ov = bt4_segments_overflowed(c, c->stream2, &sets);     // (may move the overflow pool to a larger one: nothing of this stream is running)
        if (ov < 0) return ov;
        if (ov) {                                                    // the pool was too small: the whole stream again, the producer first (it grows the pool itself)
          hipStreamSynchronize(c->stream);
          c->bt4_reruns++;
          seg_shift = 32; cap = ~0ull;
          if ((rc = bt4_produce(c, jobs, d_in, arena_bytes, &sets, nullptr))) return rc;
          fresh_state();
        } else {
          k++;
          if (k < nseg && (rc = bt4_walk_segment(c, k, c->stream2))) return rc;
          cap = k < nseg ? ((uint64_t)k << seg_shift) - LZ_SEG_MARGIN : ~0ull;
        }
      }
      uint64_t pos = 0;
      for (;;) {                                                     // the launches up to `cap`
        if ((rc = lzma_launch(c, d_jobs, d_order, E, d_in, d_tok, d_out, sets, d_res, (uint8_t *)c->lz_save, budget, cap, waves))) return rc;
        c->lzma_launches++;
        hipMemcpyAsync(res.data(), d_res, 16 * (size_t)E, hipMemcpyDeviceToHost, c->stream);
        if (hip_check(c, hipStreamSynchronize(c->stream), "k_lzma_encode")) return ZADA_E_HIP;
        bool more = false;
        uint64_t coded = 0;
        for (uint32_t e = 0; e < E; e++) { more = more || (res[2 * e + 1] >> 63); res[2 * e + 1] &= ~(1ull << 63); coded += res[2 * e + 1] & ~(1ull << 62); }
        pos = more ? coded : ~0ull;
        if (!more) break;
        if (fb && fb(pct_lo + (int)((uint64_t)(pct_hi - pct_lo) * coded / (total ? total : 1)), user)) {
          if (seg_shift < 32) hipStreamSynchronize(c->stream2);      // (walks under way write into the context's buffers)
          return ZADA_ABORTED;
        }
        if (coded >= cap) break;
      }
      if (pos == ~0ull) break;
    }
    // (a stream that ended before its last segment -- refused, ZADA_E_REFERENCE -- leaves the walks of the next segment under way: they write into
    // the context's buffers, which the next call fills again)
    if (nseg && hip_check(c, hipStreamSynchronize(c->stream2), "k_bt4_walk (segment)")) return ZADA_E_HIP;
  }
  // bit 62 of an entry's second result: ZADA_E_REFERENCE (the callers look at it: lzma_refused)
  for (uint32_t e = 0; e < E; e++) if (!(res[2 * e + 1] >> 62 & 1) && res[2 * e + 1] != jobs[e].n) { c->err = "LZMA: the coder did not consume the entry"; return ZADA_E_HIP; }
  return 0;
}
static bool lzma_refused(Ctx *c, const std::vector<uint64_t> &res, uint32_t e) {
  if (!(res[2 * (size_t)e + 1] >> 62 & 1)) return false;
  c->err = "LZMA_3: the reference's matcher reports a match that is none on this entry (positions read behind pending bytes no window fill took up, "
           "lz77.adb:1000-1017, 1262-1290): its own stream does not decode to the input";
  return true;
}
static int lzma_tokens(Ctx *c, int level, const uint8_t *d_in, uint64_t n, uint64_t *ntok) {     // IZ_6 / IZ_10, lzma-encoding.adb:118-122
  int rc = range_open(c, level == 1 ? ZADA_DEFLATE_1 : ZADA_DEFLATE_3, d_in, n, 0, n, 0, 0);
  if (!rc) rc = range_lz(c, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (hip_check(c, hipStreamSynchronize(c->stream2), "LZMA tokens")) return ZADA_E_HIP;             // (range_lz runs the range's CRC there)
  *ntok = c->rg.T;
  c->rg.open = false;
  return 0;
}
// Positions a launch codes before the coder's state goes back to HBM and the caller is asked whether to go on ("lzma_chunk" knob;
// 0 = by level, sized for launches of about half a second: one stream is ONE wave walking a chain).
static uint64_t lzma_budget(const Ctx *c, int level) {
  if (c->knob_lzma_chunk > 0) return (uint64_t)c->knob_lzma_chunk;
  if (c->knob_lzma_chunk < 0) return 0;                               // (one launch per stream, as a batch does it)
  return level == 0 ? 4ull << 20 : level == 1 ? 1ull << 20 : level == 2 ? 256ull << 10 : 64ull << 10;
}
static int lzma_core(Ctx *c, int method, const uint8_t *d_in, uint64_t n, uint8_t *d_out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
                     zada_feedback_fn fb, void *user) {
  if (method < ZADA_LZMA_0 || method > ZADA_LZMA_3) { c->err = "not an LZMA method"; return ZADA_E_INVALID; }
  // (BT4's positions are Integers that the reference re-bases at Integer'Last, lz77.adb:981-990, 1078-1083; the kernel does not)
  if (n >= (2ull << 30) - 65536) { c->err = "LZMA: entries of 2 GiB and more are not taken"; return ZADA_E_TOO_LARGE; }
  const int level = method - ZADA_LZMA_0;
  int rc;
  if (fb && fb(0, user)) return ZADA_ABORTED;                         // zip-compress-lzma_e.adb:78-92
  c->tbegin(); c->tmark("lzma:begin");
  if (crc_inout && n) {
    if ((rc = ensure_crc_workspace(c, n))) return rc;
    if ((rc = crc_launch(c, d_in, n)) || (rc = crc_finish(c, n, crc_inout))) return rc;
  }
  std::vector<LzmaJob> jobs(1);
  LzmaJob &J = jobs[0];
  memset(&J, 0, sizeof J);
  J.n = n; J.cap = cap; J.level = level; J.zip_prefix = 1;
  const uint32_t *d_tok = nullptr;
  if ((level == 1 || level == 2) && n) {
    if ((rc = lzma_tokens(c, level, d_in, n, &J.ntok))) return rc;
    d_tok = c->ws.ea_atoms + LB_CAP;
  }
  c->tmark("lzma:tokens");
  const int pct0 = d_tok ? 10 : 1;
  if (fb && fb(pct0, user)) return ZADA_ABORTED;
  std::vector<uint64_t> res;
  if ((rc = lzma_run(c, jobs, d_in, n, d_tok, d_out, res, nullptr, 0, nullptr, lzma_budget(c, level), fb, user, pct0, 99))) return rc;
  c->tmark("lzma:end"); c->tend();
  if (lzma_refused(c, res, 0)) { *out_len = 0; return ZADA_E_REFERENCE; }
  if (fb && fb(100, user)) return ZADA_ABORTED;
  *out_len = res[0];
  if (res[0] > cap) { if (res[0] >= n) return ZADA_INEFFICIENT; c->err = "output buffer too small"; return ZADA_E_INVALID; }
  return res[0] >= n ? ZADA_INEFFICIENT : ZADA_OK;                   // zip-compress.adb:479-486
}
int zada_lzma_device(zada_ctx *z, int method, const void *d_in, uint64_t n, void *d_out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  const uint8_t *src = (const uint8_t *)d_in;
  if (((uintptr_t)d_in & 15) != 0 && n) {
    if ((rc = ensure_rin(c, n))) return rc;
    hipMemcpyAsync(c->ws.rin_own, d_in, n, hipMemcpyDeviceToDevice, c->stream);
    src = c->ws.rin_own;
  }
  uint64_t ol = 0;
  rc = finish_call(c, lzma_core(c, method, src, n, (uint8_t *)d_out, cap, &ol, crc_inout, nullptr, nullptr));
  if (out_len && rc >= 0) *out_len = ol;
  return rc;
}
int zada_lzma(zada_ctx *z, int method, const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
              zada_feedback_fn fb, void *user) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  if ((rc = ensure_rin(c, n + cap + 128))) return rc;           // input, then the stream, in one device buffer
  copy_in(c, c->ws.rin_own, in, n);
  uint8_t *d_out = c->ws.rin_own + ((n + 63) & ~63ull);
  uint64_t ol = 0;
  c->lz_last_n = n; c->lz_last_out_off = (n + 63) & ~63ull;
  rc = finish_call(c, lzma_core(c, method, c->ws.rin_own, n, d_out, cap, &ol, crc_inout, fb, user));
  c->lz_resume.clear();                                           // (an imported state is for one call)
  if (rc < 0 || rc == ZADA_ABORTED) return rc;
  if (out_len) *out_len = ol;
  if (ol <= cap && copy_out(c, out, d_out, ol)) return ZADA_E_HIP;
  return rc;
}
// A stream that stopped between two launches (its feedback said so: ZADA_ABORTED) can be taken up again -- by this context, another one, or another
// process: export the coder's state and the stream bytes written so far, import the state into a context and call zada_lzma with the SAME input and
// method again.  The second call's output buffer holds the whole stream's length with the bytes from *out_bytes of the export on; the bytes before are
// the export's.  (The match producer's sets are a function of the input alone and are made again.)  This is how a stream that takes longer than a
// machine lets one call run is coded in two (tests/gpu_lzma_c4.py: BASELINE config 4 on a pool that ends a call after an hour).
int zada_lzma_export_state(zada_ctx *z, uint8_t *state, uint64_t state_cap, uint64_t *state_len, uint8_t *out, uint64_t out_cap, uint64_t *out_bytes, uint64_t *positions) {
  if (!z) return ZADA_E_INVALID;
  Ctx *c = &z->c;
  const uint64_t sb = lzma_save_stride();
  if (state_len) *state_len = sb;
  if (!state) return ZADA_OK;                                      // (the length only)
  if (!c->lz_save || c->cap_lz_save < sb || state_cap < sb) { c->err = "zada_lzma_export_state: no stopped stream, or the state buffer is too small"; return ZADA_E_INVALID; }
  if (hipSetDevice(c->device) != hipSuccess) return ZADA_E_HIP;
  if (hipMemcpy(state, c->lz_save, sb, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return ZADA_E_HIP; }
  uint64_t pos = 0, olen = 0, n = 0;
  if (!lzma_save_info(state, &pos, &olen, &n) || n != c->lz_last_n) { c->err = "zada_lzma_export_state: the last zada_lzma call did not stop between two launches of ONE stream"; return ZADA_E_INVALID; }
  if (out_bytes) *out_bytes = olen;
  if (positions) *positions = pos;
  if (out) {
    if (olen > out_cap) { c->err = "zada_lzma_export_state: output buffer too small"; return ZADA_E_INVALID; }
    if (olen && copy_out(c, out, c->ws.rin_own + c->lz_last_out_off, olen)) return ZADA_E_HIP;
  }
  return ZADA_OK;
}
int zada_lzma_import_state(zada_ctx *z, const uint8_t *state, uint64_t state_len) {
  if (!z) return ZADA_E_INVALID;
  Ctx *c = &z->c;
  uint64_t pos = 0, olen = 0, n = 0;
  if (!state || state_len != lzma_save_stride() || !lzma_save_info(state, &pos, &olen, &n)) { c->err = "zada_lzma_import_state: not the state of a stopped stream"; return ZADA_E_INVALID; }
  c->lz_resume.assign(state, state + state_len);
  return ZADA_OK;
}
// Test hook: the match sets the BT4 producer leaves for ONE entry (the stage the coder reads; compare zo_bt4_match_sets of the
// oracle): cnt [n], len / dist [n * stride] (stride >= 50).  The dictionary is the entry's size or the "lzma_dict" knob.
int zada_lzma_match_sets(zada_ctx *z, const uint8_t *in, uint64_t n, uint8_t *cnt, uint16_t *len, uint32_t *dist, int stride) {
  if (!z || stride < BT4_SET || n >= (2ull << 30) - 65536) return ZADA_E_INVALID;
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  if (n == 0) return 0;
  if ((rc = ensure_rin(c, n + 64))) return rc;
  copy_in(c, c->ws.rin_own, in, n);
  std::vector<LzmaJob> jobs(1);
  LzmaJob &J = jobs[0];
  memset(&J, 0, sizeof J);
  J.n = n; J.level = 3;
  J.sbs = lzma_string_buffer_size(3, c->knob_lzma_dict > 0 ? (uint64_t)c->knob_lzma_dict : n);
  J.hash4_size = lzma_hash4_size(J.sbs);
  Bt4Sets S;
  const uint32_t seg_shift = lzma_segment_shift(c, n);                // (segments as a stream in launches takes them, one after the other)
  uint32_t nseg = 0;
  rc = bt4_produce(c, jobs, c->ws.rin_own, n, &S, nullptr, seg_shift, &nseg);
  for (uint32_t k = 0; !rc && k < nseg; k++) rc = bt4_walk_segment(c, k, c->stream);
  if (!rc && nseg) { const int ov = bt4_segments_overflowed(c, c->stream); if (ov) { if (ov > 0) c->err = "match sets: the overflow pool is too small for segments"; rc = ov < 0 ? ov : ZADA_E_NOMEM; } }
  if ((rc = finish_call(c, rc))) return rc;
  std::vector<uint16_t> sl(n * BT4_INLINE); std::vector<uint32_t> sd(n * BT4_INLINE);
  const uint64_t nov = c->bt4_overflow;
  std::vector<uint16_t> ol(nov * BT4_OVF + 1); std::vector<uint32_t> od(nov * BT4_OVF + 1);
  hipMemcpy(cnt, S.cnt, n, hipMemcpyDeviceToHost);
  hipMemcpy(sl.data(), S.sl, 2 * n * BT4_INLINE, hipMemcpyDeviceToHost);
  hipMemcpy(sd.data(), S.sd, 4 * n * BT4_INLINE, hipMemcpyDeviceToHost);
  if (nov) { hipMemcpy(ol.data(), S.ol, 2 * nov * BT4_OVF, hipMemcpyDeviceToHost); hipMemcpy(od.data(), S.od, 4 * nov * BT4_OVF, hipMemcpyDeviceToHost); }
  if (hip_check(c, hipGetLastError(), "match sets")) return ZADA_E_HIP;
  for (uint64_t p = 0; p < n; p++)
    for (int i = 0; i < cnt[p]; i++) {
      if (i < BT4_INLINE - 1) { len[p * stride + i] = sl[p * BT4_INLINE + i]; dist[p * stride + i] = sd[p * BT4_INLINE + i]; }
      else {
        const uint64_t o = (uint64_t)sd[p * BT4_INLINE + BT4_INLINE - 1] * BT4_OVF + (uint32_t)(i - (BT4_INLINE - 1));
        if (o >= ol.size()) { c->err = "match sets: overflow block out of range"; return ZADA_E_HIP; }
        len[p * stride + i] = ol[o]; dist[p * stride + i] = od[o];
      }
    }
  return 0;
}
// Many entries in one call: every entry is a stream of ONE launch of k_lzma_encode -- the only parallelism LZMA's chain of
// adaptive probabilities leaves (see zada_lzma.hip).  This one: Level_0 and Level_3 (no tokens from the LZ stage).
static int lzma_batch_core(Ctx *c, int method, const int *idx, uint32_t E, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out, const uint64_t *cap,
                           uint64_t *out_len, uint32_t *crc, int *rc_out) {
  hipStream_t st = c->stream;
  const int level = method - ZADA_LZMA_0;
  std::vector<uint64_t> start(E), ostart(E); std::vector<uint32_t> start32(E + 1), len(E + 1), crc_in(E + 1);
  uint64_t total = 0, ototal = 0;
  for (uint32_t e = 0; e < E; e++) {
    start[e] = total; start32[e] = (uint32_t)total; len[e] = (uint32_t)n[idx[e]]; crc_in[e] = crc ? crc[idx[e]] : 0xFFFFFFFFu;
    total += ((n[idx[e]] ? n[idx[e]] : 1) + 63) & ~63ull;                        // (64-byte slots: Level_0 / Level_3 never see the LZ stage)
    ostart[e] = ototal; ototal += (n[idx[e]] + n[idx[e]] / 8 + 128 + 63) & ~63ull;
  }
  if (total >= (1ull << 32)) return ZADA_E_TOO_LARGE;
  int rc = ensure_rin(c, total + ototal + 64 + 16ull * (E + 1));
  if (!rc) rc = grow_pinned((void **)&c->bstage, &c->cap_bstage, (total > ototal ? total : ototal) + 64);
  if (rc) return rc;
  parallel_entries(E, total, [&](uint32_t e) { if (len[e]) memcpy(c->bstage + start[e], in[idx[e]], len[e]); });
  c->tbegin(); c->tmark("lzma:begin");
  uint8_t *d_arena = c->ws.rin_own, *d_out = d_arena + total;
  uint32_t *d_tab = (uint32_t *)(d_out + ototal);
  hipMemcpyAsync(d_arena, c->bstage, total, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_tab, start32.data(), 4ull * E, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_tab + E, len.data(), 4ull * E, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_tab + 2 * E, crc_in.data(), 4ull * E, hipMemcpyHostToDevice, st);
  hipLaunchKernelGGL(k_batch_crc, dim3(E), dim3(64), 0, st, E, d_arena, d_tab, d_tab + E, d_tab + 2 * E);
  hipMemcpyAsync(crc_in.data(), d_tab + 2 * E, 4ull * E, hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "LZMA batch in")) return ZADA_E_HIP;
  std::vector<LzmaJob> jobs(E);
  for (uint32_t e = 0; e < E; e++) {
    LzmaJob &J = jobs[e];
    memset(&J, 0, sizeof J);
    J.in_off = start[e]; J.n = len[e]; J.out_off = ostart[e]; J.cap = (len[e] + len[e] / 8 + 128ull); J.level = level; J.zip_prefix = 1;
  }
  c->tmark("lzma:tokens");
  std::vector<uint64_t> res;
  if ((rc = lzma_run(c, jobs, d_arena, total, nullptr, d_out, res))) return rc;
  hipMemcpyAsync(c->bstage, d_out, ototal, hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "LZMA batch out")) return ZADA_E_HIP;
  c->tmark("lzma:end"); c->tend();
  std::atomic<int> bad(0);
  std::atomic<int> refused(0);
  parallel_entries(E, ototal, [&](uint32_t e) {
    const int i = idx[e];
    const uint64_t bytes = res[2 * e];
    out_len[i] = bytes;
    if (crc) crc[i] = crc_in[e];
    if (res[2 * (size_t)e + 1] >> 62 & 1) { rc_out[i] = ZADA_E_REFERENCE; out_len[i] = 0; refused = 1; return; }
    rc_out[i] = bytes >= n[i] ? ZADA_INEFFICIENT : ZADA_OK;
    if (bytes <= cap[i] && bytes <= jobs[e].cap) memcpy(out[i], c->bstage + ostart[e], bytes);
    else if (rc_out[i] == ZADA_OK) { rc_out[i] = ZADA_E_INVALID; bad = 1; }
  });
  if (bad) c->err = "output buffer too small";
  if (refused) for (uint32_t e = 0; e < E; e++) if (lzma_refused(c, res, e)) break;     // (the error text)
  return 0;
}
// Level_1 / Level_2 batches: the tokens of ALL entries from one pass of the LZ stage (the batch layout of batch_core above:
// 32 KiB slots, the segment table that ends every entry's searches at its own end), then one launch of the coder.
static int lzma_batch_iz(Ctx *c, int method, const int *idx, uint32_t E, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out, const uint64_t *cap,
                         uint64_t *out_len, uint32_t *crc, int *rc_out) {
  const int level = method - ZADA_LZMA_0, iz_level = level == 1 ? 6 : 10;
  Workspace &W = c->ws;
  hipStream_t st = c->stream;
  std::vector<uint32_t> start(E + 1), len(E + 1), crc_in(E + 1);
  std::vector<uint64_t> ostart(E);
  uint64_t total = 0, nfs = 0, ototal = 0;
  for (uint32_t e = 0; e < E; e++) {
    const uint64_t ln = n[idx[e]], slot = ((ln ? ln : 1) + 32767) & ~32767ull;
    start[e] = (uint32_t)total; len[e] = (uint32_t)ln; crc_in[e] = crc ? crc[idx[e]] : 0xFFFFFFFFu;
    total += slot; nfs += ln ? (ln + FLUSH - 1) / FLUSH : 1;
    ostart[e] = ototal; ototal += (ln + ln / 8 + 128 + 63) & ~63ull;
  }
  start[E] = (uint32_t)total; len[E] = 0;
  if (total >= (1ull << 31)) return ZADA_E_TOO_LARGE;
  const uint32_t nseg = (uint32_t)(total >> 15);
  int rc = ensure_lz_workspace(c, total + 4096);
  if (!rc) rc = ensure_entropy_workspace(c, total, nfs);
  if (!rc) rc = ensure_batch_workspace(c, E, nfs, nseg);
  if (!rc) rc = ensure_rin(c, ototal + 64);
  if (!rc) rc = grow_pinned((void **)&c->bstage, &c->cap_bstage, (total > ototal ? total : ototal) + 64);
  const uint64_t tabw = (uint64_t)nseg + 3ull * (E + 1) + 64;
  uint64_t capw = c->cap_btab * 4;
  if (!rc) { rc = grow_pinned((void **)&c->btab, &capw, tabw * 4); c->cap_btab = capw / 4; }
  if (rc) return rc;
  uint32_t *t_seg = c->btab, *t_ent = c->btab + nseg;
  parallel_entries(E, total, [&](uint32_t e) {
    if (len[e]) memcpy(c->bstage + start[e], in[idx[e]], len[e]);
    for (uint32_t s = start[e] >> 15; s < (start[e + 1] >> 15); s++) t_seg[s] = (start[e] + len[e]) | (s == (start[e] >> 15) ? 0x80000000u : 0u);
  });
  memcpy(t_ent, start.data(), (E + 1) * 4); memcpy(t_ent + (E + 1), len.data(), (E + 1) * 4); memcpy(t_ent + 2 * (E + 1), crc_in.data(), (E + 1) * 4);
  c->tbegin(); c->tmark("lzma:begin");
  hipMemcpyAsync(W.in, c->bstage, total, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.segend, t_seg, (size_t)nseg * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_start, t_ent, (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_len, t_ent + (E + 1), (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(W.ent_crc, t_ent + 2 * (E + 1), (size_t)(E + 1) * 4, hipMemcpyHostToDevice, st);
  {
    PadArgs pa; pa.in_end = W.in + total; pa.n_in = IN_PAD;
    for (int l = 0; l < NLEVELS; l++) pa.link_end[l] = W.lprev[l] + (total >= 2 ? total - 2 : 0);
    hipLaunchKernelGGL(k_pad_init, dim3(1), dim3(256), 0, st, pa);
  }
  c->rg = Range();
  c->last_nblocks = 0; c->demand_rounds = 0; c->parse_rounds = 0;
  hipLaunchKernelGGL(k_batch_crc, dim3(E), dim3(64), 0, st, E, W.in, W.ent_start, W.ent_len, W.ent_crc);
  ShardJob job;
  job.nbuf = total; job.tok_lo = 0; job.tok_hi = (uint32_t)total; job.final = true; job.entry_known = true; job.entry = ExitState{0, SYNC_F};
  job.dst_atoms = W.ea_atoms + LB_CAP; job.dst_apos = W.ea_apos + LB_CAP; job.apos_bias = 0; job.cap_atoms = W.cap_atoms; job.segend = W.segend;
  ShardResult sres;
  rc = lz_shard(c, iz_level, job, &sres);
  if (rc) return rc == -2 ? ZADA_E_NOMEM : rc;
  c->tmark("lzma:tokens");
  std::vector<LzmaJob> jobs(E);
  for (uint32_t e = 0; e < E; e++) {
    LzmaJob &J = jobs[e];
    memset(&J, 0, sizeof J);
    J.in_off = start[e]; J.n = len[e]; J.out_off = ostart[e]; J.cap = len[e] + len[e] / 8 + 128ull; J.level = level; J.zip_prefix = 1;
  }
  std::vector<uint64_t> res;
  uint8_t *d_out = c->ws.rin_own;
  if ((rc = lzma_run(c, jobs, W.in, total, W.ea_atoms + LB_CAP, d_out, res, W.ea_apos + LB_CAP, sres.ntok, W.ent_start))) return rc;
  hipMemcpyAsync(crc_in.data(), W.ent_crc, 4ull * E, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(c->bstage, d_out, ototal, hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "LZMA batch out")) return ZADA_E_HIP;
  c->tmark("lzma:end"); c->tend();
  std::atomic<int> bad(0);
  parallel_entries(E, ototal, [&](uint32_t e) {
    const int i = idx[e];
    const uint64_t bytes = res[2 * e];
    out_len[i] = bytes;
    if (crc) crc[i] = crc_in[e];
    rc_out[i] = bytes >= n[i] ? ZADA_INEFFICIENT : ZADA_OK;
    if (bytes <= cap[i] && bytes <= jobs[e].cap) memcpy(out[i], c->bstage + ostart[e], bytes);
    else if (rc_out[i] == ZADA_OK) { rc_out[i] = ZADA_E_INVALID; bad = 1; }
  });
  if (bad) c->err = "output buffer too small";
  return 0;
}
int zada_lzma_batch(zada_ctx *z, int method, int count, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out, const uint64_t *cap, uint64_t *out_len,
                    uint32_t *crc, int *rc) {
  if (!z || count < 0 || method < ZADA_LZMA_0 || method > ZADA_LZMA_3) return ZADA_E_INVALID;
  int prc = prepare(z);
  if (prc) return prc;
  Ctx *c = &z->c;
  int worst = 0;
  std::vector<int> group;
  uint64_t gbytes = 0;
  auto flush_group = [&]() {
    if (group.empty()) return;
    const bool iz = method == ZADA_LZMA_1 || method == ZADA_LZMA_2;
    int r = finish_call(c, iz ? lzma_batch_iz(c, method, group.data(), (uint32_t)group.size(), in, n, out, cap, out_len, crc, rc)
                              : lzma_batch_core(c, method, group.data(), (uint32_t)group.size(), in, n, out, cap, out_len, crc, rc));
    if (r < 0) { for (int i : group) rc[i] = r; worst = r; }
    else { for (int i : group) if (rc[i] < 0) worst = rc[i]; }
    group.clear(); gbytes = 0;
  };
  for (int i = 0; i < count; i++) {
    if (n[i] >= (2ull << 30) - 65536) { rc[i] = ZADA_E_TOO_LARGE; worst = rc[i]; continue; }
    const bool iz_slots = method == ZADA_LZMA_1 || method == ZADA_LZMA_2;       // (whole 32 KiB segments of the LZ stage; 64-byte slots otherwise)
    const uint64_t slot = iz_slots ? ((n[i] ? n[i] : 1) + 32767) & ~32767ull : ((n[i] ? n[i] : 1) + 63) & ~63ull;
    if (gbytes + slot > (1ull << 30)) flush_group();
    group.push_back(i); gbytes += slot;
  }
  flush_group();
  return worst;
}
// raw CRC-32 register (started from 0) of n bytes in device memory: the piece a rank contributes to a stream's CRC
// (zada_crc32_combine chains the pieces)
int zada_crc32_device(zada_ctx *z, const void *d_in, uint64_t n, uint32_t *raw) {
  int rc = prepare(z);
  if (rc || !raw) return rc ? rc : ZADA_E_INVALID;
  Ctx *c = &z->c;
  uint32_t reg = 0;
  const uint64_t piece = 2ull << 30;
  if (n && ((uintptr_t)d_in & 15) != 0) { c->err = "zada_crc32_device: the buffer must be 16-byte aligned"; return ZADA_E_INVALID; }
  if (n && (rc = ensure_crc_workspace(c, n < piece ? n : piece))) return rc;
  for (uint64_t o = 0; o < n; o += piece) {
    const uint64_t k = n - o < piece ? n - o : piece;
    if ((rc = crc_launch(c, (const uint8_t *)d_in + o, k)) || (rc = crc_finish(c, k, &reg))) return finish_call(c, rc);
  }
  *raw = reg;
  return 0;
}
uint64_t zada_bz2_last_blocks(zada_ctx *z, uint64_t *dst, uint64_t cap_items) { return z ? bz2_last_blocks(&z->c, dst, cap_items) : 0; }

int zada_deflate_device(zada_ctx *z, int method, const void *d_in, uint64_t n, void *d_out, uint64_t cap, uint64_t *out_len,
                        uint32_t *crc_inout) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  if (n > ((uint64_t)c->knob_span_mib << 20) && ((uintptr_t)d_in & 15) == 0) {
    uint64_t ol = 0;
    rc = finish_call(c, deflate_spans(c, method, (const uint8_t *)d_in, nullptr, n, (uint8_t *)d_out, nullptr, cap, &ol, crc_inout, nullptr, nullptr));
    if (out_len && rc >= 0) *out_len = ol;
    return rc;
  }
  const uint8_t *src = (const uint8_t *)d_in;
  if (((uintptr_t)d_in & 15) != 0 && n) {              // the kernels read 16 bytes at a time
    rc = ensure_rin(c, n);
    if (rc) return rc;
    hipMemcpyAsync(c->ws.rin_own, d_in, n, hipMemcpyDeviceToDevice, c->stream);
    src = c->ws.rin_own;
  }
  uint64_t ol = 0;
  rc = finish_call(c, deflate_core(c, method, src, n, &ol, crc_inout, nullptr, nullptr));
  if (rc < 0) return rc;
  if (out_len) *out_len = ol;
  if (rc == ZADA_OK) {
    if (ol > cap) { c->err = "output buffer too small"; return ZADA_E_INVALID; }
    hipMemcpyAsync(d_out, c->ws.out, ol, hipMemcpyDeviceToDevice, c->stream);
    if (hip_check(c, hipStreamSynchronize(c->stream), "copy out")) return ZADA_E_HIP;
  }
  return rc;
}

// ---- one stream over several contexts (GPUs): see "Ranges" above and INTEGRATION.md ----

int zada_range_open(zada_ctx *z, int method, const void *d_in, uint64_t stream_size, uint64_t lo, uint64_t n, uint64_t pre, uint64_t post) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  c->tbegin(); c->tmark("begin");
  return finish_call(c, range_open(c, method, (const uint8_t *)d_in, stream_size, lo, n, pre, post));
}

int zada_range_lz(zada_ctx *z, const zada_parse_state *entry, zada_range_info *info) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  GlobalState e{0, SYNC_F, 0};
  if (entry) { e.pos = entry->pos; e.kind = entry->kind; }
  rc = finish_call(c, range_lz(c, entry ? &e : nullptr, nullptr, nullptr));
  if (rc) return rc;
  const Range &R = c->rg;
  info->atoms = R.T;
  info->exit.pos = R.exit.pos; info->exit.kind = R.exit.kind; info->exit.pad = 0;
  info->warm.pos = R.warm.pos; info->warm.kind = R.warm.kind; info->warm.pad = 0;
  info->crc_raw = R.crc_raw; info->entry_known = R.entry_known ? 1u : 0u;
  return ZADA_OK;
}

// positions travel between ranges as the low 32 bits of the position in the STREAM
__global__ void k_edge_copy(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ sp, uint32_t *__restrict__ da, uint32_t *__restrict__ dp,
                            uint32_t n, uint32_t bias) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { da[i] = sa[i]; dp[i] = sp[i] + bias; }
}

int zada_range_edges(zada_ctx *z, void *d_head_atoms, void *d_head_pos, uint32_t *n_head, void *d_tail_atoms, void *d_tail_pos, uint32_t *n_tail) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  const Range &R = c->rg;
  if (!R.open) { c->err = "no range open"; return ZADA_E_INVALID; }
  Workspace &W = c->ws;
  const uint32_t nh = R.T < LA_CAP ? (uint32_t)R.T : LA_CAP, nt = R.T < LB_CAP ? (uint32_t)R.T : LB_CAP;
  const uint32_t bias = (uint32_t)(R.lo - R.pre);                  // rin-relative -> stream (low 32 bits)
  if (nh) hipLaunchKernelGGL(k_edge_copy, dim3((nh + 255) / 256), dim3(256), 0, c->stream, W.ea_atoms + LB_CAP, W.ea_apos + LB_CAP,
                             (uint32_t *)d_head_atoms, (uint32_t *)d_head_pos, nh, bias);
  if (nt) hipLaunchKernelGGL(k_edge_copy, dim3((nt + 255) / 256), dim3(256), 0, c->stream, W.ea_atoms + LB_CAP + (R.T - nt), W.ea_apos + LB_CAP + (R.T - nt),
                             (uint32_t *)d_tail_atoms, (uint32_t *)d_tail_pos, nt, bias);
  *n_head = nh; *n_tail = nt;
  return hip_check(c, hipStreamSynchronize(c->stream), "range_edges") ? ZADA_E_HIP : ZADA_OK;
}

int zada_range_place(zada_ctx *z, uint64_t atoms_before, uint64_t atoms_total, const void *d_lb_atoms, const void *d_lb_pos, uint32_t n_lb,
                     const void *d_la_atoms, const void *d_la_pos, uint32_t n_la) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  Range &R = c->rg;
  if (!R.open) { c->err = "no range open"; return ZADA_E_INVALID; }
  if (n_lb > LB_CAP || n_la > LA_CAP) { c->err = "range_place: too many neighbour atoms"; return ZADA_E_INVALID; }
  Workspace &W = c->ws;
  const uint32_t bias = 0u - (uint32_t)(R.lo - R.pre);             // stream (low 32 bits) -> rin-relative
  if (n_lb) hipLaunchKernelGGL(k_edge_copy, dim3((n_lb + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)d_lb_atoms, (const uint32_t *)d_lb_pos,
                               W.ea_atoms + LB_CAP - n_lb, W.ea_apos + LB_CAP - n_lb, n_lb, bias);
  if (n_la) hipLaunchKernelGGL(k_edge_copy, dim3((n_la + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)d_la_atoms, (const uint32_t *)d_la_pos,
                               W.ea_atoms + LB_CAP + R.T, W.ea_apos + LB_CAP + R.T, n_la, bias);
  return finish_call(c, range_place(c, atoms_before, atoms_total, n_lb, n_la));
}

int zada_range_analyze(zada_ctx *z) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  if (!c->rg.open || !c->rg.placed) { c->err = "range_analyze: range not placed"; return ZADA_E_INVALID; }
  return finish_call(c, entropy_analyze(c));
}

int zada_range_choose(zada_ctx *z, const void *carry_in, void *carry_out, uint64_t *bit_begin, uint64_t *bit_end) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  Range &R = c->rg;
  if (!R.open || !R.analyzed) { c->err = "range_choose: range not analysed"; return ZADA_E_INVALID; }
  if (carry_in) memcpy(&R.carry_in, carry_in, sizeof(ChooserCarry));
  else { R.carry_in = ChooserCarry(); R.carry_in.last_type = BT_RESERVED; R.carry_in.cur_eob = 7u << 16; }   // the stream starts here
  hipStreamWaitEvent(c->stream, c->ev_out, 0);
  rc = finish_call(c, entropy_choose(c));
  if (rc) return rc;
  if (carry_out) memcpy(carry_out, &R.carry_out, sizeof(ChooserCarry));
  if (bit_begin) *bit_begin = R.carry_in.pos;
  if (bit_end) *bit_end = R.carry_out.pos;
  return ZADA_OK;
}

int zada_range_emit(zada_ctx *z, void *d_out, uint64_t cap, uint64_t *nbytes) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  Range &R = c->rg;
  if (!R.open || !R.chosen) { c->err = "range_emit: nothing chosen"; return ZADA_E_INVALID; }
  const uint64_t nb = (R.co.total_bits + 7) / 8;
  if (nb > cap) { c->err = "output buffer too small"; return ZADA_E_INVALID; }
  rc = finish_call(c, entropy_emit(c, (uint8_t *)d_out));
  if (rc) return rc;
  if (hip_check(c, hipStreamSynchronize(c->stream), "range_emit")) return ZADA_E_HIP;
  c->tmark("end"); c->tend();
  if (nbytes) *nbytes = nb;
  return ZADA_OK;
}

uint32_t zada_crc32_combine(uint32_t reg, uint32_t raw, uint64_t len) { return crc32_advance(reg, len) ^ raw; }

int zada_deflate_batch(zada_ctx *z, int method, int count, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out,
                       const uint64_t *cap, uint64_t *out_len, uint32_t *crc, int *rc) {
  if (!z || count < 0) return ZADA_E_INVALID;
  int prc = prepare(z);
  if (prc) return prc;
  Ctx *c = &z->c;
  int worst = 0;
  // Entries of up to 4 MiB go through ONE launch sequence, as many at a time as the workspace takes (batch_core); larger
  // ones fill the GPU by themselves and are compressed one after the other.
  const bool batchable = method >= ZADA_DEFLATE_FIXED && method <= ZADA_DEFLATE_3;
  std::vector<int> group;
  uint64_t gbytes = 0;
  auto flush_group = [&]() {
    if (group.empty()) return;
    int r = group.size() == 1 ? 1 : finish_call(c, batch_core(c, method, group.data(), (uint32_t)group.size(), in, n, out, cap, out_len, crc, rc));
    if (group.size() == 1 || r == ZADA_E_NOMEM) {            // a single entry, or no room for the batch tables: one by one
      for (int i : group) { rc[i] = zada_deflate(z, method, in[i], n[i], out[i], cap[i], &out_len[i], crc ? &crc[i] : nullptr, nullptr, nullptr); if (rc[i] < 0) worst = rc[i]; }
    } else if (r < 0) { for (int i : group) rc[i] = r; worst = r; }
    else { for (int i : group) if (rc[i] < 0) worst = rc[i]; }
    group.clear(); gbytes = 0;
  };
  for (int i = 0; i < count; i++) {
    if (batchable && n[i] <= BATCH_ENTRY_MAX) {
      const uint64_t slot = ((n[i] ? n[i] : 1) + 32767) & ~32767ull;
      if (gbytes + slot > ((uint64_t)c->knob_batch_mib << 20)) flush_group();
      group.push_back(i); gbytes += slot;
    } else {
      rc[i] = zada_deflate(z, method, in[i], n[i], out[i], cap[i], &out_len[i], crc ? &crc[i] : nullptr, nullptr, nullptr);
      if (rc[i] < 0) worst = rc[i];
    }
  }
  flush_group();
  return worst;
}

int zada_compress_data(zada_ctx *z, int method, const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *out_len,
                       uint32_t *crc_out, uint16_t *zip_type) {
  uint32_t crc = 0xFFFFFFFFu;                                   // Init, zip-compress.adb:144
  const bool bz = method >= ZADA_BZIP2_1 && method <= ZADA_BZIP2_3;                  // :204-209 (bzip2_code = 12, zip.ads:502)
  const bool lz = method >= ZADA_LZMA_0 && method <= ZADA_LZMA_3;                    // :211-216 (lzma_code = 14, zip.ads:503)
  int rc = bz ? zada_bzip2(z, method, in, n, out, cap, out_len, &crc, nullptr, nullptr)
         : lz ? zada_lzma(z, method, in, n, out, cap, out_len, &crc, nullptr, nullptr)
              : zada_deflate(z, method, in, n, out, cap, out_len, &crc, nullptr, nullptr);
  if (rc < 0 || rc == ZADA_ABORTED) return rc;
  *zip_type = bz ? 12 : lz ? 14 : 8;
  crc = ~crc;                                                   // Final :218
  if (rc == ZADA_INEFFICIENT) {                                 // :224-237 Store_data; the CRC of the same bytes is unchanged
    if (cap < n) { z->c.err = "output buffer too small"; return ZADA_E_INVALID; }
    memcpy(out, in, n);
    *out_len = n; *zip_type = 0;
  }
  *crc_out = crc;
  return ZADA_OK;
}

int zada_lz77_tokens(zada_ctx *z, int method, const uint8_t *in, uint64_t n, uint32_t *tokens, uint64_t cap, uint64_t *ntok) {
  int rc = prepare(z);
  if (rc) return rc;
  Ctx *c = &z->c;
  rc = ensure_rin(c, n);
  if (rc) return rc;
  if (n) hipMemcpyAsync(c->ws.rin_own, in, n, hipMemcpyHostToDevice, c->stream);
  c->tbegin(); c->tmark("begin");
  rc = range_open(c, method, c->ws.rin_own, n, 0, n, 0, 0);
  if (!rc) rc = range_lz(c, nullptr, nullptr, nullptr);
  if (finish_call(c, rc)) return rc;
  c->tmark("end"); c->tend();
  *ntok = c->rg.T;
  const uint64_t k = c->rg.T < cap ? c->rg.T : cap;
  if (k) hipMemcpyAsync(tokens, c->ws.ea_atoms + LB_CAP, k * 4, hipMemcpyDeviceToHost, c->stream);
  return hip_check(c, hipStreamSynchronize(c->stream), "tokens out") ? ZADA_E_HIP : ZADA_OK;
}

int zada_last_blocks(zada_ctx *z, uint64_t *rec, uint64_t cap_blocks, uint64_t *nblocks) {
  if (!z) return ZADA_E_INVALID;
  // (fetched on demand: the records stay in the workspace until the next call; copying them after every call cost a
  // device-to-host round trip inside the timed path)
  Ctx *c = &z->c;
  const uint64_t nb = c->last_nblocks;
  *nblocks = nb;
  const uint64_t k = nb < cap_blocks ? nb : cap_blocks;
  if (k == 0) return ZADA_OK;
  if (hipSetDevice(c->device) != hipSuccess) return ZADA_E_HIP;
  std::vector<EmitRec> he(k); std::vector<BlockRange> hb(k);
  hipMemcpyAsync(he.data(), c->ws.emit, k * sizeof(EmitRec), hipMemcpyDeviceToHost, c->stream);
  hipMemcpyAsync(hb.data(), c->ws.blocks, k * sizeof(BlockRange), hipMemcpyDeviceToHost, c->stream);
  if (hip_check(c, hipStreamSynchronize(c->stream), "trace")) return ZADA_E_HIP;
  const uint64_t g0 = c->rg.G - c->rg.n_lb;            // index in the stream of element 0 of the range's local atom array
  for (uint64_t i = 0; i < k; i++) { rec[4 * i] = g0 + hb[i].first; rec[4 * i + 1] = hb[i].count; rec[4 * i + 2] = he[i].fmt; rec[4 * i + 3] = he[i].cost_bits; }
  return ZADA_OK;
}

// The similarity tests of the Taillaule splitter in the last call, as the reference's trace lists them
// (zip-compress-deflate.adb:480-488, 1384-1390): rec[3*i+0..2] = atom (index in the stream) at which a window was
// compared with the reference descriptor, the L1 distance of the tweaked length vectors, the step level that cut there
// (1: 6000-atom steps / threshold 420, 2: 3000 / 430, 3: 750 / 2050; 0 = similar, no cut).
int zada_last_trace(zada_ctx *z, uint64_t *rec, uint64_t cap, uint64_t *count) {
  if (!z) return ZADA_E_INVALID;
  Ctx *c = &z->c;
  const Range &R = c->rg;
  *count = 0;
  if (!R.analyzed || R.method == ZADA_DEFLATE_FIXED || R.nflush == 0) return ZADA_OK;
  if (hipSetDevice(c->device) != hipSuccess) return ZADA_E_HIP;
  std::vector<uint32_t> h((size_t)R.nflush * SLOTS * 2);
  hipMemcpyAsync(h.data(), c->ws.cut_trace, h.size() * 4, hipMemcpyDeviceToHost, c->stream);
  if (hip_check(c, hipStreamSynchronize(c->stream), "trace")) return ZADA_E_HIP;
  uint64_t k = 0;
  for (uint32_t j = 0; j < R.nflush; j++)
    for (uint32_t s = 1; s < SLOTS; s++) {
      const uint32_t d = h[((size_t)j * SLOTS + s) * 2], lvl = h[((size_t)j * SLOTS + s) * 2 + 1];
      if (d == 0xFFFFFFFFu) continue;
      if (k < cap) { rec[3 * k] = (R.j0 + j) * FLUSH + (uint64_t)MIN_STEP * s; rec[3 * k + 1] = d; rec[3 * k + 2] = lvl; }
      k++;
    }
  *count = k;
  return ZADA_OK;
}

int zada_last_timing(zada_ctx *z, const char **names, float *ms, int cap) {
  if (!z) return 0;
  int k = 0;
  for (auto &t : z->c.timing) { if (k >= cap) break; names[k] = t.first; ms[k] = t.second; k++; }
  return k;
}

}  // extern "C"
