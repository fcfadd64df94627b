// zada_internal.h -- context, workspace layout and shared constants of libzada_hip.so
#pragma once
#ifndef ZADA_NLEVELS
#define ZADA_NLEVELS 2
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "zada_logic.h"

namespace zada {

constexpr int ZADA_E_HIP_ = -3;

// ---- LZ stage geometry ----
#ifndef ZADA_PCHUNK
#define ZADA_PCHUNK 512
#endif
constexpr uint32_t PCHUNK = ZADA_PCHUNK;
static_assert(PCHUNK > 258, "k_fix_forward: a 258-byte step crosses at most one chunk boundary");                 // bytes parsed per lane (speculative chunk)
constexpr uint32_t PTOK_STRIDE = PCHUNK + 640;    // token slots per chunk (a parse may overrun its chunk by < 600 B)
constexpr uint32_t CRC_CHUNK = 4096, CRC_SUB = 256;   // CRC: one lane per 256 B, folded to one value per 4 KiB on the device
constexpr uint64_t IN_PAD = 1024;                 // zero bytes kept after the input

constexpr int NLEVELS = ZADA_NLEVELS;               // hash levels 4 .. 3+NLEVELS; the last one is the chain the match kernel walks
struct LevelPtrs { uint16_t *prev[NLEVELS]; uint16_t *tails[NLEVELS]; };
// per-position planes: d[l] = distance of the nearest position sharing 3 + l bytes (0 = none); dlim = Dfull | Dquarter << 16
// Last level of the nested chains: sorted order of every segment, and per position its index in it and the number
// of members of its bucket before it (inside the segment).
struct RunPtrs { uint16_t *S, *idx, *cnt; };
struct DistPlanes { uint16_t *d[NLEVELS]; uint32_t *dlim; };

// ---- entropy stage geometry (zip-compress-deflate.adb:942, 1294, 1313) ----
constexpr uint32_t FLUSH = 65536;                 // atoms per Flush_half_buffer
constexpr uint32_t MIN_STEP = 750, HALF_SLIDER = 2048, SLIDER = 4096;
constexpr uint32_t SLOTS = 88;                    // descriptor slots per flush segment (1 initial + <= 84 sliding)
constexpr uint32_t MAXBLK_PER_SEG = 86;
constexpr uint32_t TILE = 2048;                   // atoms per emission tile

enum { FMT_STORED = 0, FMT_FIXED = 1, FMT_DYN1 = 2, FMT_DYN2 = 3, FMT_RECYCLE = 4 };
enum { BT_STORED = 0, BT_FIXED = 1, BT_DYNAMIC = 2, BT_RESERVED = 3 };

struct BlockRange { uint32_t first, count; uint32_t last_flush; uint32_t pad; };

// Everything the chooser and the emitters need to know about one candidate block.
struct BlockInfo {
  uint32_t stats[320];          // 288 lit/len (EOB preset to 1) + 32 distance counts
  uint8_t bl1[320], bl2[320];   // code lengths: plain / after Tweak_for_better_RLE
  uint8_t truc1[20], truc2[20]; // 19 code-length-code lengths + a_non_zero
  uint32_t hdr1_bits, hdr2_bits;            // Put_Compression_Structure cost (:682-685)
  uint64_t fixed_data, dyn1_data, dyn2_data; // bits of the LZ data alone under each code set
  uint32_t bytes;               // uncompressed bytes covered
  uint32_t stored_possible;
};

// Decision record written by the sequential chooser.
struct EmitRec {
  uint64_t hdr_bitpos;          // dynamic header position (after the 3 block-header bits)
  uint64_t data_bitpos;         // first bit of the LZ data
  uint64_t cost_bits;           // optimal_format_bits (:1234), for the trace
  uint32_t fmt;                 // FMT_*
  int32_t code_block;           // block whose code table is in force (-1 = fixed table)
  uint32_t code_variant;        // 1 or 2 (bl1 / bl2 of code_block)
  uint32_t tile_base;           // first emission tile of this block
  // what precedes the block in the stream when it opens a new Deflate block (written by k_emit_prefix, not by the
  // sequential chooser): [end-of-block code of the block being finished] BFINAL, BTYPE
  uint64_t pre_pos;             // bit position of that prefix
  uint32_t pre_eob;             // (len << 16) | code of the end-of-block symbol to write first, 0 = none
  uint32_t pre_flags;           // bit 0: the prefix exists; bit 1: BFINAL; bits 2-3: BTYPE
};

struct StoredPiece { uint64_t dst_byte; uint32_t src_byte, nbytes; };

struct ChooserOut {
  uint64_t total_bits;
  uint32_t n_tiles, n_pieces, n_blocks, overflow;
};

struct Workspace {
  uint64_t cap_n = 0;           // input capacity in bytes
  uint8_t *in = nullptr;
  uint16_t *lprev[NLEVELS] = {}, *ltails[NLEVELS] = {};   // per level: chain links (16-bit distances) / per-segment bucket tails
  uint16_t *S3 = nullptr; uint8_t *T3 = nullptr; uint32_t *bsc3 = nullptr;   // 15-bit hash order of every segment (positions, tags, buckets)
  uint16_t *dplane[NLEVELS] = {}; uint32_t *dlim = nullptr;  // DistPlanes
  uint16_t *SK = nullptr, *idxK = nullptr, *cntK = nullptr;   // RunPtrs
  MatchPair *M = nullptr;                    // match tables; alias: atoms / apos (two halves of the same buffer)
  uint32_t *atoms = nullptr, *apos = nullptr;
  uint32_t *spec_tok = nullptr, *fix_tok = nullptr;
  uint32_t *spec_cnt = nullptr, *fix_cnt = nullptr, *take_from = nullptr, *start_pos = nullptr;
  uint32_t *counts = nullptr, *offsets = nullptr, *scan_sums = nullptr;
  uint32_t *Fbits = nullptr, *Lbits = nullptr;
  ExitState *spec_exits = nullptr, *true_exits = nullptr;
  uint8_t *dirty[2] = {nullptr, nullptr};
  uint32_t *n_changed = nullptr;
  uint32_t *blk_demand = nullptr, *n_demand = nullptr;   // demanded match records per k_match block / in total
  uint32_t *dbits = nullptr;                             // one bit per position: marked for the next demand pass
  uint8_t *chg = nullptr;                                // per parse chunk: a guess it used turned out different
  // entropy stage
  uint8_t *descr = nullptr;                  // [nseg][SLOTS][320]
  uint32_t *seg_nblk = nullptr, *seg_cut = nullptr, *seg_blk_off = nullptr;   // cuts [nseg][MAXBLK_PER_SEG]
  BlockRange *blocks = nullptr;
  BlockInfo *binfo = nullptr;
  EmitRec *emit = nullptr;
  uint64_t *chrec = nullptr;                 // ChRec[nblocks] (128 B each)
  uint32_t *codes = nullptr;                 // [nblocks+1][320]  (len << 16 | code); last = fixed table
  StoredPiece *pieces = nullptr;
  uint32_t *tile_block = nullptr;            // tile -> block
  uint64_t *tile_bitpos = nullptr;
  uint32_t *tile_bits = nullptr;
  ChooserOut *chooser = nullptr;
  uint32_t *crc_lvl[4] = {nullptr, nullptr, nullptr, nullptr}, *crc_mat = nullptr;
  bool crc_mat_ready = false;
  uint64_t *dbg = nullptr;
  uint8_t *out = nullptr;
  uint64_t cap_blocks = 0, cap_tiles = 0, cap_pieces = 0, cap_out = 0;
  std::vector<void *> allocs;
};

// host side of a CRC-32 in flight (crc_launch / crc_finish)
constexpr uint32_t CRC_HOST_TOP = 4096;           // top-level CRC values per call (one per MiB of input)
struct CrcPending { uint32_t nsub = 0, nfull0 = 0, cnt[4] = {0, 0, 0, 0}, nrest[4] = {0, 0, 0, 0}; };

constexpr uint64_t STAGE_BYTES = 8ull << 20;
struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: CRC-32, next to the LZ stage
  hipEvent_t ev_input = nullptr, ev_out = nullptr;
  CrcPending crc;
  uint32_t *crc_host = nullptr;                     // pinned: [CRC_HOST_TOP] top-level values, then 4 x 16 leftovers
  uint8_t *stage[2] = {nullptr, nullptr};            // pinned staging buffers of the host-buffer entry points (copy_in / copy_out)
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  Workspace ws;
  std::string err;
  int parse_rounds = 0, demand_rounds = 0;
  // timing
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<const char *, hipEvent_t>> marks;
  std::vector<std::pair<const char *, float>> timing;
  bool timing_on = true;
  // last-call block trace: fetched from the workspace when zada_last_blocks asks for it
  uint32_t last_nblocks = 0;
  // knobs read from the environment once, when the context is created (tests set them before zada_create, or call
  // zada_set_knob)
  int knob_budget = -1;             // ZADA_BUDGET: rounds of chain steps per position in the first match pass (0 = unbounded, -1 = default)
  int knob_max_demand_rounds = 12;  // ZADA_MAX_DEMAND_ROUNDS
  int knob_batch_streams = 4;       // ZADA_BATCH_STREAMS
  void tmark(const char *name);
  void tbegin();
  void tend();
};

int hip_check(Ctx *c, hipError_t e, const char *what);
int ensure_workspace(Ctx *c, uint64_t n);
int lz_stage(Ctx *c, int level, uint64_t n, uint32_t *ntok_out);
int huff_stage(Ctx *c, int method, uint64_t n, uint32_t T, uint64_t *total_bits);
int crc_launch(Ctx *c, uint64_t n);
int crc_finish(Ctx *c, uint64_t n, uint32_t *crc_inout);
void exclusive_scan_u32(hipStream_t st, const uint32_t *d_in, uint32_t *d_out, uint32_t *d_sums, uint32_t *d_total, uint32_t n);

}  // namespace zada
