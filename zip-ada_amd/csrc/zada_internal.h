// zada_internal.h -- context, workspace layout and shared constants of libzada_hip.so
#pragma once
#ifndef ZADA_NLEVELS
#define ZADA_NLEVELS 2
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <functional>
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
constexpr uint32_t FIX_STRIDE_SMALL = 128;        // ... of the splice's tokens to start with (1 instead of 9 bytes per input byte): a splice meets the speculative parse after a few tokens
constexpr uint32_t CRC_CHUNK = 4096, CRC_SUB = 256;   // CRC: one lane per 256 B, folded to one value per 4 KiB on the device
constexpr uint64_t IN_PAD = 1024;                 // zero bytes kept after the input

constexpr int NLEVELS = ZADA_NLEVELS;               // hash levels 4 .. 3+NLEVELS; the last one is the chain the match kernel walks
struct LevelPtrs { uint16_t *prev[NLEVELS]; uint16_t *tails[NLEVELS]; };
// per-position planes: d[l] = distance of the nearest position sharing 3 + l bytes (0 = none); dlim = Dfull | Dquarter << 16
// Last level of the nested chains: sorted order of every segment, and per position its index in it and the number
// of members of its bucket before it (inside the segment).
struct RunPtrs { uint16_t *S, *idx, *cnt; };
// dlim is sparse: only the members of the few buckets with a quarter chain's worth of members have a limit (k_bucket_limits); one bit per position
// (dlim_bits) says whether dlim [p] holds one -- "no limit" is the absence of the bit, not 4 bytes per position written by a memset and read back.
struct DistPlanes {
  uint16_t *d[NLEVELS]; uint32_t *dlim; uint32_t *dlim_bits;
  __device__ __forceinline__ uint32_t limits(uint64_t p) const { return ((dlim_bits[p >> 5] >> (p & 31)) & 1u) ? dlim[p] : 0xFFFFFFFFu; }
};

// ---- entropy stage geometry (zip-compress-deflate.adb:942, 1294, 1313) ----
constexpr uint32_t FLUSH = 65536;                 // atoms per Flush_half_buffer
constexpr uint32_t MIN_STEP = 750, HALF_SLIDER = 2048, SLIDER = 4096;
constexpr uint32_t SLOTS = 88;                    // descriptor slots per flush segment (1 initial + <= 84 sliding)
constexpr uint32_t MAXBLK_PER_SEG = 86;
constexpr uint32_t TILE = 2048;                   // atoms per emission tile

enum { FMT_STORED = 0, FMT_FIXED = 1, FMT_DYN1 = 2, FMT_DYN2 = 3, FMT_RECYCLE = 4 };
enum { BT_STORED = 0, BT_FIXED = 1, BT_DYNAMIC = 2, BT_RESERVED = 3 };

// A candidate block: atoms [first, first + count) of the range's LOCAL atom array (see EntropyView).
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

// Decision record written by the sequential chooser.  Bit positions are relative to the range's output base
// (Range::base_bits, a multiple of 8).
struct EmitRec {
  uint64_t hdr_bitpos;          // dynamic header position (after the 3 block-header bits)
  uint64_t data_bitpos;         // first bit of the LZ data
  uint64_t cost_bits;           // optimal_format_bits (:1234), for the trace
  uint32_t fmt;                 // FMT_*
  int32_t code_block;           // block whose code table is in force (CODE_FIXED = fixed table, CODE_CARRIED = the table
                                // that was in force when the range began)
  uint32_t code_variant;        // 1 or 2 (bl1 / bl2 of code_block)
  uint32_t tile_base;           // first emission tile of this block
  // what precedes the block in the stream when it opens a new Deflate block (written by k_emit_prefix, not by the
  // sequential chooser): [end-of-block code of the block being finished] BFINAL, BTYPE
  uint64_t pre_pos;             // bit position of that prefix
  uint32_t pre_eob;             // (len << 16) | code of the end-of-block symbol to write first, 0 = none
  uint32_t pre_flags;           // bit 0: the prefix exists; bit 1: BFINAL; bits 2-3: BTYPE
};
constexpr int32_t CODE_FIXED = -1, CODE_CARRIED = -2;

struct StoredPiece { uint64_t dst_byte; uint32_t src_byte, nbytes; };   // src_byte: offset in the range's input (Range::rin)

// The cross-block state of the reference's encoder (zip-compress-deflate.adb:722, 993-997) at a range boundary: what
// Send_as_block needs to know about the blocks before -- last_block_type, block_to_finish, last_block_marked,
// curr_descr (its code lengths; the codes follow from them) -- and the bit position in the stream.
struct ChooserCarry {
  uint64_t pos;                 // bits written so far (global, from the start of the stream)
  int32_t last_type, block_to_finish, last_marked;
  uint32_t cur_eob;             // (length << 16) | code of symbol 256 under curr_descr
  uint8_t bl[320];              // code lengths of curr_descr (meaningful when last_type is BT_DYNAMIC)
  uint32_t pad[2];
};
static_assert(sizeof(ChooserCarry) == 352, "ChooserCarry is exchanged between ranks as bytes");

struct ChooserOut {
  uint64_t total_bits;          // position after the range's last bit, relative to the range's output base
  uint32_t n_tiles, n_pieces, n_blocks, overflow;
};

// How the entropy kernels see the atoms of a range: one LOCAL array = [look-behind atoms of the previous range]
// [the range's own atoms][look-ahead atoms of the next ranges, up to the end of the last flush the range owns].
// The reference flushes its LZ buffer every 65 536 atoms counted from the start of the STREAM
// (zip-compress-deflate.adb:1424-1432); a range owns the flushes whose first atom is one of its own.
// A batch of independent streams (Zip entries) through ONE launch sequence: every entry has its own flush grid, chooser
// state and output.  The flushes of all entries are listed in a table (slots reserved from the entries' byte lengths; a slot
// may stay empty when the entry has fewer atoms than bytes).
struct FlushGeom {
  uint32_t F, to;               // first and last atom of the flush (index in the atom array)
  uint32_t gj;                  // number of the flush inside its entry
  uint32_t flags;               // FG_*
  uint32_t end_byte;            // position behind the entry's last byte (the position array has no sentinel per entry)
  uint32_t entry;               // index of the entry in the batch
  uint32_t pad[2];
};
enum { FG_EMPTY = 1, FG_LAST_PARTIAL = 2, FG_ENTRY_FIRST = 4, FG_ENTRY_LAST = 8 };
// BlockRange::last_flush carries flags: bit 0 = the reference's last_flush (BFINAL candidate); batches: bit 1 = first block
// of its entry, bit 2 = last block of its entry; BlockRange::pad = end_byte of the entry for an entry's last block
enum { BR_LAST_FLUSH = 1, BR_ENTRY_FIRST = 2, BR_ENTRY_LAST = 4 };
struct EntOut {                 // what k_ch_layout leaves per entry of a batch
  uint64_t bits;                // bits of the entry's stream before the epilogue (relative to the entry's first bit)
  uint32_t eob;                 // end-of-block code to write in the epilogue ((length << 16) | code), 0 = none
  uint32_t fake;                // 1: the fake final fixed block follows (:1624-1634)
};

struct EntropyView {
  const FlushGeom *ftab;        // batch: the flush table (nflush slots); nullptr: one range, flushes computed from foff / j0
  const uint32_t *atoms, *apos; // local arrays
  uint32_t foff;                // local index of the first atom of the first owned flush
  uint32_t nflush;              // owned flushes
  uint32_t lvalid;              // atoms in the local array
  uint32_t stream_final;        // the array ends where the stream ends
  uint64_t j0;                  // number of the first owned flush in the stream (parity and "first flush" matter)
};

// Look-behind / look-ahead capacity of the local atom array
constexpr uint32_t LB_CAP = HALF_SLIDER, LA_CAP = FLUSH;

struct Workspace {
  // ---- LZ stage: sized for one shard (cap_n bytes of input buffer) ----
  uint64_t cap_n = 0;           // shard buffer capacity in bytes
  uint8_t *in = nullptr;
  uint16_t *lprev[NLEVELS] = {}, *ltails[NLEVELS] = {};   // per level: chain links (16-bit distances) / per-segment bucket tails
  uint16_t *S3 = nullptr; uint8_t *T3 = nullptr; uint32_t *bsc3 = nullptr;   // 15-bit hash order of every segment (positions, tags, buckets)
  uint32_t *segmax = nullptr;                // largest 15-bit bucket of every segment
  uint16_t *heavy = nullptr;                 // per segment 2048 x u16: the number of its heavy 15-bit buckets, then their hashes (zada_lz.hip)
  uint32_t *bloom4 = nullptr;                // per segment 2^17 bits: which four-byte values it holds (k_bloom4, asked by k_cross_dist)
  uint32_t *cd_list = nullptr; uint64_t cd_cap = 0;   // positions whose level-4 walk k_cross_dist's sweep left open (its second pass takes them)
  uint16_t *dplane[NLEVELS] = {}; uint32_t *dlim = nullptr, *dlim_bits = nullptr;  // DistPlanes
  uint16_t *SK = nullptr, *idxK = nullptr, *cntK = nullptr;   // RunPtrs
  MatchPair *M = nullptr;                    // match tables
  uint32_t *spec_tok = nullptr, *fix_tok = nullptr;
  uint32_t fix_stride = 0;                   // token slots per chunk in fix_tok: FIX_STRIDE_SMALL, or PTOK_STRIDE once a splice has needed more (lz_shard)
  uint32_t *spec_cnt = nullptr, *fix_cnt = nullptr, *take_from = nullptr, *start_pos = nullptr;
  uint32_t *counts = nullptr, *offsets = nullptr, *scan_sums = nullptr;
  uint32_t *Fbits = nullptr, *Lbits = nullptr;
  ExitState *spec_exits = nullptr, *true_exits = nullptr;
  uint8_t *dirty[2] = {nullptr, nullptr};
  uint32_t *n_changed = nullptr;
  uint32_t *blk_demand = nullptr, *n_demand = nullptr;   // demanded match records per k_match block / in total
  uint32_t *dbits = nullptr;                             // one bit per position: marked for the next demand pass
  uint8_t *chg = nullptr;                                // per parse chunk: a guess it used turned out different
  uint32_t *crc_lvl[4] = {nullptr, nullptr, nullptr, nullptr}, *crc_mat = nullptr;
  bool crc_mat_ready = false;
  uint64_t *dbg = nullptr;
  std::vector<void *> allocs;   // of the LZ group
  // ---- entropy stage: sized for the atoms of one range (cap_atoms) ----
  uint64_t cap_atoms = 0, cap_flush = 0;
  uint32_t *ea_atoms = nullptr, *ea_apos = nullptr;   // local atom array: LB_CAP slots, the range's atoms, LA_CAP slots (+ sentinel)
  uint8_t *descr = nullptr;                  // [nflush][SLOTS][320]
  uint32_t *seg_nblk = nullptr, *seg_cut = nullptr, *seg_blk_off = nullptr;   // cuts [nflush][MAXBLK_PER_SEG]
  uint32_t *cut_trace = nullptr;             // [nflush][SLOTS][2]: similarity distance and cut level at every test point
  BlockRange *blocks = nullptr;
  BlockInfo *binfo = nullptr;
  EmitRec *emit = nullptr;
  uint64_t *chrec = nullptr;                 // ChRec[nblocks] (128 B each)
  uint64_t *chw = nullptr;                   // ChW[nblocks] (32 B each): the chooser's per-block results
  uint32_t *piece_base = nullptr;            // first piece of a stored block
  uint32_t *codes = nullptr;                 // [nblocks+2][320]  (len << 16 | code); the last two = fixed table, carried table
  StoredPiece *pieces = nullptr;
  uint32_t *tile_block = nullptr;            // tile -> block
  uint64_t *tile_bitpos = nullptr;
  uint32_t *tile_bits = nullptr;
  ChooserOut *chooser = nullptr;
  ChooserCarry *carry = nullptr;             // [2]: in, out
  uint32_t *scan2 = nullptr, *total2 = nullptr;   // scan scratch of the entropy stage
  uint64_t cap_blocks = 0, cap_tiles = 0, cap_pieces = 0;
  uint32_t *blk_entry = nullptr;             // batches: entry of a block
  std::vector<void *> en_allocs;
  std::vector<void *> crc_allocs;                    // CRC-32 levels (their own group: the BZip2 path needs them without the Deflate entropy workspace)
  uint64_t cap_crc = 0;
  // ---- batches of entries (zada_deflate_batch): per entry / per flush slot / per segment tables ----
  uint64_t cap_ent = 0, cap_fslots = 0, cap_bseg = 0;
  FlushGeom *ftab = nullptr;
  EntOut *ent_out = nullptr;
  uint32_t *ent_chunk0 = nullptr, *ent_fl0 = nullptr, *ent_start = nullptr, *ent_len = nullptr, *ent_bytes = nullptr, *ent_base = nullptr, *ent_crc = nullptr;
  uint32_t *segend = nullptr;
  std::vector<void *> bt_allocs;
  // ---- buffers of the host-buffer entry points ----
  uint8_t *rin_own = nullptr; uint64_t cap_rin = 0;   // the range's input, copied from the host
  uint8_t *out = nullptr; uint64_t cap_out = 0;       // the range's output
};

// One range of a stream in flight on this context (zada_range_* / deflate_core): bytes [lo, lo + n) of the stream, with
// `pre` bytes before and `post` bytes after it also resident at rin (halo for the match finder, and the bytes of the
// look-ahead atoms).
struct GlobalState { uint64_t pos; uint32_t kind, pad; };      // parser state at a history-free point: stream position, SYNC_F / SYNC_L
struct Range {
  bool open = false;
  bool batch = false;                // a batch of entries instead of a range of one stream (entropy stage per entry)
  uint32_t n_entries = 0;
  const uint8_t *rin = nullptr;      // device: stream byte lo - pre
  uint64_t lo = 0, pre = 0, n = 0, post = 0;
  bool first = true, last = true;
  int method = 0, level = 0;
  uint64_t T = 0;                    // the range's own atoms
  uint64_t T_carried = 0;            // ... of which already in the array when the LZ stage starts (carried over from the span before)
  uint64_t T_view = ~0ull;           // own atoms the entropy stage takes now (spans: whole flushes only); ~0 = all
  uint32_t n_lb = 0, n_la = 0;       // look-behind / look-ahead atoms in the local array
  GlobalState exit{}, warm{};        // exit: first history-free state at or beyond the end of the range; warm: the one at or
                                     // beyond its start, found by the warm-up parse when the entry was not known
  bool entry_known = true;
  uint64_t G = 0, T_total = 0;       // atoms of the stream before the range / in all
  bool placed = false;               // G / T_total / neighbours' atoms are known
  uint32_t nflush = 0, foff = 0, nblocks = 0;
  uint64_t j0 = 0;
  uint64_t base_bits = 0;            // output base: carry_in.pos rounded down to a byte
  ChooserCarry carry_in{}, carry_out{};
  ChooserOut co{};
  bool analyzed = false, chosen = false;
  uint32_t crc_raw = 0;              // CRC register of the range's bytes started from 0 (the linear part)
};

// host side of a CRC-32 in flight (crc_launch / crc_finish)
constexpr uint32_t CRC_HOST_TOP = 4096;           // top-level CRC values per call (one per MiB of input)
struct CrcPending { uint32_t nsub = 0, nfull0 = 0, cnt[4] = {0, 0, 0, 0}, nrest[4] = {0, 0, 0, 0}; };

constexpr uint64_t STAGE_BYTES = 8ull << 20;
constexpr int MAX_COPY_LANES = 8;
struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: CRC-32, next to the LZ stage
  hipStream_t stream_in[MAX_COPY_LANES] = {};                     // the copy lanes of an input that arrives while the LZ stage has begun (zada_deflate)
  hipEvent_t ev_in[MAX_COPY_LANES] = {};                          // ... how far each lane has come
  void *arrival = nullptr;                           // ... the copy in flight (Arrival, zada_api.hip), null when the input is resident
  hipEvent_t ev_input = nullptr, ev_out = nullptr, ev_dlim = nullptr;
  CrcPending crc;
  uint32_t *crc_host = nullptr;                     // pinned: [CRC_HOST_TOP] top-level values, then 4 x 16 leftovers
  uint8_t *stage[2 * MAX_COPY_LANES] = {};           // pinned staging buffers of the host-buffer entry points: two per copy lane (copy_in / copy_out)
  uint8_t *bstage = nullptr; uint64_t cap_bstage = 0; // pinned: a batch's packed input, then its output
  uint32_t *btab = nullptr; uint64_t cap_btab = 0;    // pinned: a batch's tables on their way to / from the device
  hipEvent_t ev_stage[2 * MAX_COPY_LANES] = {};
  Workspace ws;
  Range rg;                                          // the range in flight
  std::string err;
  int parse_rounds = 0, demand_rounds = 0;
  int atoms_grown = 0, fix_grown = 0;                // how often a call had to enlarge the atom arrays / the splice's token slots (last_timing: #atoms_grown, #fix_grown)
  bool lz_attrs_set = false;
  void *bz = nullptr;                                // BZip2 state (zada_bz2.hip), made on first use
  void *lz_tab = nullptr; size_t cap_lz_tab = 0;     // LZMA (zada_lzma.hip): job table + results
  void *lz_save = nullptr; size_t cap_lz_save = 0;   // ... the coder's state between the launches of one stream
  int lzma_launches = 0;                             // launches the last chunked LZMA call took
  std::vector<uint8_t> lz_resume;                    // zada_lzma_import_state: the coder's state the next zada_lzma call goes on from (one stream)
  uint64_t lz_last_n = 0, lz_last_out_off = 0;       // the last zada_lzma call: its input length and where its stream lies in the context's buffer (zada_lzma_export_state)
  void *bt4 = nullptr;                               // ... the BT4 match producer's buffers (zada_bt4.hip), made on first use
  uint32_t bt4_buckets = 0, bt4_long = 0, bt4_overflow = 0;   // last producer run: hash-4 buckets, long ones among them, overflow blocks booked
  uint32_t bt4_pool_grown = 0;                       // the overflow pool of a stream's match sets was enlarged between two segments (since the context was made)
  int knob_lzma_pool_fixed = 0;                      // test knob "lzma_pool_fixed": 1 = the pool never grows between segments (the round-4 behaviour: run out, start again)
  uint32_t bt4_reruns = 0;                           // walks repeated with a larger overflow pool (since the context was made)
  int knob_lzma_waves = 0;                           // LZMA_3, one stream: waves of its workgroup (0 = by the call: 4 for zada_lzma, 1 = the chain's wave alone)
  int knob_lzma_segment = 0;                         // LZMA_3, one stream: log2 of the positions per producer segment (0 = 20, -1 = no segments)
  int knob_lzma_pool = 0;                            // LZMA_3 test knob: blocks of the match sets' overflow pool to start with (0 = by size)
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
  int knob_atoms_pct = 50;          // "atoms_pct": the atom arrays of a stream start with room for this many atoms per 100 input bytes (they grow when a shard has more: one atom per byte is the worst case, the benchmark stream has 0.3)
  int knob_fix_stride = 0;          // "fix_stride" (test knob): token slots per chunk the splice starts with (0 = FIX_STRIDE_SMALL)
  int knob_cd_list_cap = 0;         // "cd_list_cap" (test knob): entries of k_cross_dist's list of open walks (0 = by size; its second list a quarter of it) -- a full list leaves the walks in the sweep
  int knob_cd_filter = 1;           // "cd_filter" / ZADA_CD_FILTER: k_cross_dist asks a Bloom filter of the previous segment's four-byte values before a level-4 walk, walks at most six steps in its sweep and leaves longer walks to a second, packed pass (0: one pass, no filter, no limit -- rounds 1-5)
  int knob_exact_respec = 32768;    // "exact_respec" / ZADA_EXACT_RESPEC: lists of up to this many flagged chunks are parsed again by one wave per chunk with the exact search inside the parse (0: never -- the lane-per-chunk parse with guesses in every round, as in rounds 1-5)
  int knob_bz_pipe_prio = 0;        // "bz_pipe_prio": 1 = the worker stream of the BZip2 pipeline (entropy stage of the batch before) has the lowest priority (measured: no gain)
  int knob_link_run = 0;            // ZADA_LINK_RUN: segments per workgroup of k_prev_links (0 = by size: lz_shard)
  int knob_inner_budget = 0;        // ZADA_INNER_BUDGET: rounds for positions deep inside a match (0 = as every other position; A/B: 1 round saves 4.7 ms in k_match and costs 8.9 ms of demand searches, 2 rounds: -3.2 / +3.7)
  int knob_span_mib = 2048;         // MiB of a stream one pass takes (longer streams: spans one after the other, deflate_spans)
  int knob_batch_mib = 512;         // MiB of LZ buffer one batch of small entries may take (zada_deflate_batch)
  int knob_bz_batch_mib = 256;      // BZip2: MiB of small entries zada_bzip2_batch takes through one launch sequence
  int knob_bz_span_mib = 1024;      // BZip2: MiB of the stream whose block limits are found at a time
  int knob_bz_text_order = 1;       // BZip2: the group lists of the late sort rounds in text order (one library radix sort per build)
  int knob_bz_tail_pct = 0;         // BZip2: share of the last pass that the last (short) batch of a pipelined call takes
  int knob_bz_batch_melems = 768;   // BZip2: Mi RLE_1 bytes (summed over the sub-blocks) one batch of blocks may hold
  int knob_bz_lists = -1;           // BZip2 rotation sort: from this prefix length on, sub-blocks whose unsorted groups have at most 8 192 rows ("bz_list_rows") leave the
                                    // full sweeps for per-group sorts driven by a list (0 = never: every round sweeps; -1 = 16 for a batch that holds a stream's blocks -- a sub-block of 300 000 bytes and more --,
                                    // 8 for small entries' sub-blocks)
  int knob_bz_pipeline = 1;         // BZip2: the transforms of a batch of sub-blocks run next to the entropy stage of the batch before (0: one batch at a time)
  int knob_bz_list_rows = 0;        // BZip2: a sub-block leaves the sweeps when its unsorted groups have at most this many rows (0 = 8 192, the most a workgroup sorts)
  int knob_bz_split = 1;            // BZip2 entropy search: the long sub-blocks' four chains on four workgroups (0: one workgroup per sub-block)
  int knob_bz_small_wg = 1;         // BZip2 entropy search: short sub-blocks on small workgroups (0: every sub-block on the large ones)
  int knob_lzma_dict = 0;           // LZMA_3: dictionary_size in bytes instead of the entry's size (0 = the entry's size, as Zip.Compress.LZMA_E asks;
                                    // lzma_enc.adb's default is 32 KiB) -- this one DOES change the output: it is the reference's parameter
  int knob_lzma_chunk = 0;          // LZMA: positions one launch codes before the stream's state goes back to HBM and the caller's feedback is
                                    // called (0 = by level, about half a second per launch; -1 = the whole stream in one launch)
  int knob_shard_kib = 1 << 20;     // ZADA_SHARD_KIB: bytes of a range the LZ stage takes at a time, in KiB (multiple of 64)
  void tmark(const char *name);
  void tbegin();
  void tend();
};

extern thread_local std::string *tls_err;          // error text of a worker thread (see hip_check)
int hip_check(Ctx *c, hipError_t e, const char *what);
void bz2_destroy(Ctx *c);
int bz2_encode_device(Ctx *c, int option, const uint8_t *d_in, uint64_t n, int64_t size_hint, uint8_t *d_out, uint64_t cap, uint64_t *out_len,
                      int (*fb)(int, void *), void *user);
uint64_t bz2_last_blocks(Ctx *c, uint64_t *dst, uint64_t cap_items);
int bz2_batch_encode(Ctx *c, int option, const uint8_t *d_arena, uint32_t E, const uint64_t *starts, const uint32_t *lens, uint8_t *h_out, uint64_t cap,
                     uint64_t *out_off, uint64_t *out_bytes);
int bz2_range_open(Ctx *c, int option, const uint8_t *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t stream_total, uint64_t start, uint64_t own_end,
                   uint64_t *next_start, uint64_t *nblocks);
int bz2_range_encode(Ctx *c);
uint64_t bz2_range_table(Ctx *c, uint64_t *tab, uint64_t cap_blocks);
int bz2_range_assemble(Ctx *c, const uint8_t *choice, uint64_t nblk, uint64_t bit_begin, int flags, uint32_t footer_crc, uint8_t *d_out, uint64_t cap, uint64_t *nbytes);
// LZMA (zada_lzma.hip): one stream per job.  Offsets in bytes / tokens / ints from the bases given to lzma_launch.
struct LzmaJob {
  uint64_t in_off, n;               // the entry
  uint64_t tok_off, ntok;           // its LZ77 tokens (Level_1 / Level_2)
  uint64_t out_off, cap;            // where the stream goes (the bytes beyond cap are counted, not written)
  uint64_t verify;                  // Level_3: 1 = the coder checks every match of the sets it reads against the text (bt4_reads_behind_a_gap, zada_bt4.h)
  uint32_t sbs, hash4_size;         // String_buffer_size (lzma-encoding.adb:137-149), BT4's hash4 size (lz77.adb:1019-1032)
  int32_t level, zip_prefix;        // 0 .. 3; 1: the four bytes of zip-compress-lzma_e.adb:155-158 go first
};
// The match sets the BT4 producer leaves in HBM (zada_bt4.hip), indexed by arena position p: cnt [p] matches; match i < 7 at slot
// p * 8 + i of sl (length) / sd (distance); match i >= 7 at slot sd [p * 8 + 7] * 43 + (i - 7) of ol / od.
struct Bt4Sets { const uint8_t *cnt; const uint16_t *sl; const uint32_t *sd; const uint16_t *ol; const uint32_t *od; };
int bt4_produce(Ctx *c, const std::vector<LzmaJob> &jobs, const uint8_t *d_arena, uint64_t arena_bytes, Bt4Sets *out, std::vector<uint32_t> *weights = nullptr,
                uint32_t seg_shift = 32, uint32_t *nseg = nullptr);
int bt4_walk_segment(Ctx *c, uint32_t k, hipStream_t st);
int bt4_segments_overflowed(Ctx *c, hipStream_t st, Bt4Sets *out = nullptr);
void bt4_destroy(Ctx *c);
uint32_t lzma_string_buffer_size(int level, uint64_t dictionary_size);
uint32_t lzma_hash4_size(uint32_t sbs);
int lzma_token_ranges(Ctx *c, uint32_t E, const uint32_t *d_apos, uint32_t T, const uint32_t *d_ent_start, LzmaJob *d_jobs);
int lzma_launch(Ctx *c, const LzmaJob *d_jobs, const uint32_t *d_order, uint32_t count, const uint8_t *d_in, const uint32_t *d_tok, uint8_t *d_out, const Bt4Sets &sets, uint64_t *d_result,
                uint8_t *d_save = nullptr, uint64_t budget = 0, uint64_t pos_cap = ~0ull, int waves = 1);
uint64_t lzma_save_stride();
int lzma_save_info(const uint8_t *blob, uint64_t *pos, uint64_t *olen, uint64_t *n);
int lzma_save_fits(const uint8_t *blob, const LzmaJob &J);
int ensure_lz_workspace(Ctx *c, uint64_t nbuf);
int ensure_entropy_workspace(Ctx *c, uint64_t atoms, uint64_t flushes, uint64_t out_bytes = 0);   // out_bytes: the input the stream is made of (0: as many bytes as atoms)
int ensure_crc_workspace(Ctx *c, uint64_t n);

// One shard of a range through the LZ stage.  W.in holds `nbuf` bytes (zero pad behind): a 32 KiB halo in front of the
// shard unless it starts the stream, the shard, and a tail behind it unless it ends the stream.  The tokens of the parse
// chunks [tok_lo / PCHUNK, tok_hi / PCHUNK) are the shard's atoms (all chunks from tok_lo on when `final`).
struct ShardJob {
  uint64_t nbuf = 0;
  uint32_t tok_lo = 0, tok_hi = 0;
  bool final = false;                 // the buffer ends where the stream ends
  bool entry_known = true;            // else: warm-up parse from the start of the buffer
  ExitState entry{0, SYNC_F};         // buffer coordinates
  uint32_t *dst_atoms = nullptr, *dst_apos = nullptr;
  uint32_t apos_bias = 0;             // added to buffer positions: offset of the buffer in the range's input
  uint64_t cap_atoms = 0;             // room at dst
  // grow_atoms (total): the shard has `total` atoms and dst has room for fewer -- the caller makes room (keeping what the shards before have written)
  // and says where the shard's atoms go now; null, or a non-zero return: ZADA_E_NOMEM ("atom array overflow")
  std::function<int(uint64_t, uint32_t **, uint32_t **)> grow_atoms;
  const uint32_t *segend = nullptr;   // a batch of entries in the buffer (Layout); then tok_lo = 0, final, entry at 0
  // need (x): the first x bytes of W.in are to be valid before whatever is enqueued next on the context's stream runs (null: all of
  // them are).  The host-buffer entry point's input arrives piece by piece while the first kernel already works on what has come.
  std::function<int(uint64_t)> need;
};
struct ShardResult { uint32_t ntok = 0; ExitState exit{0, SYNC_F}, warm{0, SYNC_F}; };
int lz_shard(Ctx *c, int level, const ShardJob &job, ShardResult *res);

int ensure_batch_workspace(Ctx *c, uint64_t entries, uint64_t fslots, uint64_t segs);
int batch_geometry(Ctx *c, uint32_t E, const uint32_t *d_total_atoms);
int entropy_analyze(Ctx *c);
int entropy_choose(Ctx *c);
int entropy_emit(Ctx *c, uint8_t *d_out);
int crc_launch(Ctx *c, const uint8_t *d_in, uint64_t n);
int crc_finish(Ctx *c, uint64_t n, uint32_t *crc_inout);
// zada_sort.hip: stable radix sort of n (key, value) pairs by the key bits [begin_bit, end_bit); values of 4 or 16 bytes
size_t radix_sort_tmp_bytes(size_t n, size_t value_bytes);
int radix_sort_pairs(Ctx *c, hipStream_t st, void *tmp, size_t tmp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const void *vals_in, void *vals_out, size_t value_bytes,
                     size_t n, unsigned begin_bit, unsigned end_bit);
void exclusive_scan_u32(hipStream_t st, const uint32_t *d_in, uint32_t *d_out, uint32_t *d_sums, uint32_t *d_total, uint32_t n);

}  // namespace zada
