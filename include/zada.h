/*
 * zada.h -- C ABI of the MI355X-native Deflate encoder (libzada_hip.so).
 *
 * This is the drop-in boundary for the reference's private child procedure
 *
 *     procedure Zip.Compress.Deflate (input, output, input_size_known, input_size, feedback,
 *                                     method, CRC, crypto, output_size, compression_ok);
 *         -- zip_lib/zip-compress-deflate.ads:36-46, body zip-compress-deflate.adb:68-1679,
 *         -- sole caller zip_lib/zip-compress.adb:197-202
 *
 * The reference has no FFI seam (it is 100 % Ada); INTEGRATION.md shows the Ada body a
 * maintainer would substitute (Interfaces.C + pragma Import of the entry points below).
 * Plain pointers and sizes only; no global mutable state; a context is single-owner, many
 * contexts may be used concurrently (the reference is task-safe the same way, doc/zipada.txt:26).
 *
 * Everything computed behind these entry points runs in hand-written HIP kernels for gfx950.
 * There is NO CPU fallback: without a usable GPU zada_create() fails.
 */
#ifndef ZADA_H
#define ZADA_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Compression_Method'Pos (zip_lib/zip-compress.ads:59-122); Deflation_Method = 6 .. 11 (:132). */
enum {
  ZADA_DEFLATE_FIXED = 6,   /* single fixed block, LZ77 level IZ_4      (zip-compress-deflate.adb:1574) */
  ZADA_DEFLATE_0 = 7,       /* no LZ77, Taillaule + Huffman only        (:1575) */
  ZADA_DEFLATE_1 = 8,       /* IZ_6  (8,16,128,128),   1 scan level     (:1576, lz77.adb:542) */
  ZADA_DEFLATE_2 = 9,       /* IZ_8  (32,128,258,1024), 2 scan levels   (:1577, lz77.adb:544) */
  ZADA_DEFLATE_3 = 10,      /* IZ_10 (34,258,258,4096), 3 scan levels   (:1578, lz77.adb:546) */
  ZADA_DEFLATE_R = 11,      /* LZ77.Rich -- not implemented (out of scope) */
  ZADA_BZIP2_1 = 12,        /* BZip2, 100 000-byte blocks  (zip-compress-bzip2_e.adb:138-142, bzip2-encoding.adb:93-98) */
  ZADA_BZIP2_2 = 13,        /* BZip2, 400 000-byte blocks */
  ZADA_BZIP2_3 = 14,        /* BZip2, 900 000-byte blocks, four splitting tactics per block (bzip2-encoding.adb:1214-1345) */
  ZADA_LZMA_0 = 15,         /* LZMA, no LZ77: literals and short repeats only (lzma-encoding.adb:118-122) */
  ZADA_LZMA_1 = 16,         /* LZMA, Info-Zip matcher level 6, matches written as they come */
  ZADA_LZMA_2 = 17,         /* LZMA, Info-Zip matcher level 10, simple comparison of the ways to write a match */
  ZADA_LZMA_3 = 18          /* LZMA, BT4 matcher, dictionary = the entry's size (up to 256 MiB), all comparisons incl. splitting */
};

/* Return codes.  1 and 2 mirror the reference's two non-error outcomes:
 *   1  <=> compression_ok = False  (Compression_inefficient, zip-compress.adb:479-486,
 *          caught at zip-compress-deflate.adb:1667-1669; the caller then Stores the entry)
 *   2  <=> User_abort raised from the feedback callback (zip-compress-deflate.adb:1489-1491) */
enum {
  ZADA_OK = 0,
  ZADA_INEFFICIENT = 1,
  ZADA_ABORTED = 2,
  ZADA_E_INVALID = -1,      /* bad argument / unsupported method */
  ZADA_E_NOMEM = -2,        /* host or device allocation failed */
  ZADA_E_HIP = -3,          /* HIP runtime error; see zada_last_error() */
  ZADA_E_TOO_LARGE = -4,    /* the input is larger than the call takes: zada_range_open takes ranges below 4 GiB - 64 MiB (zada_deflate* take streams of any
                             * length, span after span), zada_lzma* entries below 2 GiB - 64 KiB and batches below 4 GiB; LZMA_3 also what the match producer's 156 bytes of
                             * device memory per input byte allow */
  ZADA_E_NO_DEVICE = -5,    /* no gfx950 device / HIP extension unusable */
  ZADA_E_REFERENCE = -6     /* LZMA_3 only: on this entry the reference's BT4 matcher reports a match that is none (lz77.adb:1262-1290 read behind pending
                             * bytes that no window fill took up, :1000-1017, 1397-1406: lzPos lags behind readPos) and the reference's own stream does not
                             * decode to the input.  Cannot happen with the dictionary Zip.Compress.LZMA_E asks for unless the entry is beyond 256 MiB and its
                             * last window fill brings 163 .. 4 368 bytes; the shim Stores such an entry or takes another method.  Nothing was written. */
};

typedef struct zada_ctx zada_ctx;

/* Feedback_Proc (zip_lib/zip.ads:301-305).  Return non-zero to request user_abort.
 * Called with a monotone 0..100 at kernel-phase granularity. */
typedef int (*zada_feedback_fn)(int percents_done, void *user);

/* Context = device + stream + workspace.  device >= 0 selects a HIP device. */
zada_ctx *zada_create(int device);
void zada_destroy(zada_ctx *ctx);
const char *zada_last_error(const zada_ctx *ctx);
const char *zada_version(void);
/* Tuning / test knobs of a context (also read from the environment when the context is created):
 * "budget" (ZADA_BUDGET: rounds of chain steps per position in the first match pass; 0 = unbounded, -1 = default),
 * "max_demand_rounds" (ZADA_MAX_DEMAND_ROUNDS), "exact_respec" (ZADA_EXACT_RESPEC: lists of up to this many flagged 512-byte chunks are parsed again by
 * one wave per chunk with the exact match search inside the parse, default 32768; 0 = never), "cd_filter" (ZADA_CD_FILTER: 1 = the cross-segment
 * continuation of the four-byte searches asks a Bloom filter of the previous segment first and leaves long walks to a second pass, 0 = one pass as in rounds 1-5),
 * "cd_list_cap" (test knob: entries of that pass's list of open walks, 0 = by size; a full list leaves the walks where they are),
 * "atoms_pct" (ZADA_ATOMS_PCT: the atom arrays of a stream start with room for this many atoms per 100 input bytes, default 50, and grow when the match finder
 * writes more -- one atom per byte is the worst case), "fix_stride" (test knob: token slots per 512-byte chunk the parse splice starts with, 0 = 128; a splice that
 * needs more gets the full 1152 and the parse starts again), "inner_budget" (ZADA_INNER_BUDGET), "shard_kib" (ZADA_SHARD_KIB: KiB of
 * a stream the match finder takes at a time, multiple of 64), "span_mib" (MiB of a stream one pass takes; longer streams go span after
 * span, default 2048), "link_run" (segments of 32 KiB one workgroup of the link stage takes one after the other, making their cross links itself:
 * a power of two from 1 to 64, 0 = by size; anything else is ZADA_E_INVALID), "batch_mib" (MiB one batch of small entries may take), "bz_batch_mib" / "bz_span_mib" / "bz_batch_melems" (BZip2
 * batching; "bz_lists", "bz_list_rows", "bz_text_order", "bz_pipeline", "bz_pipe_prio", "bz_small_wg", "bz_split", "bz_tail_pct": scheduling of the BZip2 stages), "lzma_chunk" (positions of an LZMA stream one launch codes between two feedback calls; 0 = by level, -1 = one launch
 * per stream), "lzma_pool" (test knob: blocks of the LZMA_3 match sets' overflow pool to start with, 0 = by size; a pool that is too small is
 * counted and the match producer's walk runs again; for a stream whose producer works in segments the pool grows between the segments), "lzma_pool_fixed" (test knob: 1 = it
 * does not), "lzma_segment" (one LZMA_3 stream coded in launches: log2 of the positions per segment of
 * the match producer, whose walks of segment k + 1 run beside the coder of segment k; 13 .. 30, 0 = by size: 2 ** 20 positions for streams from 2 MiB on, 2 ** 18 from 512 KiB on, none below,
 * -1 = all match sets before the coder starts), "lzma_waves" (one LZMA_3 stream alone: waves of its workgroup -- 0 or 4 = the wave that walks the stream's
 * chain and three helpers that take shares of its forks, 1 = that wave alone, as every entry of a batch has it).  None of them changes a byte.  One knob is a parameter of the reference instead: "lzma_dict" = LZMA.Encoding.Encode's
 * dictionary_size for LZMA_3 in bytes (0, the default: the entry's size, as Zip.Compress.LZMA_E passes it; lzma_enc.adb uses 32 KiB). */
int zada_set_knob(zada_ctx *ctx, const char *name, int value);

/* Zip.Compress.Deflate on host buffers (the Ada shim drains `input` with Zip.Block_Read into
 * `in`, and passes `out` through CRC_Crypto.Encode + Zip.Block_Write afterwards).
 *   crc_inout : the RUNNING CRC register ("only updated here": caller does Init before and
 *               Final after, zip-compress.adb:144, 218).  May be NULL.
 *   cap       : capacity of out; cap >= n + 64 always suffices.
 *   out_len   : output_size.
 * Output bytes are bit-exact with the CPU restatement of the reference encoder (oracle/) for the same method; parity with an
 * Ada build of the reference is unpinned (no GNAT in this image: DESIGN.md 2). */
int zada_deflate(zada_ctx *ctx, int method, const uint8_t *in, uint64_t n,
                 uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
                 zada_feedback_fn fb, void *user);

/* Same, with `d_in` / `d_out` already resident in device memory (HBM) of ctx's device.
 * d_in must be readable for n bytes; d_out must hold cap >= n + 64 bytes. */
int zada_deflate_device(zada_ctx *ctx, int method, const void *d_in, uint64_t n,
                        void *d_out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout);

/* `count` independent streams (e.g. one per Zip entry: Zip.Create.Add_Stream, zip-create.adb:194-297, is per
 * entry; zipada's usual workload is many small files, tools/zipada.adb:126-134).  Entries of up to 4 MiB go through ONE
 * launch sequence, up to 512 MiB of them at a time ("batch_mib"); larger ones one after the other.  The bytes are those of one zada_deflate call per entry.  rc[i] receives the per-stream return code
 * (0, 1 = inefficient: Store it, or < 0); crc[i] is in/out as above.  Returns the last negative rc[i], or 0. */
int zada_deflate_batch(zada_ctx *ctx, int method, int count,
                       const uint8_t *const *in, const uint64_t *n,
                       uint8_t *const *out, const uint64_t *cap,
                       uint64_t *out_len, uint32_t *crc, int *rc);

/* Zip.Compress.Compress_Data for one unencrypted Deflate method (zip-compress.adb:142-241):
 * CRC Init/Final around zada_deflate, Store fallback when compression_ok = False.
 * zip_type: 8 (deflate) or 0 (store).  crc_out is the final CRC-32. */
int zada_compress_data(zada_ctx *ctx, int method, const uint8_t *in, uint64_t n,
                       uint8_t *out, uint64_t cap, uint64_t *out_len,
                       uint32_t *crc_out, uint16_t *zip_type);

/* ---- One stream over several contexts (GPUs) -------------------------------------------------------------------
 * The reference compresses an entry as ONE sequential stream (a 32 KiB window, a lazy-match state machine, a flush of the
 * LZ buffer every 65 536 atoms and the block chooser's state all run through it: lz77.adb:827-933,
 * zip-compress-deflate.adb:993-997, 1424-1432).  A stream is cut into RANGES at multiples of 64 KiB, one per context; the
 * calls below run a range's stages and expose exactly the state the ranges have to exchange, so that the concatenated
 * output is bit for bit the stream of one zada_deflate call on the whole input (tests/test_ranges.py).  The exchange itself
 * (RCCL all_gather / send-recv in zip-ada_amd/sharding.py) is the caller's.  zada_deflate* use the same machinery inside
 * for one range, taking `shard_kib` KiB (knob, default 1 GiB) through the match finder at a time.
 *
 *   1. zada_range_open    d_in = device address of stream byte lo - pre; resident: pre = (lo ? 32768 : 0) bytes before the
 *                         range and post = min(1 MiB, stream_size - lo - n) behind it; lo, n multiples of 64 KiB (n free
 *                         for the last range); pre + n + post < 4 GiB - 64 MiB.
 *   2. zada_range_lz      match finding + lazy parse.  entry = the state the range before ended in (its info.exit), or
 *                         NULL if not known yet: the parse then starts 32 KiB earlier and info.warm is the first
 *                         history-free state at or beyond lo that it went through -- if it equals the neighbour's
 *                         info.exit the result stands, otherwise call zada_range_lz again with that exit.
 *   3. zada_range_edges   the range's first <= 65 536 and last <= 2 048 atoms, for its neighbours.
 *      zada_range_place   atoms of the stream before this range / in all, and the neighbours' atoms the range lacks:
 *                         n_lb = max(0, 2048 - (F - atoms_before)) behind, F = first multiple of 65 536 >= atoms_before,
 *                         (0 if the range owns no flush or F = 0) and enough ahead to complete its last flush.
 *   4. zada_range_analyze everything that does not depend on earlier blocks.
 *   5. zada_range_choose  the block decisions, from the 352-byte state of the range before (NULL: start of the stream);
 *                         bit_begin / bit_end: the range's bits in the stream.
 *   6. zada_range_emit    bytes [bit_begin / 8, ceil(bit_end / 8)) of the stream into d_out; a byte shared with a
 *                         neighbour holds only this range's bits (OR them together).
 * CRC-32: info.crc_raw is the register of the range's bytes started from 0; zada_crc32_combine(reg, raw, n) appends. */
typedef struct { uint64_t pos; uint32_t kind, pad; } zada_parse_state;     /* kind 1: fresh (lz77.adb:898-899), 2: literal pending */
typedef struct { uint64_t atoms; zada_parse_state exit, warm; uint32_t crc_raw, entry_known; } zada_range_info;
enum { ZADA_CARRY_BYTES = 352 };
int zada_range_open(zada_ctx *ctx, int method, const void *d_in, uint64_t stream_size, uint64_t lo, uint64_t n, uint64_t pre, uint64_t post);
int zada_range_lz(zada_ctx *ctx, const zada_parse_state *entry, zada_range_info *info);
int zada_range_edges(zada_ctx *ctx, void *d_head_atoms, void *d_head_pos, uint32_t *n_head, void *d_tail_atoms, void *d_tail_pos, uint32_t *n_tail);
int zada_range_place(zada_ctx *ctx, uint64_t atoms_before, uint64_t atoms_total, const void *d_lb_atoms, const void *d_lb_pos, uint32_t n_lb,
                     const void *d_la_atoms, const void *d_la_pos, uint32_t n_la);
int zada_range_analyze(zada_ctx *ctx);
int zada_range_choose(zada_ctx *ctx, const void *carry_in, void *carry_out, uint64_t *bit_begin, uint64_t *bit_end);
int zada_range_emit(zada_ctx *ctx, void *d_out, uint64_t cap, uint64_t *nbytes);
uint32_t zada_crc32_combine(uint32_t reg, uint32_t raw, uint64_t len);

/* ---- Introspection used by tests and bench.py (not part of the reference's interface) ---- */

/* LZ77 stage only (lz77.adb:460-943 semantics): token = byte, or 0x80000000|len<<16|dist. */
int zada_lz77_tokens(zada_ctx *ctx, int method, const uint8_t *in, uint64_t n,
                     uint32_t *tokens, uint64_t cap, uint64_t *ntok);

/* Block decisions of the last zada_deflate* call, as the reference's trace log would list
 * them (zip-compress-deflate.adb:1244-1266): rec[4*i+0..3] = first atom, atom count,
 * format (0 stored, 1 fixed, 2 dynamic, 3 dynamic RLE-tweaked, 4 recycled), bit cost. */
int zada_last_blocks(zada_ctx *ctx, uint64_t *rec, uint64_t cap_blocks, uint64_t *nblocks);

/* The similarity tests of the Taillaule splitter in the last zada_deflate* call / range, as the reference's trace log lists
 * them (zip-compress-deflate.adb:480-488, 1384-1390): rec[3*i+0..2] = atom (index in the stream) at which a sliding window
 * was compared with the reference descriptor, L1 distance of the tweaked code-length vectors, step level that cut there
 * (1, 2, 3; 0 = similar).  One record per test point: the reference tests up to three levels there, with the same distance. */
int zada_last_trace(zada_ctx *ctx, uint64_t *rec, uint64_t cap, uint64_t *count);

/* Per-phase device time of the last call, measured with HIP events on the context's stream.
 * names[i] are static strings; ms[i] milliseconds.  Returns the number of phases. */
int zada_last_timing(zada_ctx *ctx, const char **names, float *ms, int cap);

/* Deterministic synthetic corpus "silesia_mix_v1" (bench / tests): bytes [offset, offset+len). */
void zada_silesia_mix(uint64_t seed, unsigned class_mask, uint64_t offset, uint64_t len, uint8_t *dst);

/* ---------------------------------------------------------------------------------------------------------------
 * BZip2.  Replaces the body of Zip.Compress.BZip2_E (zip_lib/zip-compress-bzip2_e.ads, .adb:44-157),
 * i.e. BZip2.Encoding.Encode (zip_lib/bzip2-encoding.adb:87-1431) with size_hint = the input's size (zip-create.adb:256-257).
 * Same conventions as zada_deflate: method = Compression_Method'Pos (ZADA_BZIP2_1 .. _3), crc_inout = the running Zip CRC-32
 * register, return ZADA_OK / ZADA_INEFFICIENT (stream not smaller than the input: compression_ok := False) / ZADA_ABORTED / < 0.
 * The stream is the complete BZip2 stream ("BZh9" ... footer).  Unlike zada_deflate it is also delivered with
 * ZADA_INEFFICIENT when it fits `cap` (*out_len <= cap), so the entry points serve a stand-alone .bz2 writer (bzip2_enc.adb) too.
 * A stream that is SMALLER than the input but does not fit `cap` is an error (ZADA_E_INVALID, "output buffer too small"), not
 * ZADA_INEFFICIENT: cap >= n is always enough for the Zip use (a stream of n bytes or more is inefficient whatever cap is).
 * Streams of any length: the block limits are found a span of the stream at a time (knob "bz_span_mib", default 1024).
 * --------------------------------------------------------------------------------------------------------------- */
int zada_bzip2(zada_ctx *ctx, int method, const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *out_len,
               uint32_t *crc_inout, zada_feedback_fn fb, void *user);
/* the same with input and output in device memory (d_out: cap bytes) */
int zada_bzip2_device(zada_ctx *ctx, int method, const void *d_in, uint64_t n, void *d_out, uint64_t cap, uint64_t *out_len,
                      uint32_t *crc_inout);
/* Many entries in one call (zipada's usual workload: many small files): the entries that are one block each -- up to 0.8 block
 * capacities, 720 000 bytes for BZip2_3 -- go through ONE launch sequence (every entry is a block of the call with its own stream
 * header, tactic choice and footer); longer ones are taken one after the other.  Arrays as for zada_deflate_batch; rc[i] is
 * zada_bzip2's return code for entry i, the stream is delivered whenever it fits cap[i].  Knob "bz_batch_mib" (default 256):
 * MiB of entries per launch sequence.  Returns the worst rc. */
int zada_bzip2_batch(zada_ctx *ctx, int method, int count, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out,
                     const uint64_t *cap, uint64_t *out_len, uint32_t *crc, int *rc);
/* ---------------------------------------------------------------------------------------------------------------
 * LZMA.  Replaces the body of Zip.Compress.LZMA_E (zip_lib/zip-compress-lzma_e.ads, .adb:29-184) for the
 * methods LZMA_0 .. LZMA_3, i.e. LZMA.Encoding.Encode (zip_lib/lzma-encoding.adb:59-1563) with lc = 3, lp = 0, pb = 2, an end
 * marker and dictionary_size = the input's size (zip-compress-lzma_e.adb:121-126, 160-165).  Conventions as zada_deflate:
 * method = Compression_Method'Pos, crc_inout = the running Zip CRC-32 register, return ZADA_OK / ZADA_INEFFICIENT / < 0.
 * The output is the Zip payload: the four bytes 16, 2, 5, 0 (:155-158), the 5-byte LZMA header, the range-coded stream.
 * The coder of a stream is one chain of dependent steps (adaptive probabilities): one workgroup codes it; entries are what runs in
 * parallel -- use zada_lzma_batch for many of them.  LZMA_3's BT4 matcher (lz77.adb:953-1827) is NOT part of that chain: its match sets are
 * a function of the input alone and are produced by data-parallel kernels before the coder starts (about 156 bytes of device memory per
 * input byte of the call, kept by the context: an LZMA_3 entry or batch of more than free device memory / 156 -- about 1.6 GiB on a
 * 288 GB device with nothing else on it -- is refused with ZADA_E_TOO_LARGE and the limit in zada_last_error).  A stream runs as a sequence of bounded launches (about half a second each,
 * "lzma_chunk"), the coder's state waiting in device memory in between: fb (may be NULL) is called with 0, between the launches
 * and with 100, and a non-zero return ends the call with ZADA_ABORTED (Feedback / User_abort, zip-compress-lzma_e.adb:78-92).
 * LZMA_3 can also return ZADA_E_REFERENCE (see the enum: an entry on which the reference's own matcher leaves the format -- not with the dictionary
 * Zip.Compress.LZMA_E asks for unless the entry is beyond 256 MiB; per entry in zada_lzma_batch's rc array).
 * Limits: entries below 2 GiB - 64 KiB, and for LZMA_3 below what the producer's memory allows (see above) (ZADA_E_TOO_LARGE beyond: the shim
 * Stores such an entry or raises); only the
 * (lc, lp, pb) = (3, 0, 2) methods LZMA_0 .. LZMA_3, not the data-specific LZMA_for_* variants (ZADA_E_INVALID).
 * --------------------------------------------------------------------------------------------------------------- */
int zada_lzma(zada_ctx *ctx, int method, const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
              zada_feedback_fn fb, void *user);
/* the same with input and output in device memory (d_out: cap bytes) */
int zada_lzma_device(zada_ctx *ctx, int method, const void *d_in, uint64_t n, void *d_out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout);
/* A stream that stopped between two launches (its feedback returned non-zero: ZADA_ABORTED) can be taken up again -- by this context, another one or
 * another process: zada_lzma_export_state copies out the coder's state (*state_len bytes: the probability model, the range coder, the window
 * bookkeeping; state_cap must hold them -- call with state = NULL to learn the length) and the stream bytes written so far (*out_bytes of them into
 * `out`, NULL to skip; *positions = input positions coded); zada_lzma_import_state hands a state to a context, whose NEXT zada_lzma call -- same
 * input, same method -- goes on from there and returns the whole stream's length, with its output buffer valid from byte *out_bytes of the export on
 * (the bytes before are the export's).  The match sets of LZMA_3 are a function of the input alone: the resumed call makes them again.  This is
 * Feedback / User_abort (zip-compress-lzma_e.adb:78-92) turned into a checkpoint: a stream that takes longer than one call may run is coded in two.
 * A state is spent by the next zada_lzma / zada_lzma_device call whatever comes of it, and it is checked before the coder takes it: a blob that is not
 * a stopped stream's (length, phase) is refused by zada_lzma_import_state, one whose stream length, method level, dictionary or counters do not fit the
 * call it meets by that call (ZADA_E_INVALID both times).  What cannot be checked is the input's CONTENT: the same bytes are the caller's to hand over. */
int zada_lzma_export_state(zada_ctx *ctx, uint8_t *state, uint64_t state_cap, uint64_t *state_len, uint8_t *out, uint64_t out_cap, uint64_t *out_bytes,
                           uint64_t *positions);
int zada_lzma_import_state(zada_ctx *ctx, const uint8_t *state, uint64_t state_len);
/* Many entries, one launch of the coder for all of them.  Arrays as for zada_deflate_batch; rc[i] is zada_lzma's return code
 * for entry i.  Returns the worst rc. */
int zada_lzma_batch(zada_ctx *ctx, int method, int count, const uint8_t *const *in, const uint64_t *n, uint8_t *const *out,
                    const uint64_t *cap, uint64_t *out_len, uint32_t *crc, int *rc);
/* Trace of the last zada_bzip2* call: per block of Read_and_Split_Block (bzip2-encoding.adb:1144) four values -- raw start,
 * raw length, splitting tactic kept (0 single, 1 parts_4, 2 segmented_1, 3 segmented_2), its number of sub-blocks.
 * Returns the number of values there are; at most cap_items are stored. */
uint64_t zada_bz2_last_blocks(zada_ctx *ctx, uint64_t *dst, uint64_t cap_items);
/* One BZip2 stream over several contexts / GPUs (BASELINE config 5: blocks sharded over the GPUs of a node).  Blocks are
 * independent once their limits are known (bzip2-encoding.adb:1161-1209) and once the bit phase in front of them is known
 * (:1312-1318), so a context takes the blocks that START inside its range of the stream:
 *   zada_bz2_range_open     d_buf holds the stream bytes [buf_off, buf_off + buf_len): the range and, behind it, enough of the
 *                           next ranges for the last block (a block takes at most ten capacities: 9 000 000 bytes + 259).  `start`
 *                           = where the range's first block starts (0 for the first range, the previous range's *next_start
 *                           otherwise: the one sequential hand-over, 8 bytes), own_end = where the range ends.  Fast (block limits only).
 *   zada_bz2_range_encode   every piece of every tactic of those blocks through Encode_Block (the work).
 *   zada_bz2_range_table    per block 12 values -- per tactic: bits, pieces, the pieces' CRCs folded from zero.  All ranks'
 *                           tables, in stream order, go to
 *   zada_bz2_select         (pure host arithmetic, any rank): tactic per block, stream bit position and combined CRC behind
 *                           the blocks, from the position / CRC in front of them (32 and 0 at the stream's start).
 *   zada_bz2_range_assemble the range's bytes of the stream from byte bit_begin / 8 on; flags: 1 = stream header in front
 *                           (bit_begin must be 32), 2 = footer with footer_crc behind.  Neighbours share a byte when a
 *                           range does not end on a byte: the gatherer ORs (as for the Deflate ranges). */
/* Raw CRC-32 register (started from 0) of n bytes in device memory (16-byte aligned): a rank's piece of the stream's Zip CRC-32;
 * zada_crc32_combine chains the pieces in stream order. */
int zada_crc32_device(zada_ctx *ctx, const void *d_in, uint64_t n, uint32_t *raw);
int zada_bz2_range_open(zada_ctx *ctx, int method, const void *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t stream_total,
                        uint64_t start, uint64_t own_end, uint64_t *next_start, uint64_t *nblocks);
int zada_bz2_range_encode(zada_ctx *ctx);
uint64_t zada_bz2_range_table(zada_ctx *ctx, uint64_t *tab, uint64_t cap_blocks);
void zada_bz2_select(uint64_t nblk, const uint64_t *tab, uint64_t bitpos_in, uint32_t crc_in, uint8_t *choice, uint64_t *bitpos_out, uint32_t *crc_out);
int zada_bz2_range_assemble(zada_ctx *ctx, const uint8_t *choice, uint64_t nblk, uint64_t bit_begin, int flags, uint32_t footer_crc,
                            void *d_out, uint64_t cap, uint64_t *nbytes);
/* Test hook: the match sets LZMA_3's BT4 matcher (lz77.adb:1234-1361, BT4_Algo.Read_One_and_Get_Matches) finds at every position of
 * ONE entry, as the producer kernels leave them for the coder (zip-ada_amd/csrc/zada_bt4.hip): cnt [n] matches per position, lengths and
 * distances in len / dist [n * stride], stride >= 50.  Dictionary = the entry's size, or the "lzma_dict" knob. */
int zada_lzma_match_sets(zada_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *cnt, uint16_t *len, uint32_t *dist, int stride);
/* Test hooks: sub-blocks (Encode_Block jobs) of a host buffer through the stages, and the tables they leave. */
int zada_bz2_run(zada_ctx *ctx, const uint8_t *in, uint64_t n, uint32_t nsb, const uint64_t *starts, const uint32_t *lens, int option, int stages);
int zada_bz2_fetch(zada_ctx *ctx, const char *name, void *dst, uint64_t cap, uint64_t *nbytes);

#ifdef __cplusplus
}
#endif
#endif
