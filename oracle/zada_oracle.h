/*
 * zada_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A single-threaded, plain-C restatement of the reference's Deflate encoder path
 * (zertovitch/zip-ada, lib version "62"):
 *     zip_lib/zip-compress-deflate.adb            (whole file)
 *     zip_lib/lz77.adb:460-943, 2181-2198         (Info-Zip match finder + dispatch)
 *     zip_lib/huffman-encoding.adb:34-80          (canonical codes)
 *     zip_lib/huffman-encoding-length_limited_coding.adb:46-280  (boundary package-merge)
 *     zip_lib/zip-compress.adb:48-62, 142-241, 430-490 (buffers, store fallback, inefficiency rule)
 *     zip_lib/zip-crc_crypto.adb:31-76            (CRC-32)
 *     zip_lib/zip-create.adb / zip-headers.adb    (container bytes; see zada_oracle_zip.c)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * The product (zip-ada_amd/) never links, imports or executes anything in oracle/.
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - The reference is pure Ada; no Ada compiler exists in the build image, so the
 *     reference binary cannot be run here.  Byte-level parity against the Ada binary
 *     is therefore UNPINNED ("parity unpinned").
 *   - What IS pinned: the LZ77 stage against zlib 1.2.11 deflateTune() token streams
 *     (oracle/zlib_pin.c), the entropy stage by round trip through zlib/zipfile/unzip,
 *     the length-limited code lengths against an independent optimal-cost DP, and the
 *     reference's own test input vectors (test/test_llhc.adb) as self-pinned fixtures.
 */
#ifndef ZADA_ORACLE_H
#define ZADA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Compression_Method'Pos values -- zip_lib/zip-compress.ads:59-122 */
enum {
  ZO_STORE = 0,
  ZO_DEFLATE_FIXED = 6,
  ZO_DEFLATE_0 = 7,
  ZO_DEFLATE_1 = 8,
  ZO_DEFLATE_2 = 9,
  ZO_DEFLATE_3 = 10,
  ZO_DEFLATE_R = 11,
  ZO_BZIP2_1 = 12,
  ZO_BZIP2_2 = 13,
  ZO_BZIP2_3 = 14,
  ZO_LZMA_0 = 15,
  ZO_LZMA_1 = 16,
  ZO_LZMA_2 = 17,
  ZO_LZMA_3 = 18
};

/* Return codes of zo_deflate */
enum { ZO_OK = 0, ZO_INEFFICIENT = 1, ZO_ABORTED = 2, ZO_EINVAL = -1, ZO_ENOMEM = -2 };

/* Token encoding used by the token-level entry points (tests only):
 *   literal : the byte value 0..255
 *   match   : 0x80000000 | (length << 16) | distance      (3..258, 1..32768) */
#define ZO_TOKEN_MATCH 0x80000000u

/* Feedback_Proc (zip.ads:301-305): percents_done, entry_skipped; returns user_abort */
typedef int (*zo_feedback_fn)(int percents_done, int entry_skipped, void *user);

/* Trace events (the reference's compile-time trace, zip-compress-deflate.adb:83-90) */
enum { ZO_TR_CUT = 1, ZO_TR_BLOCK = 2, ZO_TR_SIMILAR = 3, ZO_TR_BITPOS = 4 /* test hook: stream bit position before a block */ };
typedef void (*zo_trace_fn)(void *user, int kind, int64_t a, int64_t b, int64_t c, int64_t d);

/* CRC-32, zip-crc_crypto.adb:49-76.  Init = 0xFFFFFFFF, Final = NOT. */
uint32_t zo_crc32_init(void);
uint32_t zo_crc32_update(uint32_t crc, const uint8_t *buf, uint64_t n);
uint32_t zo_crc32_final(uint32_t crc);

/* Huffman.Encoding.Length_Limited_Coding.  freq[n], lengths[n] (out). Returns 0 or <0. */
int zo_llhc(const uint64_t *freq, int n, int max_bits, int *lengths);

/* Huffman.Encoding.Prepare_Codes with invert_bit_order = True. lengths[n] in, codes[n] out. */
void zo_prepare_codes(const int *lengths, int n, int max_huffman_bits, int invert, int32_t *codes);

/* LZ77.Encode, methods IZ_4 .. IZ_10 (level 4..10) and No_LZ77 (level 0).
 * Tokens are appended to tokens[0..cap); returns the number of tokens produced
 * (may exceed cap: then only the first cap were stored). */
uint64_t zo_lz77_tokens(const uint8_t *in, uint64_t n, int level, uint32_t *tokens, uint64_t cap);

/* Zip.Compress.Deflate (zip-compress-deflate.adb:68-78).
 *   crc_inout : running CRC register (caller does Init / Final), may be NULL.
 *   out/cap   : receives the raw RFC 1951 stream; cap >= n + 16 is always enough.
 * Returns ZO_OK, ZO_INEFFICIENT (compression_ok = False), ZO_ABORTED, or <0. */
int zo_deflate(const uint8_t *in, uint64_t n, int method,
               uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
               zo_feedback_fn fb, void *fb_user, zo_trace_fn tr, void *tr_user);

/* Entropy stage alone: feed an explicit token stream (as from zo_lz77_tokens)
 * through the Taillaule splitter / Huffman / bit emission of `method`.  `in` is the
 * original data (needed for the stored-block expansion). */
int zo_deflate_from_tokens(const uint8_t *in, uint64_t n, const uint32_t *tokens, uint64_t ntok,
                           int method, uint8_t *out, uint64_t cap, uint64_t *out_len,
                           zo_trace_fn tr, void *tr_user);

/* Zip.Compress.Compress_Data for a single (unencrypted) method, incl. the
 * Store fallback (zip-compress.adb:142-241).  zip_type: 8 = deflate, 0 = store.
 * crc_out is the FINAL CRC. */
int zo_compress_data(const uint8_t *in, uint64_t n, int method,
                     uint8_t *out, uint64_t cap, uint64_t *out_len,
                     uint32_t *crc_out, uint16_t *zip_type);

/* Archive writer either side of the hot path (zada_oracle_zip.c): Zip.Create.Create_Archive /
 * Add_Stream / Finish on a memory stream, incl. the Zip_64 promotion.  zo_zip_add_compressed / zo_zip_set_bias are
 * test hooks: an entry whose payload was made elsewhere (with any recorded uncompressed size), and a pretended number of
 * bytes in front of the buffer, so that the Zip_64 paths can be compared without 4 GiB of data. */
typedef struct zoz_archive zoz_archive;
zoz_archive *zo_zip_create(int method);
int zo_zip_add(zoz_archive *a, const char *entry_name, const uint8_t *data, uint64_t n,
               uint32_t file_time, int unicode_name);
int zo_zip_add_compressed(zoz_archive *a, const char *entry_name, const uint8_t *payload, uint64_t payload_len, uint32_t crc,
                          uint64_t uncompressed_size, int zip_type, uint32_t file_time, int unicode_name);
void zo_zip_set_bias(zoz_archive *a, uint64_t bias);
int zo_zip_finish(zoz_archive *a, const uint8_t **bytes, uint64_t *len);
void zo_zip_free(zoz_archive *a);

/* ---- BZip2 (zada_oracle_bz2.c) ---- */
/* BZip2.Encoding.Encode (bzip2-encoding.adb:87): option 0/1/2 = block_100k/400k/900k, size_hint = -1 when unknown.
 * The trace callback is called once per Read_and_Split_Block with the raw range and the splitting tactic kept
 * (0 single, 1 parts_4, 2 segmented_1, 3 segmented_2) and its number of sub-blocks. */
typedef void (*zo_bz2_trace_fn)(void *user, int64_t raw_start, int64_t raw_len, int tactic, int sub_blocks);
int zo_bzip2_encode(const uint8_t *in, uint64_t n, int option, int64_t size_hint, uint8_t *out, uint64_t cap, uint64_t *out_len,
                    zo_bz2_trace_fn tr, void *tr_user);
/* Zip.Compress.BZip2_E (zip-compress-bzip2_e.adb): method ZO_BZIP2_1..3; crc_inout = running Zip CRC register. */
int zo_bzip2(const uint8_t *in, uint64_t n, int method, uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout);
/* BZip2.CRC over a buffer (Init, Update, Final). */
uint32_t zo_bz2_crc(const uint8_t *buf, uint64_t n);
/* Stages of one Encode_Block (test hooks). */
typedef struct {
  int32_t rle_n, bwt_index, mtf_n, selector_count;
  int32_t coders, max_code_len, sample_width, alphabet;
  uint32_t block_crc, pad;
  uint64_t bits;            /* size of the block in the stream */
} zo_bz2_block_info;
int zo_bz2_block(const uint8_t *raw, int32_t n, int option, uint8_t *rle_out, uint8_t *bwt_out, uint16_t *mtf_out,
                 uint8_t *selectors_out, uint8_t *lens_out, zo_bz2_block_info *info, uint8_t *bits_out, uint64_t bits_cap);
int32_t zo_bz2_segments(const uint8_t *buf, int32_t len, int tactic, int32_t *seg, int32_t cap);

/* ---- LZMA (row f4): zada_oracle_lzma.c ---- */

/* LZMA.Encoding.Encode (lzma-encoding.adb:59-1563): level 0..3 = Level_0 .. Level_3, uncompressed_size_info = False.
 * Output: the 5-byte LZMA header, then the range-coded stream.  stats8 (may be NULL) receives counters of the choices taken:
 * [0] literal then DL (probable literal), [1] literal then DL (estimate), [2] DL then literal, [3] DL expanded, [4] DL split,
 * [5] short repeat matches, [6] repeat matches, [7] simple matches. */
int zo_lzma_encode(const uint8_t *in, uint64_t n, int level, int lc, int lp, int pb, int end_marker, int64_t dictionary_size,
                   uint8_t *out, uint64_t cap, uint64_t *out_len, uint64_t *stats8);

/* Zip.Compress.LZMA_E (zip-compress-lzma_e.adb:121-172) for methods ZO_LZMA_0 .. ZO_LZMA_3: the Zip payload (4-byte prefix,
 * LZMA header, stream with end marker).  Return codes as zo_deflate. */
int zo_lzma(const uint8_t *in, uint64_t n, int method, uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout);

/* Stage export: the match sets of BT4_Algo.Read_One_and_Get_Matches (lz77.adb:1234-1361) at every position of the input, the window
 * filled as LZ77_using_BT4 fills it; cnt [n], len / dist [n * stride] (stride >= 50: two hash matches + Depth_Limit tree matches). */
int zo_bt4_match_sets(const uint8_t *in, uint64_t n, int64_t dictionary_size, uint8_t *cnt, uint16_t *len, uint32_t *dist, int stride);

#ifdef __cplusplus
}
#endif
#endif
