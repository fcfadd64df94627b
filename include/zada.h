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
  ZADA_DEFLATE_R = 11       /* LZ77.Rich -- not implemented (out of scope) */
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
  ZADA_E_TOO_LARGE = -4,    /* single stream >= 2 GiB - 64 KiB per call in this version */
  ZADA_E_NO_DEVICE = -5     /* no gfx950 device / HIP extension unusable */
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
 * "max_demand_rounds" (ZADA_MAX_DEMAND_ROUNDS), "batch_streams" (ZADA_BATCH_STREAMS).  None of them changes a byte. */
int zada_set_knob(zada_ctx *ctx, const char *name, int value);

/* Zip.Compress.Deflate on host buffers (the Ada shim drains `input` with Zip.Block_Read into
 * `in`, and passes `out` through CRC_Crypto.Encode + Zip.Block_Write afterwards).
 *   crc_inout : the RUNNING CRC register ("only updated here": caller does Init before and
 *               Final after, zip-compress.adb:144, 218).  May be NULL.
 *   cap       : capacity of out; cap >= n + 64 always suffices.
 *   out_len   : output_size.
 * Output bytes are bit-exact with the reference encoder's for the same method. */
int zada_deflate(zada_ctx *ctx, int method, const uint8_t *in, uint64_t n,
                 uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout,
                 zada_feedback_fn fb, void *user);

/* Same, with `d_in` / `d_out` already resident in device memory (HBM) of ctx's device.
 * d_in must be readable for n bytes; d_out must hold cap >= n + 64 bytes. */
int zada_deflate_device(zada_ctx *ctx, int method, const void *d_in, uint64_t n,
                        void *d_out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout);

/* `count` independent streams (e.g. one per Zip entry: Zip.Create.Add_Stream, zip-create.adb:194-297, is per
 * entry).  Small entries are compressed several at a time (host threads with a stream and a workspace each,
 * owned by ctx; ZADA_BATCH_STREAMS, default 4), entries above 64 MiB one after the other.  rc[i] receives the
 * per-stream return code; crc[i] is in/out as above.  Returns the last negative rc[i], or 0. */
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

/* ---- Introspection used by tests and bench.py (not part of the reference's interface) ---- */

/* LZ77 stage only (lz77.adb:460-943 semantics): token = byte, or 0x80000000|len<<16|dist. */
int zada_lz77_tokens(zada_ctx *ctx, int method, const uint8_t *in, uint64_t n,
                     uint32_t *tokens, uint64_t cap, uint64_t *ntok);

/* Block decisions of the last zada_deflate* call, as the reference's trace log would list
 * them (zip-compress-deflate.adb:1244-1266): rec[4*i+0..3] = first atom, atom count,
 * format (0 stored, 1 fixed, 2 dynamic, 3 dynamic RLE-tweaked, 4 recycled), bit cost. */
int zada_last_blocks(zada_ctx *ctx, uint64_t *rec, uint64_t cap_blocks, uint64_t *nblocks);

/* Per-phase device time of the last call, measured with HIP events on the context's stream.
 * names[i] are static strings; ms[i] milliseconds.  Returns the number of phases. */
int zada_last_timing(zada_ctx *ctx, const char **names, float *ms, int cap);

/* Deterministic synthetic corpus "silesia_mix_v1" (bench / tests): bytes [offset, offset+len). */
void zada_silesia_mix(uint64_t seed, unsigned class_mask, uint64_t offset, uint64_t len, uint8_t *dst);

#ifdef __cplusplus
}
#endif
#endif
