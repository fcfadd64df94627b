/*
 * zlib_pin.c -- CPU ORACLE SUPPORT (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Independent pin for the LZ77 stage of the oracle.  The reference's Info-Zip match
 * finder (lz77.adb:460-943) descends from the same deflate.c as zlib's deflate_slow;
 * with windowBits 15, memLevel 8 (=> 15 hash bits) and deflateTune(good, lazy, nice,
 * chain) set to a row of lz77.adb:534-546, libz 1.2.11 makes the same LZ77 decisions
 * (SURVEY.md section 0, fact 5).  This file runs libz, then recovers the LZ77 token
 * stream from its output with a small inflater of its own (RFC 1951), so that tests
 * can compare it, position by position, with zo_lz77_tokens().
 *
 * zlib may choose *stored* blocks, inside which its tokens are not observable; those
 * byte ranges are reported as ZP_UNKNOWN.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define ZP_MATCH 0x80000000u
#define ZP_NONE 0xFFFFFFFEu     /* no token starts at this position (inside a match) */
#define ZP_UNKNOWN 0xFFFFFFFFu  /* position covered by a stored block */

typedef struct { const uint8_t *p; uint64_t n, pos; uint32_t bitbuf; int bitcnt; int err; } bitrd;

static uint32_t getbits(bitrd *b, int need) {
  uint32_t v;
  while (b->bitcnt < need) {
    if (b->pos >= b->n) { b->err = 1; return 0; }
    b->bitbuf |= (uint32_t)b->p[b->pos++] << b->bitcnt;
    b->bitcnt += 8;
  }
  v = b->bitbuf & ((need == 32) ? 0xFFFFFFFFu : ((1u << need) - 1));
  b->bitbuf >>= need; b->bitcnt -= need;
  return v;
}

typedef struct { short count[16]; short symbol[288]; } huff;

static int construct(huff *h, const short *length, int n) {
  short offs[16];
  int left = 1;
  memset(h->count, 0, sizeof h->count);
  for (int s = 0; s < n; s++) h->count[length[s]]++;
  if (h->count[0] == n) return 0;
  for (int len = 1; len <= 15; len++) { left <<= 1; left -= h->count[len]; if (left < 0) return left; }
  offs[1] = 0;
  for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len];
  for (int s = 0; s < n; s++) if (length[s] != 0) h->symbol[offs[length[s]]++] = (short)s;
  return left;
}

static int decode(bitrd *b, const huff *h) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)getbits(b, 1);
    if (b->err) return -1;
    int count = h->count[len];
    if (code - count < first) return h->symbol[index + (code - first)];
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return -1;
}

static const short lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const short lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const short dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const short dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

/* Recover per-position tokens from a raw deflate stream. postok[n]. Returns 0 or <0. */
static int tokens_of_stream(const uint8_t *z, uint64_t zn, uint32_t *postok, const uint8_t *orig, uint64_t n) {
  bitrd b = {z, zn, 0, 0, 0, 0};
  uint64_t pos = 0;
  int last;
  do {
    last = (int)getbits(&b, 1);
    int type = (int)getbits(&b, 2);
    if (b.err) return -1;
    if (type == 0) {
      b.bitbuf = 0; b.bitcnt = 0;
      if (b.pos + 4 > b.n) return -2;
      uint32_t len = b.p[b.pos] | (b.p[b.pos + 1] << 8);
      b.pos += 4;
      if (b.pos + len > b.n || pos + len > n) return -3;
      for (uint32_t i = 0; i < len; i++) postok[pos + i] = ZP_UNKNOWN;
      pos += len; b.pos += len;
    } else if (type == 1 || type == 2) {
      huff lencode, distcode;
      short lengths[320];
      if (type == 1) {
        int s = 0;
        for (; s < 144; s++) lengths[s] = 8;
        for (; s < 256; s++) lengths[s] = 9;
        for (; s < 280; s++) lengths[s] = 7;
        for (; s < 288; s++) lengths[s] = 8;
        construct(&lencode, lengths, 288);
        for (s = 0; s < 30; s++) lengths[s] = 5;
        construct(&distcode, lengths, 30);
      } else {
        static const short order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
        int nlen = (int)getbits(&b, 5) + 257, ndist = (int)getbits(&b, 5) + 1, ncode = (int)getbits(&b, 4) + 4;
        int index;
        for (index = 0; index < ncode; index++) lengths[order[index]] = (short)getbits(&b, 3);
        for (; index < 19; index++) lengths[order[index]] = 0;
        if (construct(&lencode, lengths, 19) != 0) return -4;
        index = 0;
        while (index < nlen + ndist) {
          int sym = decode(&b, &lencode);
          if (sym < 0) return -5;
          if (sym < 16) lengths[index++] = (short)sym;
          else {
            int len = 0, rep;
            if (sym == 16) { if (index == 0) return -6; len = lengths[index - 1]; rep = 3 + (int)getbits(&b, 2); }
            else if (sym == 17) rep = 3 + (int)getbits(&b, 3);
            else rep = 11 + (int)getbits(&b, 7);
            if (index + rep > nlen + ndist) return -7;
            while (rep--) lengths[index++] = (short)len;
          }
        }
        construct(&lencode, lengths, nlen);
        construct(&distcode, lengths + nlen, ndist);
      }
      for (;;) {
        int sym = decode(&b, &lencode);
        if (sym < 0 || b.err) return -8;
        if (sym < 256) {
          if (pos >= n || orig[pos] != (uint8_t)sym) return -9;
          postok[pos++] = (uint32_t)sym;
        } else if (sym == 256) break;
        else {
          sym -= 257;
          if (sym >= 29) return -10;
          int len = lbase[sym] + (int)getbits(&b, lext[sym]);
          int ds = decode(&b, &distcode);
          if (ds < 0) return -11;
          int dist = dbase[ds] + (int)getbits(&b, dext[ds]);
          if ((uint64_t)dist > pos || pos + (uint64_t)len > n) return -12;
          postok[pos] = ZP_MATCH | ((uint32_t)len << 16) | (uint32_t)dist;
          for (int i = 1; i < len; i++) postok[pos + i] = ZP_NONE;
          pos += (uint64_t)len;
        }
      }
    } else return -13;
  } while (!last);
  return pos == n ? 0 : -14;
}

/* Run libz (raw deflate, level 9, memLevel 8, default strategy) tuned to (good, lazy, nice,
 * chain); fill postok[n] with the token that starts at each input position. */
int zp_zlib_position_tokens(const uint8_t *in, uint64_t n, int good, int lazy, int nice, int chain, uint32_t *postok) {
  z_stream s;
  uint64_t cap = n + n / 8 + 1024;
  uint8_t *z = (uint8_t *)malloc(cap);
  int rc;
  if (!z) return -100;
  memset(&s, 0, sizeof s);
  if (deflateInit2(&s, 9, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { free(z); return -101; }
  if (deflateTune(&s, good, lazy, nice, chain) != Z_OK) { deflateEnd(&s); free(z); return -102; }
  s.next_in = (Bytef *)in; s.avail_in = (uInt)n;
  s.next_out = z; s.avail_out = (uInt)cap;
  rc = deflate(&s, Z_FINISH);
  if (rc != Z_STREAM_END) { deflateEnd(&s); free(z); return -103; }
  uint64_t zn = s.total_out;
  deflateEnd(&s);
  rc = tokens_of_stream(z, zn, postok, in, n);
  free(z);
  return rc;
}

/* Same position-indexed view of a raw deflate stream made by anybody (used to inspect
 * the oracle's / the GPU's own output). */
int zp_stream_position_tokens(const uint8_t *z, uint64_t zn, const uint8_t *orig, uint64_t n, uint32_t *postok) {
  return tokens_of_stream(z, zn, postok, orig, n);
}

/* libz alone, tuned to (good, lazy, nice, chain): compressed size, or < 0.  The secondary CPU anchor of bench.py
 * (deflateTune(34, 258, 258, 4096) = the reference's IZ_10 row, lz77.adb:546). */
int64_t zp_zlib_tuned_size(const uint8_t *in, uint64_t n, int good, int lazy, int nice, int chain) {
  z_stream s;
  uint64_t cap = n + n / 8 + 1024;
  uint8_t *z = (uint8_t *)malloc(cap);
  if (!z) return -100;
  memset(&s, 0, sizeof s);
  if (deflateInit2(&s, 9, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { free(z); return -101; }
  if (deflateTune(&s, good, lazy, nice, chain) != Z_OK) { deflateEnd(&s); free(z); return -102; }
  s.next_in = (Bytef *)in; s.avail_in = (uInt)n;
  s.next_out = z; s.avail_out = (uInt)cap;
  if (deflate(&s, Z_FINISH) != Z_STREAM_END) { deflateEnd(&s); free(z); return -103; }
  int64_t zn = (int64_t)s.total_out;
  deflateEnd(&s);
  free(z);
  return zn;
}
