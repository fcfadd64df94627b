/*
 * zada_oracle_zip.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Restatement of the archive writer either side of the Deflate hot path:
 *   Zip.Create.Add_Stream   zip_lib/zip-create.adb:194-297   (local header, Compress_Data, header rewrite)
 *   Zip.Create.Finish       zip_lib/zip-create.adb:645-756   (central directory, end-of-central-dir)
 *   Zip.Headers.Write       zip_lib/zip-headers.adb:168-195 (PK\1\2), 244-276 (PK\3\4), 494-511 (PK\5\6)
 * Zip_32 archives only (every size and offset below 0xFFFFFFFF, fewer than 65535 entries);
 * the Zip_64 promotion (zip-create.adb:161-179, 682-752) returns ZO_EINVAL here.
 */
#include "zada_oracle.h"
#include <stdlib.h>
#include <string.h>

#define ZOZ_MAX_ENTRIES 65534

typedef struct {
  /* Central_File_Header / Local_File_Header fields, zip-headers.ads */
  uint16_t made_by_version, needed_extract_version, bit_flag, zip_type;
  uint32_t file_timedate, crc_32;
  uint64_t compressed_size, uncompressed_size, local_header_offset;
  uint16_t filename_length;
  uint32_t external_attributes;
  char *name;
} zoz_entry;

typedef struct zoz_archive {
  uint8_t *buf; uint64_t cap, len;
  int method;
  zoz_entry *e; int n;
} zoz_archive;

static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { put16(p, v & 0xFFFF); put16(p + 2, v >> 16); }

static int grow(zoz_archive *a, uint64_t need) {
  if (a->len + need <= a->cap) return 0;
  uint64_t nc = a->cap ? a->cap * 2 : 1 << 16;
  while (nc < a->len + need) nc *= 2;
  uint8_t *nb = (uint8_t *)realloc(a->buf, nc);
  if (!nb) return ZO_ENOMEM;
  a->buf = nb; a->cap = nc;
  return 0;
}

/* Zip.Create.Create_Archive, zip-create.adb:36-58 (memory stream) */
zoz_archive *zo_zip_create(int method) {
  zoz_archive *a = (zoz_archive *)calloc(1, sizeof *a);
  if (a) a->method = method;
  return a;
}

/* Zip.Headers.Write (local header), zip-headers.adb:244-276, policy force_empty */
static void write_local(uint8_t *lhb, const zoz_entry *h) {
  lhb[0] = 'P'; lhb[1] = 'K'; lhb[2] = 3; lhb[3] = 4;
  put16(lhb + 4, h->needed_extract_version);
  put16(lhb + 6, h->bit_flag);
  put16(lhb + 8, h->zip_type);
  put32(lhb + 10, h->file_timedate);
  put32(lhb + 14, h->crc_32);
  put32(lhb + 18, (uint32_t)h->compressed_size);
  put32(lhb + 22, (uint32_t)h->uncompressed_size);
  put16(lhb + 26, h->filename_length);
  put16(lhb + 28, 0);
}

/* Zip.Create.Add_Stream, zip-create.adb:194-297.  `unicode_name` mirrors
 * Zip_Streams.Is_Unicode_Name (tools/zipada.adb:131 always sets it);
 * file_time = Zip_Streams.Time, default 16789*65536 (zip_streams.ads:223). */
int zo_zip_add(zoz_archive *a, const char *entry_name, const uint8_t *data, uint64_t n,
               uint32_t file_time, int unicode_name) {
  zoz_entry *h;
  uint64_t mem1, out_len = 0;
  uint32_t crc = 0; uint16_t zt = 0;
  size_t nl = strlen(entry_name);
  int rc;
  if (a->n >= ZOZ_MAX_ENTRIES || n >= 0xFFFFFFFFull - (1u << 17) || nl > 0xFFFF) return ZO_EINVAL;
  a->e = (zoz_entry *)realloc(a->e, (size_t)(a->n + 1) * sizeof(zoz_entry));
  h = &a->e[a->n];
  memset(h, 0, sizeof *h);
  /* Add_catalogue_entry :103-134 */
  h->made_by_version = 23; h->needed_extract_version = 10; h->bit_flag = 0;
  if (unicode_name) h->bit_flag |= 0x0800;                       /* Language_Encoding_Flag_Bit */
  h->name = (char *)malloc(nl + 1);
  memcpy(h->name, entry_name, nl + 1);
  for (size_t i = 0; i < nl; i++) if (h->name[i] == '\\') h->name[i] = '/';   /* Unixify :181-192 */
  h->file_timedate = file_time;
  h->uncompressed_size = n; h->compressed_size = n;
  h->filename_length = (uint16_t)nl;
  mem1 = a->len;
  h->local_header_offset = mem1;
  if ((rc = grow(a, 30 + nl + n + 64)) != 0) return rc;
  write_local(a->buf + a->len, h);                                /* provisional header :243 */
  a->len += 30;
  memcpy(a->buf + a->len, h->name, nl); a->len += nl;
  rc = zo_compress_data(data, n, a->method, a->buf + a->len, a->cap - a->len, &out_len, &crc, &zt);   /* :253-265 */
  if (rc != ZO_OK) return rc;
  h->crc_32 = crc; h->compressed_size = out_len; h->zip_type = zt;
  a->len += out_len;
  write_local(a->buf + mem1, h);                                  /* rewrite :279-283 */
  if (a->len >= 0xFFFFFFFFull - (1u << 17)) return ZO_EINVAL;     /* would need Zip_64 */
  a->n++;
  return ZO_OK;
}

/* Zip.Create.Finish, zip-create.adb:645-756 (Zip_32 branch) */
int zo_zip_finish(zoz_archive *a, const uint8_t **bytes, uint64_t *len) {
  uint64_t central_dir_offset = a->len, central_dir_size = 0;
  int rc;
  for (int i = 0; i < a->n; i++) {
    const zoz_entry *h = &a->e[i];
    uint8_t *chb;
    if ((rc = grow(a, 46 + h->filename_length)) != 0) return rc;
    chb = a->buf + a->len;                                        /* zip-headers.adb:168-195 */
    chb[0] = 'P'; chb[1] = 'K'; chb[2] = 1; chb[3] = 2;
    put16(chb + 4, h->made_by_version);
    put16(chb + 6, h->needed_extract_version);
    put16(chb + 8, h->bit_flag);
    put16(chb + 10, h->zip_type);
    put32(chb + 12, h->file_timedate);
    put32(chb + 16, h->crc_32);
    put32(chb + 20, (uint32_t)h->compressed_size);
    put32(chb + 24, (uint32_t)h->uncompressed_size);
    put16(chb + 28, h->filename_length);
    put16(chb + 30, 0);                                           /* extra_field_length */
    put16(chb + 32, 0);                                           /* comment_length */
    put16(chb + 34, 0);                                           /* disk_number_start */
    put16(chb + 36, 0);                                           /* internal_attributes */
    put32(chb + 38, h->external_attributes);
    put32(chb + 42, (uint32_t)h->local_header_offset);
    memcpy(chb + 46, h->name, h->filename_length);
    a->len += 46 + h->filename_length;
    central_dir_size += 46 + h->filename_length;
  }
  if ((rc = grow(a, 22)) != 0) return rc;
  {
    uint8_t *eb = a->buf + a->len;                                /* zip-headers.adb:494-511 */
    eb[0] = 'P'; eb[1] = 'K'; eb[2] = 5; eb[3] = 6;
    put16(eb + 4, 0); put16(eb + 6, 0);
    put16(eb + 8, (uint32_t)a->n); put16(eb + 10, (uint32_t)a->n);
    put32(eb + 12, (uint32_t)central_dir_size);
    put32(eb + 16, (uint32_t)central_dir_offset);
    put16(eb + 20, 0);
    a->len += 22;
  }
  *bytes = a->buf; *len = a->len;
  return ZO_OK;
}

void zo_zip_free(zoz_archive *a) {
  if (!a) return;
  for (int i = 0; i < a->n; i++) free(a->e[i].name);
  free(a->e); free(a->buf); free(a);
}
