/*
 * zada_oracle_zip.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Restatement of the archive writer either side of the Deflate hot path:
 *   Zip.Create.Add_Stream   zip_lib/zip-create.adb:194-297   (local header, Compress_Data, header rewrite)
 *   Zip.Create.Finish       zip_lib/zip-create.adb:645-756   (central directory, end-of-central-dir)
 *   Zip.Headers.Write       zip_lib/zip-headers.adb:168-195 (PK\1\2), 244-276 (PK\3\4), 494-511 (PK\5\6)
 * incl. the Zip_64 promotion: Check_Size zip-create.adb:161-179, the local header extension :237-251, 283-289 and
 * zip-headers.adb:197-210, 336-355, the central extension and the Zip64 end records zip-create.adb:682-752,
 * zip-headers.adb:534-579.
 */
#include "zada_oracle.h"
#include <stdlib.h>
#include <string.h>


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
  int zip64;                     /* zip_archive_format = Zip_64 */
  uint64_t bias;                 /* test hook: pretend that `bias` bytes precede the buffer (offsets beyond 4 GiB without the data) */
} zoz_archive;

static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { put16(p, v & 0xFFFF); put16(p + 2, v >> 16); }
static void put64(uint8_t *p, uint64_t v) { put32(p, (uint32_t)v); put32(p + 4, (uint32_t)(v >> 32)); }

/* Check_Size, zip-create.adb:161-179 */
static void Check_Size(zoz_archive *a, uint64_t value) {
  const uint64_t margin = 22 + 56 + 20 + 65536 + 10;
  if (!a->zip64 && value >= 4294967296ull - margin) a->zip64 = 1;
}
/* Needs_Local_Zip_64_Header_Extension, zip-headers.adb:197-210 */
static int needs_zip64(uint64_t csize, uint64_t usize, uint64_t offset) {
  return csize >= 0xFFFFFFFFull || usize >= 0xFFFFFFFFull || offset >= 0xFFFFFFFFull;
}

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

/* Zip.Headers.Write (local header), zip-headers.adb:244-276; policy force_empty or force_zip_64 */
static void write_local(uint8_t *lhb, const zoz_entry *h, int force_zip_64) {
  lhb[0] = 'P'; lhb[1] = 'K'; lhb[2] = 3; lhb[3] = 4;
  put16(lhb + 4, h->needed_extract_version);
  put16(lhb + 6, h->bit_flag);
  put16(lhb + 8, h->zip_type);
  put32(lhb + 10, h->file_timedate);
  put32(lhb + 14, h->crc_32);
  if (force_zip_64) { put32(lhb + 18, 0xFFFFFFFFu); put32(lhb + 22, 0xFFFFFFFFu); }
  else { put32(lhb + 18, (uint32_t)h->compressed_size); put32(lhb + 22, (uint32_t)h->uncompressed_size); }
  put16(lhb + 26, h->filename_length);
  put16(lhb + 28, force_zip_64 ? 20 : 0);                        /* local_header_extension_short_length */
}

void zo_zip_set_bias(zoz_archive *a, uint64_t bias) { a->bias = bias; }

/* Zip.Create.Add_Stream, zip-create.adb:194-297.  `unicode_name` mirrors
 * Zip_Streams.Is_Unicode_Name (tools/zipada.adb:131 always sets it);
 * file_time = Zip_Streams.Time, default 16789*65536 (zip_streams.ads:223).
 * payload == NULL: compress `data` (n bytes); else the entry's payload was made elsewhere (test hook, also used to claim
 * sizes beyond 4 GiB without the data): payload_len bytes, crc, zip_type given, n = the uncompressed size to record. */
static int add_entry(zoz_archive *a, const char *entry_name, const uint8_t *data, uint64_t n, uint32_t file_time, int unicode_name,
                     const uint8_t *payload, uint64_t payload_len, uint32_t payload_crc, int payload_zt) {
  zoz_entry *h;
  uint64_t mem1, out_len = 0;
  uint32_t crc = 0; uint16_t zt = 0;
  size_t nl = strlen(entry_name);
  int rc, z64;
  if (nl > 0xFFFF) return ZO_EINVAL;
  a->e = (zoz_entry *)realloc(a->e, (size_t)(a->n + 1) * sizeof(zoz_entry));
  h = &a->e[a->n];
  memset(h, 0, sizeof *h);
  /* Add_catalogue_entry :103-134 */
  h->made_by_version = 23; h->needed_extract_version = 10; h->bit_flag = 0;
  if (unicode_name) h->bit_flag |= 0x0800;                       /* Language_Encoding_Flag_Bit */
  h->name = (char *)malloc(nl + 1);
  memcpy(h->name, entry_name, nl + 1);
  for (size_t i = 0; i < nl; i++) if (h->name[i] == '\\') h->name[i] = '/';   /* Unixify :181-192 */
  Check_Size(a, n);                                              /* :229 */
  h->file_timedate = file_time;
  h->uncompressed_size = n; h->compressed_size = n;              /* provisional :231-232 */
  h->filename_length = (uint16_t)nl;
  mem1 = a->len;
  h->local_header_offset = mem1 + a->bias;
  z64 = needs_zip64(h->compressed_size, h->uncompressed_size, h->local_header_offset);   /* :237-241, on the provisional sizes */
  if ((rc = grow(a, 30 + nl + 20 + (payload ? payload_len : n) + 64)) != 0) return rc;
  a->len += 30 + nl + (z64 ? 20 : 0);                            /* provisional header, name, extension :243-251 */
  if (payload) {
    memcpy(a->buf + a->len, payload, payload_len);
    out_len = payload_len; crc = payload_crc; zt = (uint16_t)payload_zt;
  } else {
    rc = zo_compress_data(data, n, a->method, a->buf + a->len, a->cap - a->len, &out_len, &crc, &zt);   /* :253-265 */
    if (rc != ZO_OK) return rc;
  }
  h->crc_32 = crc; h->compressed_size = out_len; h->zip_type = zt;
  if (zt == 14) h->bit_flag |= 0x0002;                            /* LZMA_EOS_Flag_Bit :266-278 */
  a->len += out_len;
  write_local(a->buf + mem1, h, z64);                             /* rewrite :279-283 */
  memcpy(a->buf + mem1 + 30, h->name, nl);
  if (z64) {                                                      /* :284-289, Local_File_Header_Extension short form */
    uint8_t *x = a->buf + mem1 + 30 + nl;
    put16(x, 1); put16(x + 2, 16);
    put64(x + 4, h->uncompressed_size); put64(x + 12, h->compressed_size);
  }
  a->n++;
  return ZO_OK;
}

int zo_zip_add(zoz_archive *a, const char *entry_name, const uint8_t *data, uint64_t n,
               uint32_t file_time, int unicode_name) {
  return add_entry(a, entry_name, data, n, file_time, unicode_name, NULL, 0, 0, 0);
}

int zo_zip_add_compressed(zoz_archive *a, const char *entry_name, const uint8_t *payload, uint64_t payload_len, uint32_t crc,
                          uint64_t uncompressed_size, int zip_type, uint32_t file_time, int unicode_name) {
  static const uint8_t none = 0;
  return add_entry(a, entry_name, NULL, uncompressed_size, file_time, unicode_name, payload ? payload : &none, payload_len, crc, zip_type);
}

/* Zip.Create.Finish, zip-create.adb:645-756 */
int zo_zip_finish(zoz_archive *a, const uint8_t **bytes, uint64_t *len) {
  uint64_t central_dir_offset = a->len + a->bias, central_dir_size = 0;
  int rc;
  if (!a->zip64 && a->n >= 65535) a->zip64 = 1;                   /* :682-687 */
  for (int i = 0; i < a->n; i++) {
    const zoz_entry *h = &a->e[i];
    uint8_t *chb;
    const int z64 = needs_zip64(h->compressed_size, h->uncompressed_size, h->local_header_offset);   /* :692-694, final sizes */
    const int xl = z64 ? 28 : 0;
    if ((rc = grow(a, 46 + h->filename_length + 28)) != 0) return rc;
    if (z64) a->zip64 = 1;                                        /* :706-707 */
    chb = a->buf + a->len;                                        /* zip-headers.adb:168-195 */
    chb[0] = 'P'; chb[1] = 'K'; chb[2] = 1; chb[3] = 2;
    put16(chb + 4, h->made_by_version);
    put16(chb + 6, h->needed_extract_version);
    put16(chb + 8, h->bit_flag);
    put16(chb + 10, h->zip_type);
    put32(chb + 12, h->file_timedate);
    put32(chb + 16, h->crc_32);
    put32(chb + 20, z64 ? 0xFFFFFFFFu : (uint32_t)h->compressed_size);
    put32(chb + 24, z64 ? 0xFFFFFFFFu : (uint32_t)h->uncompressed_size);
    put16(chb + 28, h->filename_length);
    put16(chb + 30, (uint32_t)xl);                                /* extra_field_length */
    put16(chb + 32, 0);                                           /* comment_length */
    put16(chb + 34, 0);                                           /* disk_number_start */
    put16(chb + 36, 0);                                           /* internal_attributes */
    put32(chb + 38, h->external_attributes);
    put32(chb + 42, z64 ? 0xFFFFFFFFu : (uint32_t)h->local_header_offset);
    memcpy(chb + 46, h->name, h->filename_length);
    if (z64) {                                                    /* :696-702, full form (28 bytes) */
      uint8_t *x = chb + 46 + h->filename_length;
      put16(x, 1); put16(x + 2, 24);
      put64(x + 4, h->uncompressed_size); put64(x + 12, h->compressed_size); put64(x + 20, h->local_header_offset);
    }
    a->len += 46 + h->filename_length + (uint64_t)xl;
    central_dir_size += 46 + h->filename_length + (uint64_t)xl;
  }
  if (a->n > 0) Check_Size(a, a->len + a->bias + 1);               /* :722 (current_index is 1-based) */
  if ((rc = grow(a, 22 + 56 + 20)) != 0) return rc;
  {
    uint64_t total = (uint64_t)a->n, disk_total = (uint64_t)a->n, cd_size = central_dir_size, cd_off = central_dir_offset;
    if (a->zip64) {                                               /* :729-752 */
      uint8_t *e64 = a->buf + a->len;                             /* zip-headers.adb:534-551 */
      const uint64_t e64_off = a->len + a->bias;
      e64[0] = 'P'; e64[1] = 'K'; e64[2] = 6; e64[3] = 6;
      put64(e64 + 4, 44);
      put16(e64 + 12, 0x2D); put16(e64 + 14, 0x2D);
      put32(e64 + 16, 0); put32(e64 + 20, 0);
      put64(e64 + 24, disk_total); put64(e64 + 32, total);
      put64(e64 + 40, cd_size); put64(e64 + 48, cd_off);
      a->len += 56;
      uint8_t *l64 = a->buf + a->len;                             /* zip-headers.adb:568-579 */
      l64[0] = 'P'; l64[1] = 'K'; l64[2] = 6; l64[3] = 7;
      put32(l64 + 4, 0); put64(l64 + 8, e64_off); put32(l64 + 16, 1);
      a->len += 20;
      disk_total = 0xFFFF; total = 0xFFFF; cd_size = 0xFFFFFFFFull; cd_off = 0xFFFFFFFFull;
    }
    uint8_t *eb = a->buf + a->len;                                /* zip-headers.adb:494-511 */
    eb[0] = 'P'; eb[1] = 'K'; eb[2] = 5; eb[3] = 6;
    put16(eb + 4, 0); put16(eb + 6, 0);
    put16(eb + 8, (uint32_t)disk_total); put16(eb + 10, (uint32_t)total);
    put32(eb + 12, (uint32_t)cd_size);
    put32(eb + 16, (uint32_t)cd_off);
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
