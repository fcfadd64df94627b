/*
 * zada_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See zada_oracle.h.
 *
 * Plain C restatement of the reference's Deflate encoder.  Every function cites the
 * reference file:line it follows (paths relative to the reference tree, zip_lib/).
 * The structure deliberately mirrors the Ada text (same names, same order of
 * operations, same buffers) so that the two can be read side by side.
 *
 * PARITY: unpinned against the Ada binary (no Ada toolchain in the image); LZ77 stage
 * pinned against zlib 1.2.11 deflateTune, entropy stage pinned by round trip.
 */
#include "zada_oracle.h"

#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/*  CRC-32  --  zip-crc_crypto.adb:29-76                                      */
/* ------------------------------------------------------------------------- */

static uint32_t CRC32_Table[256];
static int table_empty = 1;

/* zip-crc_crypto.adb:31-47 */
static void Prepare_table(void) {
  const uint32_t Seed = 0xEDB88320u;
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t l = i;
    for (int bit = 0; bit <= 7; bit++) {
      if ((l & 1) == 0) l = l >> 1; else l = (l >> 1) ^ Seed;
    }
    CRC32_Table[i] = l;
  }
}

/* zip-crc_crypto.adb:64-71 */
uint32_t zo_crc32_init(void) {
  if (table_empty) { Prepare_table(); table_empty = 0; }
  return 0xFFFFFFFFu;
}

/* zip-crc_crypto.adb:49-60 */
uint32_t zo_crc32_update(uint32_t crc, const uint8_t *buf, uint64_t n) {
  uint32_t local_CRC = crc;
  if (table_empty) { Prepare_table(); table_empty = 0; }
  for (uint64_t i = 0; i < n; i++)
    local_CRC = CRC32_Table[0xFF & (local_CRC ^ (uint32_t)buf[i])] ^ (local_CRC >> 8);
  return local_CRC;
}

/* zip-crc_crypto.adb:73-76 */
uint32_t zo_crc32_final(uint32_t crc) { return ~crc; }

/* ------------------------------------------------------------------------- */
/*  Huffman.Encoding.Prepare_Codes  --  huffman-encoding.adb:34-80            */
/* ------------------------------------------------------------------------- */

/* huffman-encoding.adb:34-43 */
static int32_t Invert(int32_t code, int bit_length) {
  int32_t a = code, b = 0;
  for (int i = 1; i <= bit_length; i++) { b = b * 2 + a % 2; a = a / 2; }
  return b;
}

/* huffman-encoding.adb:45-80.  NB bl_count(0) is NOT zeroed (reference behaviour). */
void zo_prepare_codes(const int *lengths, int n, int max_huffman_bits, int invert, int32_t *codes) {
  int32_t bl_count[32], next_code[32];
  int32_t code = 0;
  memset(bl_count, 0, sizeof bl_count);
  memset(next_code, 0, sizeof next_code);
  for (int i = 0; i < n; i++) bl_count[lengths[i]]++;              /* Step 1 */
  for (int bits = 1; bits <= max_huffman_bits; bits++) {           /* Step 2 */
    code = (code + bl_count[bits - 1]) * 2;
    next_code[bits] = code;
  }
  for (int k = 0; k < n; k++) {                                    /* Step 3 */
    int bl = lengths[k];
    if (bl > 0) { codes[k] = next_code[bl]; next_code[bl]++; } else codes[k] = 0;
  }
  if (invert) for (int i = 0; i < n; i++) codes[i] = Invert(codes[i], lengths[i]);
}

/* ------------------------------------------------------------------------- */
/*  Huffman.Encoding.Length_Limited_Coding                                    */
/*  huffman-encoding-length_limited_coding.adb:46-280                         */
/* ------------------------------------------------------------------------- */

#define LL_MAX_BITS 17   /* capacity of the tables: Deflate instantiates 15 and 7, BZip2 15, 16 and 17 (bzip2-encoding.adb:900-903) */
#define LL_MAX_ALPHA 288
#define LL_NULL (-1)

typedef struct { uint64_t weight; int64_t count; int tail; int in_use; } ll_node;      /* :56-61 */
typedef struct { uint64_t weight; int symbol; } ll_leaf;                               /* :63-66 */

typedef struct {
  ll_node pool[2 * LL_MAX_BITS * (LL_MAX_BITS + 1)];                                   /* :68-70 */
  int pool_last, pool_next;
  int lists[LL_MAX_BITS][2];                                                           /* :72-73 */
  ll_leaf leaves[LL_MAX_ALPHA];                                                        /* :75-76 */
  int64_t num_symbols;
  int max_bits;
} ll_state;

/* :87-93 */
static void Init_Node(ll_state *s, uint64_t weight, int64_t count, int tail, int node_idx) {
  s->pool[node_idx].weight = weight;
  s->pool[node_idx].count = count;
  s->pool[node_idx].tail = tail;
  s->pool[node_idx].in_use = 1;
}

/* :97-122 */
static int Get_Free_Node(ll_state *s, int use_lists) {
  for (;;) {
    if (s->pool_next > s->pool_last) {
      for (int i = 0; i <= s->pool_last; i++) s->pool[i].in_use = 0;       /* Garbage collection */
      if (use_lists) {
        for (int i = 0; i <= s->max_bits * 2 - 1; i++) {
          int node_idx = s->lists[i / 2][i % 2];
          while (node_idx != LL_NULL) { s->pool[node_idx].in_use = 1; node_idx = s->pool[node_idx].tail; }
        }
      }
      s->pool_next = 0;
    }
    if (!s->pool[s->pool_next].in_use) break;
    s->pool_next++;
  }
  s->pool_next++;
  return s->pool_next - 1;
}

/* :131-163 */
static void Boundary_PM(ll_state *s, int index, int final) {
  int newchain, oldchain;
  const int64_t lastcount = s->pool[s->lists[index][1]].count;
  uint64_t sum;
  if (index == 0 && lastcount >= s->num_symbols) return;
  newchain = Get_Free_Node(s, 1);
  oldchain = s->lists[index][1];
  s->lists[index][0] = oldchain; s->lists[index][1] = newchain;
  if (index == 0) {
    Init_Node(s, s->leaves[lastcount].weight, lastcount + 1, LL_NULL, newchain);
  } else {
    sum = s->pool[s->lists[index - 1][0]].weight + s->pool[s->lists[index - 1][1]].weight;
    if (lastcount < s->num_symbols && sum > s->leaves[lastcount].weight) {
      Init_Node(s, s->leaves[lastcount].weight, lastcount + 1, s->pool[oldchain].tail, newchain);
    } else {
      Init_Node(s, sum, lastcount, s->lists[index - 1][1], newchain);
      if (!final) { Boundary_PM(s, index - 1, 0); Boundary_PM(s, index - 1, 0); }
    }
  }
}

/* :167-174 */
static void Init_Lists(ll_state *s) {
  const int node0 = Get_Free_Node(s, 0);
  const int node1 = Get_Free_Node(s, 0);
  Init_Node(s, s->leaves[0].weight, 1, LL_NULL, node0);
  Init_Node(s, s->leaves[1].weight, 2, LL_NULL, node1);
  for (int i = 0; i < s->max_bits; i++) { s->lists[i][0] = node0; s->lists[i][1] = node1; }
}

/* :196-223 -- the reference's own quicksort; its tie-breaking decides which of several
 * equal-weight symbols gets the shorter code, so it is transcribed literally. */
static void Quick_sort(ll_leaf *a, int64_t n) {
  int64_t i, j;
  ll_leaf p, t;
  if (n < 2) return;
  p = a[n / 2];
  i = 0; j = n - 1;
  for (;;) {
    while (a[i].weight < p.weight) i++;
    while (p.weight < a[j].weight) j--;
    if (i >= j) break;
    t = a[i]; a[i] = a[j]; a[j] = t;
    i++; j--;
  }
  Quick_sort(a, i);
  Quick_sort(a + i, n - i);
}

/* :46-54, 227-280 */
int zo_llhc(const uint64_t *freq, int n, int max_bits, int *bit_lengths) {
  ll_state st;
  ll_state *s = &st;
  int64_t num_Boundary_PM_runs;
  if (n > LL_MAX_ALPHA || max_bits > LL_MAX_BITS || max_bits < 1) return ZO_EINVAL;
  memset(s, 0, sizeof *s);
  s->max_bits = max_bits;
  s->pool_last = 2 * max_bits * (max_bits + 1) - 1;
  s->pool_next = 0;
  for (int i = 0; i <= s->pool_last; i++) { s->pool[i].tail = LL_NULL; s->pool[i].in_use = 0; }
  for (int a = 0; a < n; a++) bit_lengths[a] = 0;
  s->num_symbols = 0;
  for (int a = 0; a < n; a++) {
    if (freq[a] > 0) { s->leaves[s->num_symbols].weight = freq[a]; s->leaves[s->num_symbols].symbol = a; s->num_symbols++; }
  }
  if (s->num_symbols > ((int64_t)1 << max_bits)) return ZO_EINVAL;   /* too_many_symbols_for_length_limit */
  if (s->num_symbols == 0) return 0;
  if (s->num_symbols == 1) { bit_lengths[s->leaves[0].symbol] = 1; return 0; }
  Quick_sort(s->leaves, s->num_symbols);
  Init_Lists(s);
  num_Boundary_PM_runs = 2 * s->num_symbols - 4;
  for (int64_t i = 1; i <= num_Boundary_PM_runs; i++)
    Boundary_PM(s, max_bits - 1, i == num_Boundary_PM_runs);
  /* Extract_Bit_Lengths :180-189 */
  for (int node_idx = s->lists[max_bits - 1][1]; node_idx != LL_NULL; node_idx = s->pool[node_idx].tail)
    for (int64_t i = 0; i <= s->pool[node_idx].count - 1; i++)
      bit_lengths[s->leaves[i].symbol]++;
  return 0;
}

/* ------------------------------------------------------------------------- */
/*  LZ77.Encode / LZ77_using_IZ  --  lz77.adb:460-943                          */
/* ------------------------------------------------------------------------- */

typedef struct zo_lz_io {
  uint8_t (*Read_Byte)(void *u);
  int (*More_Bytes)(void *u);
  void (*Write_Literal)(void *u, uint8_t b);
  void (*Write_DL_Code)(void *u, int distance, int length);
  void *u;
} zo_lz_io;

enum {
  HASH_BITS = 15, HASH_SIZE = 1 << HASH_BITS, HASH_MASK = HASH_SIZE - 1,        /* :461-463 */
  WSIZE = 32768, WMASK = WSIZE - 1,                                               /* :464-465 */
  NIL = 0, TOO_FAR = 4096,                                                        /* :467-468 */
  MIN_MATCH = 3, MAX_MATCH = 258, MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1,      /* :495-498 */
  MAX_DIST = WSIZE - MIN_LOOKAHEAD,                                               /* :500 = 32506 */
  H_SHIFT = (HASH_BITS + MIN_MATCH - 1) / MIN_MATCH                               /* :501 = 5 */
};

typedef struct { int64_t good_length, max_lazy, nice_length; uint32_t max_chain; } iz_config;   /* :527-532 */

static const iz_config configuration_table[11] = {                                /* :534-546 */
  {0, 0, 0, 0}, {4, 4, 8, 4}, {4, 5, 16, 8}, {4, 6, 32, 32}, {4, 4, 16, 16}, {8, 16, 32, 32},
  {8, 16, 128, 128}, {8, 32, 128, 256}, {32, 128, 258, 1024}, {32, 258, 258, 4096}, {34, 258, 258, 4096}
};

typedef struct {
  const zo_lz_io *io;
  uint8_t window[2 * WSIZE];                                                      /* :478 */
  uint64_t prev[WSIZE];                                                           /* :484 */
  uint64_t head[HASH_SIZE];                                                       /* :488 */
  uint64_t window_size;
  int sliding;
  uint32_t ins_h;
  int64_t prev_length, strstart, match_start, lookahead;
  int eofile;
  uint32_t max_chain_length;
  int64_t max_lazy_match, good_match, nice_match;
} iz_state;

/* :553-557 */
#define UPDATE_HASH(h, c) ((h) = ((((uint32_t)(h)) << H_SHIFT) ^ (uint32_t)(c)) & HASH_MASK)

/* :566-573 */
static inline void INSERT_STRING(iz_state *z, int64_t s, int64_t *match_head) {
  UPDATE_HASH(z->ins_h, z->window[s + MIN_MATCH - 1]);
  *match_head = (int64_t)z->head[z->ins_h];
  z->prev[(uint64_t)s & WMASK] = (uint64_t)*match_head;
  z->head[z->ins_h] = (uint64_t)s;
}

/* :575-586 */
static void Read_buf(iz_state *z, int64_t from, uint64_t amount, int64_t *actual) {
  uint64_t need = amount;
  *actual = 0;
  while (need > 0 && z->io->More_Bytes(z->io->u)) {
    z->window[from + *actual] = z->io->Read_Byte(z->io->u);
    (*actual)++;
    need--;
  }
}

/* :596-661 */
static void Fill_window(iz_state *z) {
  uint64_t more, m;
  int64_t n;
  for (;;) {
    more = z->window_size - (uint64_t)z->lookahead - (uint64_t)z->strstart;
    if (z->strstart >= WSIZE + MAX_DIST && z->sliding) {
      memmove(z->window, z->window + WSIZE, WSIZE);
      z->match_start = (int64_t)(uint16_t)((uint16_t)z->match_start - (uint16_t)(WSIZE % 65536));
      z->strstart -= WSIZE;
      for (uint32_t nn = 0; nn < HASH_SIZE; nn++) {
        m = z->head[nn];
        z->head[nn] = (m >= WSIZE) ? m - WSIZE : NIL;
      }
      for (uint32_t nn = 0; nn < WSIZE; nn++) {
        m = z->prev[nn];
        z->prev[nn] = (m >= WSIZE) ? m - WSIZE : NIL;
      }
      more += WSIZE;
    }
    if (z->eofile) break;
    Read_buf(z, z->strstart + z->lookahead, more, &n);
    if (n == 0) z->eofile = 1; else z->lookahead += n;
    if (z->lookahead >= MIN_LOOKAHEAD || z->eofile) break;
  }
}

/* :671-706 */
static void LM_Init(iz_state *z, int pack_level) {
  z->sliding = 0;
  if (z->window_size == 0) { z->sliding = 1; z->window_size = 2 * (uint64_t)WSIZE; }
  memset(z->head, 0, sizeof z->head);
  z->max_lazy_match = configuration_table[pack_level].max_lazy;
  z->good_match = configuration_table[pack_level].good_length;
  z->nice_match = configuration_table[pack_level].nice_length;
  z->max_chain_length = configuration_table[pack_level].max_chain;
  z->strstart = 0;
  Read_buf(z, 0, WSIZE, &z->lookahead);
  if (z->lookahead == 0) { z->eofile = 1; return; }
  z->eofile = 0;
  if (z->lookahead < MIN_LOOKAHEAD) Fill_window(z);
  z->ins_h = 0;
  for (int j = 0; j <= MIN_MATCH - 2; j++) UPDATE_HASH(z->ins_h, z->window[j]);
}

/* :715-825 */
static void Longest_Match(iz_state *z, int64_t *current_match, int64_t *longest) {
  uint32_t chain_length = z->max_chain_length;
  int64_t scan = z->strstart, match, len, best_len = z->prev_length, limit;
  const int64_t strend = z->strstart + MAX_MATCH;
  int64_t scan_end = scan + best_len;
  const uint8_t *window = z->window;
  if (z->strstart > MAX_DIST) limit = z->strstart - MAX_DIST; else limit = NIL;
  if (z->prev_length >= z->good_match) chain_length = chain_length / 4;
  for (;;) {
    if (*current_match >= z->strstart) { *longest = MIN_MATCH - 1; return; }      /* :740-744 */
    match = *current_match;
    if (window[match + best_len] != window[scan_end] ||
        window[match + best_len - 1] != window[scan_end - 1] ||
        window[match] != window[scan] ||
        window[match + 1] != window[scan + 1]) {
      match++;                                                                      /* C: continue */
    } else {
      scan += 2; match += 2;
      /* :772-800 -- lookahead is tested only every 8th comparison */
      for (;;) {
        scan++; match++; if (window[scan] != window[match]) break;
        scan++; match++; if (window[scan] != window[match]) break;
        scan++; match++; if (window[scan] != window[match]) break;
        scan++; match++; if (window[scan] != window[match]) break;
        scan++; match++; if (window[scan] != window[match]) break;
        scan++; match++; if (window[scan] != window[match]) break;
        scan++; match++; if (window[scan] != window[match]) break;
        scan++; match++; if (window[scan] != window[match] || scan >= strend) break;
      }
      len = MAX_MATCH - (strend - scan);
      scan = strend - MAX_MATCH;
      if (len > best_len) {
        z->match_start = *current_match;
        best_len = len;
        if (len >= z->nice_match) break;
        scan_end = scan + best_len;
      }
    }
    *current_match = (int64_t)z->prev[(uint64_t)*current_match & WMASK];
    if (*current_match <= limit) break;
    chain_length--;
    if (chain_length == 0) break;
  }
  *longest = best_len;
}

/* :827-933 */
static void LZ77_part_of_IZ_Deflate(iz_state *z) {
  int64_t hash_head = NIL, prev_match, match_length = MIN_MATCH - 1, max_insert;
  int match_available = 0;
  z->match_start = 0;
  while (z->lookahead != 0) {
    if (z->lookahead >= MIN_MATCH) INSERT_STRING(z, z->strstart, &hash_head);
    z->prev_length = match_length;
    prev_match = z->match_start;
    match_length = MIN_MATCH - 1;
    if (hash_head != NIL && z->prev_length < z->max_lazy_match && z->strstart - hash_head <= MAX_DIST) {
      if (z->nice_match > z->lookahead) z->nice_match = z->lookahead;
      Longest_Match(z, &hash_head, &match_length);
      if (match_length > z->lookahead) match_length = z->lookahead;
      if (match_length == MIN_MATCH && z->strstart - z->match_start > TOO_FAR) match_length = MIN_MATCH - 1;
    }
    if (z->prev_length >= MIN_MATCH && match_length <= z->prev_length) {
      max_insert = z->strstart + z->lookahead - MIN_MATCH;
      z->io->Write_DL_Code(z->io->u, (int)(z->strstart - 1 - prev_match), (int)z->prev_length);
      z->lookahead -= (z->prev_length - 1);
      z->prev_length -= 2;
      do {
        z->strstart++;
        if (z->strstart <= max_insert) INSERT_STRING(z, z->strstart, &hash_head);
        z->prev_length--;
      } while (z->prev_length != 0);
      z->strstart++;
      match_available = 0;
      match_length = MIN_MATCH - 1;
    } else if (match_available) {
      z->io->Write_Literal(z->io->u, z->window[z->strstart - 1]);
      z->strstart++;
      z->lookahead--;
    } else {
      match_available = 1;
      z->strstart++;
      z->lookahead--;
    }
    if (z->lookahead < MIN_LOOKAHEAD) Fill_window(z);
  }
  if (match_available) z->io->Write_Literal(z->io->u, z->window[z->strstart - 1]);
}

/* :460, 936-943 */
static int LZ77_using_IZ(const zo_lz_io *io, int level) {
  iz_state *z = (iz_state *)calloc(1, sizeof *z);   /* zero-filled: beyond-EOF window bytes are 0 */
  if (!z) return ZO_ENOMEM;
  z->io = io;
  z->window_size = 0;
  LM_Init(z, level);
  LZ77_part_of_IZ_Deflate(z);
  free(z);
  return 0;
}

/* lz77.adb:2181-2198 (dispatch), restricted to the methods Deflate uses */
enum { LZ_NO_LZ77 = 0 };
static int LZ77_Encode(const zo_lz_io *io, int level) {
  if (level == LZ_NO_LZ77) {                                    /* :2191-2194 */
    while (io->More_Bytes(io->u)) io->Write_Literal(io->u, io->Read_Byte(io->u));
    return 0;
  }
  if (level >= 4 && level <= 10) return LZ77_using_IZ(io, level);
  return ZO_EINVAL;
}

/* ---- token-level entry point (tests) ---- */
typedef struct { const uint8_t *in; uint64_t n, pos; uint32_t *tok; uint64_t cap, cnt; } tok_ctx;
static uint8_t tk_read(void *u) { tok_ctx *c = (tok_ctx *)u; return c->in[c->pos++]; }
static int tk_more(void *u) { tok_ctx *c = (tok_ctx *)u; return c->pos < c->n; }
static void tk_lit(void *u, uint8_t b) { tok_ctx *c = (tok_ctx *)u; if (c->cnt < c->cap) c->tok[c->cnt] = b; c->cnt++; }
static void tk_dl(void *u, int d, int l) {
  tok_ctx *c = (tok_ctx *)u;
  if (c->cnt < c->cap) c->tok[c->cnt] = ZO_TOKEN_MATCH | ((uint32_t)l << 16) | (uint32_t)d;
  c->cnt++;
}
uint64_t zo_lz77_tokens(const uint8_t *in, uint64_t n, int level, uint32_t *tokens, uint64_t cap) {
  tok_ctx c = {in, n, 0, tokens, cap, 0};
  zo_lz_io io = {tk_read, tk_more, tk_lit, tk_dl, &c};
  if (LZ77_Encode(&io, level) != 0) return 0;
  return c.cnt;
}

/* ------------------------------------------------------------------------- */
/*  Zip.Compress.Deflate  --  zip-compress-deflate.adb                         */
/* ------------------------------------------------------------------------- */

#define default_byte_IO_buffer_size (1024 * 1024)      /* zip-compress.adb:62 */
#define feedback_steps 100                              /* zip-compress.ads:187 */

typedef uint64_t Count_type;                            /* :225 (63-bit) */
#define Count_type_Last ((Count_type)0x3FFFFFFFFFFFFFFEull)

typedef struct { int bit_length; int32_t code; } Length_Code_Pair;        /* huffman-encoding.ads */
typedef struct { Length_Code_Pair lit_len[288]; Length_Code_Pair dis[32]; } Deflate_Huff_Descriptors;  /* :187-192 */

enum { max_expand = 14, code_for_max_expand = 266 };    /* :927-928 */
enum { plain_byte = 0, distance_length = 1 };           /* :931 */
typedef struct {                                        /* :932-938 */
  uint8_t kind, plain;
  int lz_distance, lz_length;
  uint8_t lz_expanded[max_expand];
} LZ_atom;

enum { LZ_buffer_size = 1 << 17 };                      /* :942 */
#define LZ_IDX(x) ((uint32_t)(x) & (LZ_buffer_size - 1))

enum { bt_stored = 0, bt_fixed = 1, bt_dynamic = 2, bt_reserved = 3 };    /* :995 */
enum { End_Of_Block = 256 };                            /* :706 */

enum { min_step = 750, slider_size = 4096, half_slider_size = 2048, slider_max = 4095 };  /* :1294, 1313-1315 */
typedef struct { uint32_t slider_step; int cutting_threshold; } Step_threshold_metric;     /* :1296-1300 (metric always L1_tweaked) */
static const Step_threshold_metric step_choice[3] = {   /* :1304-1308 */
  {8 * min_step, 420}, {4 * min_step, 430}, {min_step, 2050}
};

typedef struct {
  /* parameters */
  const uint8_t *in; uint64_t in_size, in_pos;
  int input_size_known; uint64_t input_size;
  int method;
  zo_feedback_fn feedback; void *fb_user;
  zo_trace_fn trace; void *tr_user;
  uint32_t CRC;
  /* output */
  uint8_t *out; uint64_t out_cap; uint64_t output_size;
  /* IO_Buffers_Type, zip-compress.ads:205-214 (1-based indices kept) */
  uint8_t *InBuf; uint64_t InBuf_len; uint8_t *OutBuf;
  uint64_t InBufIdx, OutBufIdx, MaxInBufIdx; int InputEoF;
  /* bit buffer :125-127 */
  uint32_t bit_buffer; int valid_bits;
  /* block state :722, 993-997 */
  Deflate_Huff_Descriptors Deflate_fixed_descriptors, curr_descr;
  int block_to_finish, last_block_marked, last_block_type;
  /* LZ buffer :1279-1281 */
  LZ_atom *lz_buffer; uint32_t lz_buffer_index; int past_lz_data;
  uint64_t atoms_flushed_before;   /* trace only: global index of ring slot 0/65536 */
  /* Encode locals :1462-1511 */
  uint64_t feedback_milestone, Bytes_in;
  uint8_t Text_Buf[32768]; uint32_t R;
  /* exceptions */
  jmp_buf escape; int escape_code;
  /* replay of an explicit token stream (zo_deflate_from_tokens) */
  const uint32_t *replay; uint64_t nreplay;
} deflate_ctx;

/* ---- zip-compress.adb:455-490 ---- */

/* Read_Block :455-466, on an in-memory stream */
static void Read_Block(deflate_ctx *c) {
  uint64_t left = c->in_size - c->in_pos;
  uint64_t k = left < c->InBuf_len ? left : c->InBuf_len;
  memcpy(c->InBuf + 1, c->in + c->in_pos, k);
  c->in_pos += k;
  c->MaxInBufIdx = k;
  c->InputEoF = (c->MaxInBufIdx == 0);
  c->InBufIdx = 1;
}

/* Write_Block :468-490 (crypto in clear mode) */
static void Write_Block(deflate_ctx *c) {
  const uint64_t amount = c->OutBufIdx - 1;
  uint64_t before = c->output_size;
  c->output_size += amount;                                                     /* Increment */
  if (c->input_size_known && c->output_size >= c->input_size) {
    c->escape_code = ZO_INEFFICIENT; longjmp(c->escape, 1);                     /* raise Compression_inefficient */
  }
  if (c->output_size > c->out_cap) { c->escape_code = ZO_EINVAL; longjmp(c->escape, 1); }
  memcpy(c->out + before, c->OutBuf + 1, amount);                               /* Block_Write */
  c->OutBufIdx = 1;
}

/* ---- :101-116 ---- */
static inline void Put_byte(deflate_ctx *c, uint8_t B) {
  c->OutBuf[c->OutBufIdx] = B;
  c->OutBufIdx++;
  if (c->OutBufIdx > default_byte_IO_buffer_size) Write_Block(c);
}
static void Flush_byte_buffer(deflate_ctx *c) { if (c->OutBufIdx > 1) Write_Block(c); }

/* :129-137 */
static void Flush_bit_buffer(deflate_ctx *c) {
  while (c->valid_bits > 0) {
    Put_byte(c, (uint8_t)(c->bit_buffer & 0xFF));
    c->bit_buffer >>= 8;
    c->valid_bits = c->valid_bits - 8 > 0 ? c->valid_bits - 8 : 0;
  }
  c->bit_buffer = 0;
}

/* :144-160 */
static inline void Put_Bits(deflate_ctx *c, uint32_t code, int code_size) {
  c->bit_buffer |= (c->valid_bits >= 32) ? 0u : (code << c->valid_bits);      /* Ada Shift_Left by >= 32 gives 0 */
  c->valid_bits += code_size;
  if (c->valid_bits > 32) {
    Put_byte(c, (uint8_t)(c->bit_buffer & 0xFF));
    Put_byte(c, (uint8_t)((c->bit_buffer >> 8) & 0xFF));
    Put_byte(c, (uint8_t)((c->bit_buffer >> 16) & 0xFF));
    Put_byte(c, (uint8_t)((c->bit_buffer >> 24) & 0xFF));
    c->valid_bits -= 32;
    c->bit_buffer = code >> (code_size - c->valid_bits);
  }
}

/* :199-223 */
static void Build_descriptors_bl(const int *bl_for_lit_len, const int *bl_for_dis, Deflate_Huff_Descriptors *new_d) {
  for (int i = 0; i < 288; i++) { new_d->lit_len[i].bit_length = bl_for_lit_len[i]; new_d->lit_len[i].code = -1; }
  for (int i = 0; i < 32; i++) { new_d->dis[i].bit_length = bl_for_dis[i]; new_d->dis[i].code = -1; }
}

/* Tweak_for_better_RLE :238-318 */
static void Tweak_for_better_RLE(Count_type *counts, int counts_len) {
  int length = counts_len, stride;
  Count_type symbol, sum, limit, new_count;
  uint8_t good_for_rle[288];
  memset(good_for_rle, 0, sizeof good_for_rle);
  for (;;) {                                                       /* 1) */
    if (length == 0) return;
    if (counts[length - 1] != 0) break;
    length--;
  }
  symbol = counts[0];                                              /* 2) */
  stride = 0;
  for (int i = 0; i <= length; i++) {
    if (i == length || counts[i] != symbol) {
      if ((symbol == 0 && stride >= 5) || (symbol != 0 && stride >= 7))
        for (int k = 0; k <= stride - 1; k++) good_for_rle[i - k - 1] = 1;
      stride = 1;
      if (i != length) symbol = counts[i];
    } else {
      stride++;
    }
  }
  stride = 0;                                                      /* 3) */
  limit = counts[0];
  sum = 0;
  for (int i = 0; i <= length; i++) {
    int64_t diff = 0;
    if (i != length) { diff = (int64_t)counts[i] - (int64_t)limit; if (diff < 0) diff = -diff; }
    if (i == length || good_for_rle[i] || (i > 0 && good_for_rle[i - 1]) || diff >= 4) {
      if (stride >= 4 || (stride >= 3 && sum == 0)) {
        new_count = (sum + (Count_type)stride / 2) / (Count_type)stride;
        if (new_count < 1) new_count = 1;
        if (sum == 0) new_count = 0;
        for (int k = 0; k <= stride - 1; k++) counts[i - k - 1] = new_count;
      }
      stride = 0;
      sum = 0;
      if (i < length - 3) limit = (counts[i] + counts[i + 1] + counts[i + 2] + counts[i + 3] + 2) / 4;
      else if (i < length) limit = counts[i];
      else limit = 0;
    }
    stride++;
    if (i != length) sum += counts[i];
  }
}

/* Build_descriptors (stats) :324-371 */
static void Build_descriptors_stats(const Count_type *stats_lit_len, const Count_type *stats_dis, Deflate_Huff_Descriptors *d) {
  int bl_for_lit_len[288], bl_for_dis[32];
  Count_type stats_dis_copy[32];
  int used = 0;
  memcpy(stats_dis_copy, stats_dis, sizeof stats_dis_copy);
  /* Patch_statistics_for_buggy_decoders :340-365 */
  for (int i = 0; i < 32; i++) if (stats_dis_copy[i] != 0) used++;
  if (used == 0) { stats_dis_copy[0] = 1; stats_dis_copy[1] = 1; }
  else if (used == 1) { if (stats_dis_copy[0] == 0) stats_dis_copy[0] = 1; else stats_dis_copy[1] = 1; }
  zo_llhc(stats_lit_len, 288, 15, bl_for_lit_len);
  zo_llhc(stats_dis_copy, 32, 15, bl_for_dis);
  Build_descriptors_bl(bl_for_lit_len, bl_for_dis, d);
}

/* Convert :382-403 ; tweak :417-421 ; L1_tweaked :426-433 ; Similar :457-490 (L1_tweaked only) */
static const int32_t tweak[17] = {0, 100, 255, 379, 490, 594, 694, 791, 885, 978, 1069, 1159, 1249, 1338, 1426, 1513, 1600};
static void Convert(const Deflate_Huff_Descriptors *h, int32_t *bv) {
  int j = 0;
  for (int i = 0; i < 288; i++) bv[j++] = h->lit_len[i].bit_length == 0 ? 16 : h->lit_len[i].bit_length;
  for (int i = 0; i < 32; i++) bv[j++] = h->dis[i].bit_length == 0 ? 16 : h->dis[i].bit_length;
}
static int Similar(deflate_ctx *c, const Deflate_Huff_Descriptors *h1, const Deflate_Huff_Descriptors *h2, int threshold, int64_t where) {
  int32_t b1[320], b2[320];
  int64_t dist = 0, thres = (int64_t)threshold * tweak[1];
  Convert(h1, b1); Convert(h2, b2);
  for (int i = 0; i < 320; i++) { int32_t d = tweak[b1[i]] - tweak[b2[i]]; dist += d < 0 ? -d : d; }
  if (c->trace) c->trace(c->tr_user, ZO_TR_SIMILAR, where, dist, thres, 0);
  return dist < thres;
}

/* Recyclable :495-508 */
static int Recyclable(const Deflate_Huff_Descriptors *h_old, const Deflate_Huff_Descriptors *h_new) {
  for (int i = 0; i < 288; i++) if (h_old->lit_len[i].bit_length == 0 && h_new->lit_len[i].bit_length > 0) return 0;
  for (int i = 0; i < 32; i++) if (h_old->dis[i].bit_length == 0 && h_new->dis[i].bit_length > 0) return 0;
  return 1;
}

/* Prepare_Huffman_Codes :513-520 */
static void Prepare_Huffman_Codes(Deflate_Huff_Descriptors *dhd) {
  int len[288]; int32_t codes[288];
  for (int i = 0; i < 288; i++) len[i] = dhd->lit_len[i].bit_length;
  zo_prepare_codes(len, 288, 15, 1, codes);
  for (int i = 0; i < 288; i++) dhd->lit_len[i].code = codes[i];
  for (int i = 0; i < 32; i++) len[i] = dhd->dis[i].bit_length;
  zo_prepare_codes(len, 32, 15, 1, codes);
  for (int i = 0; i < 32; i++) dhd->dis[i].code = codes[i];
}

/* Put_Huffman_Code :523-532 */
static inline void Put_Huffman_Code(deflate_ctx *c, Length_Code_Pair lc) { Put_Bits(c, (uint32_t)lc.code, lc.bit_length); }

/* Put_Compression_Structure :549-704 */
static const int extra_bits_needed[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};                 /* :592-593 */
static const int alphabet_permutation[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};     /* :657-658 */

typedef struct {
  deflate_ctx *c; int effective;
  uint64_t truc_freq[19]; Length_Code_Pair truc[19];
  int cs_bl[321]; int last_cs_bl;
} pcs_state;

/* Emit_data_compression_atom :598-614 */
static void Emit_data_compression_atom(pcs_state *p, int x, uint32_t extra_code) {
  if (!p->effective) { p->truc_freq[x]++; return; }
  Put_Huffman_Code(p->c, p->truc[x]);
  if (extra_bits_needed[x] > 0) Put_Bits(p->c, extra_code, extra_bits_needed[x]);
}

/* Emit_data_compression_structures :597-650 */
static void Emit_data_compression_structures(pcs_state *p) {
  int idx = 1, rep;
  const int *cs_bl = p->cs_bl;
  for (;;) {
    rep = 1;
    for (int j = idx + 1; j <= p->last_cs_bl; j++) { if (cs_bl[j] != cs_bl[idx]) break; rep++; }
    if (idx > 1 && cs_bl[idx] == cs_bl[idx - 1] && rep >= 3 && !(cs_bl[idx] == 0 && rep > 6)) {
      rep = rep < 6 ? rep : 6;
      Emit_data_compression_atom(p, 16, (uint32_t)(rep - 3));
      idx += rep;
    } else if (cs_bl[idx] == 0 && rep >= 3) {
      if (rep <= 10) {
        Emit_data_compression_atom(p, 17, (uint32_t)(rep - 3));
      } else {
        rep = rep < 138 ? rep : 138;
        Emit_data_compression_atom(p, 18, (uint32_t)(rep - 11));
      }
      idx += rep;
    } else {
      Emit_data_compression_atom(p, cs_bl[idx], 0);
      idx++;
    }
    if (idx > p->last_cs_bl) break;
  }
}

static void Put_Compression_Structure(deflate_ctx *c, const Deflate_Huff_Descriptors *dhd, int cost_analysis, Count_type *bits) {
  pcs_state p;
  int truc_bl[19];
  int max_used_lln_code = 0, max_used_dis_code = 0, idx = 0, a_non_zero;
  memset(&p, 0, sizeof p);
  p.c = c;
  /* Concatenate_all_bit_lengths :565-590 */
  for (int a = 287; a >= 0; a--) if (dhd->lit_len[a].bit_length > 0) { max_used_lln_code = a; break; }
  for (int a = 31; a >= 0; a--) if (dhd->dis[a].bit_length > 0) { max_used_dis_code = a; break; }
  for (int a = 0; a <= max_used_lln_code; a++) p.cs_bl[++idx] = dhd->lit_len[a].bit_length;
  for (int a = 0; a <= max_used_dis_code; a++) p.cs_bl[++idx] = dhd->dis[a].bit_length;
  p.last_cs_bl = idx;
  /* :664-678 */
  p.effective = 0;
  Emit_data_compression_structures(&p);
  zo_llhc(p.truc_freq, 19, 7, truc_bl);
  a_non_zero = 3;
  for (int a = 0; a <= 18; a++) if (a > a_non_zero && truc_bl[alphabet_permutation[a]] > 0) a_non_zero = a;
  if (cost_analysis) {
    *bits += 14 + (Count_type)(1 + a_non_zero) * 3;
    for (int a = 0; a <= 18; a++) *bits += (Count_type)(p.truc_freq[a] * (uint64_t)(truc_bl[a] + extra_bits_needed[a]));
  } else {
    int32_t codes[19];
    zo_prepare_codes(truc_bl, 19, 15, 1, codes);
    for (int a = 0; a <= 18; a++) { p.truc[a].bit_length = truc_bl[a]; p.truc[a].code = codes[a]; }
    Put_Bits(c, (uint32_t)(max_used_lln_code - 256), 5);
    Put_Bits(c, (uint32_t)max_used_dis_code, 5);
    Put_Bits(c, (uint32_t)(a_non_zero - 3), 4);
    for (int a = 0; a <= a_non_zero; a++) Put_Bits(c, (uint32_t)p.truc[alphabet_permutation[a]].bit_length, 3);
    p.effective = 1;
    Emit_data_compression_structures(&p);
  }
}

/* default_lit_len_bl / default_dis_bl :709-716 */
static int default_lit_len_bl(int i) { return i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8; }
enum { default_dis_bl = 5 };

/* Put_literal_byte :725-728 */
static inline void Put_literal_byte(deflate_ctx *c, uint8_t b) { Put_Huffman_Code(c, c->curr_descr.lit_len[b]); }

/* deflate_code_for_lz_length :757-787, extra_bits_for_lz_length(_offset) :789-803 */
static int deflate_code_for_lz_length(int length) {
  if (length <= 10) return 254 + length;
  if (length <= 18) return 265 + (length - 11) / 2;
  if (length <= 34) return 269 + (length - 19) / 4;
  if (length <= 66) return 273 + (length - 35) / 8;
  if (length <= 130) return 277 + (length - 67) / 16;
  if (length <= 257) return 281 + (length - 131) / 32;
  return 285;
}
static int extra_bits_for_lz_length(int length) {
  if (length <= 10 || length == 258) return 0;
  if (length <= 18) return 1;
  if (length <= 34) return 2;
  if (length <= 66) return 3;
  if (length <= 130) return 4;
  return 5;
}
static int extra_bits_for_lz_length_offset(int length) {
  if (length <= 18) return 11;
  if (length <= 34) return 19;
  if (length <= 66) return 35;
  if (length <= 130) return 67;
  return 131;
}

/* Deflate_code_for_LZ_distance :886-918 and the case table of Put_DL_code :841-883 */
static void distance_code(int distance, int *code, int *extra_bits, int *extra_val) {
  static const int base[14] = {5, 9, 17, 33, 65, 129, 257, 513, 1025, 2049, 4097, 8193, 16385, 32769};
  if (distance <= 4) { *code = distance - 1; *extra_bits = 0; *extra_val = 0; return; }
  for (int k = 0; k < 13; k++) {
    if (distance < base[k + 1]) {
      int half = 1 << (k + 1);                      /* 2, 4, 8, ... 8192 */
      *code = 4 + 2 * k + (distance - base[k]) / half;
      *extra_bits = k + 1;
      *extra_val = (distance - base[k]) % half;
      return;
    }
  }
  *code = 29; *extra_bits = 13; *extra_val = 0;      /* unreachable for 1..32768 */
}

/* Put_DL_code :805-884 */
static void Put_DL_code(deflate_ctx *c, int distance, int length) {
  int extra_bits, dcode, dextra, dval;
  Put_Huffman_Code(c, c->curr_descr.lit_len[deflate_code_for_lz_length(length)]);
  extra_bits = extra_bits_for_lz_length(length);
  if (extra_bits > 0)
    Put_Bits(c, (uint32_t)(length - extra_bits_for_lz_length_offset(length)) & ((1u << extra_bits) - 1), extra_bits);
  distance_code(distance, &dcode, &dextra, &dval);
  Put_Huffman_Code(c, c->curr_descr.dis[dcode]);
  if (dextra > 0) Put_Bits(c, (uint32_t)dval, dextra);
}

/* LZ buffer slices: (first, last) ring indices, inclusive; count = 0 for an Ada null slice. */
typedef struct { const LZ_atom *base; uint32_t first; uint32_t count; } lz_slice;
static inline const LZ_atom *SL(const lz_slice *s, uint32_t k) { return &s->base[s->first + k]; }

/* Get_statistics :953-976 */
static void Get_statistics(const lz_slice *lzb, Count_type *stats_lit_len, Count_type *stats_dis) {
  int dcode, de, dv;
  memset(stats_lit_len, 0, 288 * sizeof(Count_type));
  stats_lit_len[End_Of_Block] = 1;                                  /* empty_lit_len_stat :946 */
  memset(stats_dis, 0, 32 * sizeof(Count_type));
  for (uint32_t i = 0; i < lzb->count; i++) {
    const LZ_atom *a = SL(lzb, i);
    if (a->kind == plain_byte) {
      stats_lit_len[a->plain]++;
    } else {
      stats_lit_len[deflate_code_for_lz_length(a->lz_length)]++;
      distance_code(a->lz_distance, &dcode, &de, &dv);
      stats_dis[dcode]++;
    }
  }
}

/* Put_LZ_buffer :981-991 */
static void Put_LZ_buffer(deflate_ctx *c, const lz_slice *lzb) {
  for (uint32_t i = 0; i < lzb->count; i++) {
    const LZ_atom *a = SL(lzb, i);
    if (a->kind == plain_byte) Put_literal_byte(c, a->plain);
    else Put_DL_code(c, a->lz_distance, a->lz_length);
  }
}

/* Mark_new_block :999-1007 */
static void Mark_new_block(deflate_ctx *c, int last_block_for_stream) {
  if (c->block_to_finish && (c->last_block_type == bt_fixed || c->last_block_type == bt_dynamic))
    Put_Huffman_Code(c, c->curr_descr.lit_len[End_Of_Block]);
  c->block_to_finish = 1;
  Put_Bits(c, last_block_for_stream ? 1u : 0u, 1);
  c->last_block_marked = last_block_for_stream;
}

/* Expand_LZ_buffer :1010-1062 */
static void Expand_LZ_buffer(deflate_ctx *c, const lz_slice *lzb, int last_block) {
  uint8_t b1, b2;
  int64_t to_be_sent = 0;
  for (uint32_t i = 0; i < lzb->count; i++) {
    const LZ_atom *a = SL(lzb, i);
    to_be_sent += (a->kind == plain_byte) ? 1 : a->lz_length;
  }
  if (to_be_sent > 0xFFFF) {
    /* mid := (lzb'First + lzb'Last) / 2 */
    uint32_t first = lzb->first, last = lzb->first + lzb->count - 1;
    uint32_t mid = (uint32_t)(((int64_t)first + (int64_t)last) / 2);
    lz_slice lo = {lzb->base, first, mid - first + 1};
    lz_slice hi = {lzb->base, mid + 1, last - mid};
    Expand_LZ_buffer(c, &lo, 0);
    Expand_LZ_buffer(c, &hi, last_block);
    return;
  }
  b1 = (uint8_t)(to_be_sent % 256);
  b2 = (uint8_t)(to_be_sent / 256);
  Mark_new_block(c, last_block);
  c->last_block_type = bt_stored;
  Put_Bits(c, 0, 2);
  Flush_bit_buffer(c);
  Put_byte(c, b1); Put_byte(c, b2); Put_byte(c, (uint8_t)~b1); Put_byte(c, (uint8_t)~b2);
  for (uint32_t i = 0; i < lzb->count; i++) {
    const LZ_atom *a = SL(lzb, i);
    if (a->kind == plain_byte) Put_byte(c, a->plain);
    else for (int j = 1; j <= a->lz_length; j++) Put_byte(c, a->lz_expanded[j - 1]);
  }
}

/* extra_bits_for_lz_length_code :1066-1074 ; extra_bits_for_lz_distance_code :1076-1091 */
static int extra_bits_for_lz_length_code(int i) {
  return (i <= 264 || i == 285) ? 0 : (i - 261) / 4;
}
static int extra_bits_for_lz_distance_code(int i) { return i <= 3 ? 0 : (i - 2) / 2; }

/* Send_as_block :1105-1269 */
static void Send_as_block(deflate_ctx *c, const lz_slice *lzb, int last_block, int64_t g_first) {
  Deflate_Huff_Descriptors new_descr, new_descr_2;
  Count_type stats_lit_len[288], stats_lit_len_2[288], stats_dis[32], stats_dis_2[32];
  Count_type stored_format_bits = 0, fixed_format_bits = 0, dynamic_format_bits = 0,
             dynamic_format_bits_2 = 0, recycled_format_bits = 0, optimal_format_bits, cc, m1, m2;
  int stored_format_possible, recycling_possible, choice;

  /* test hook (not in the reference): the bit position in the stream at which this block's decision is taken */
  if (c->trace) c->trace(c->tr_user, ZO_TR_BITPOS, g_first, (int64_t)((c->output_size + c->OutBufIdx - 1) * 8 + (uint64_t)c->valid_bits), 0, 0);
  Get_statistics(lzb, stats_lit_len, stats_dis);                                                /* :1213 */
  Build_descriptors_stats(stats_lit_len, stats_dis, &new_descr);
  memcpy(stats_lit_len_2, stats_lit_len, sizeof stats_lit_len);
  memcpy(stats_dis_2, stats_dis, sizeof stats_dis);
  Tweak_for_better_RLE(stats_lit_len_2, 288);
  Tweak_for_better_RLE(stats_dis_2, 32);
  Build_descriptors_stats(stats_lit_len_2, stats_dis_2, &new_descr_2);
  stored_format_possible = 1;                                                                     /* :1222 */
  for (int i = code_for_max_expand + 1; i <= 287; i++) if (stats_lit_len[i] != 0) stored_format_possible = 0;
  recycling_possible = c->last_block_type == bt_fixed ||                                          /* :1223-1226 */
                       (c->last_block_type == bt_dynamic && Recyclable(&c->curr_descr, &new_descr));

  /* Compute_sizes_of_variants :1147-1209 */
  for (int i = 0; i <= 255; i++) {
    cc = stats_lit_len[i];
    stored_format_bits += 8 * cc;
    fixed_format_bits += (Count_type)default_lit_len_bl(i) * cc;
    dynamic_format_bits += (Count_type)new_descr.lit_len[i].bit_length * cc;
    dynamic_format_bits_2 += (Count_type)new_descr_2.lit_len[i].bit_length * cc;
    recycled_format_bits += (Count_type)c->curr_descr.lit_len[i].bit_length * cc;
  }
  if (stored_format_possible)
    for (uint32_t i = 0; i < lzb->count; i++) {
      const LZ_atom *a = SL(lzb, i);
      if (a->kind == distance_length) stored_format_bits += 8 * (Count_type)a->lz_length;
    }
  for (int i = 257; i <= 285; i++) {
    int extra = extra_bits_for_lz_length_code(i);
    cc = stats_lit_len[i];
    fixed_format_bits += (Count_type)(default_lit_len_bl(i) + extra) * cc;
    dynamic_format_bits += (Count_type)(new_descr.lit_len[i].bit_length + extra) * cc;
    dynamic_format_bits_2 += (Count_type)(new_descr_2.lit_len[i].bit_length + extra) * cc;
    recycled_format_bits += (Count_type)(c->curr_descr.lit_len[i].bit_length + extra) * cc;
  }
  for (int i = 0; i <= 29; i++) {
    int extra = extra_bits_for_lz_distance_code(i);
    cc = stats_dis[i];
    fixed_format_bits += (Count_type)(default_dis_bl + extra) * cc;
    dynamic_format_bits += (Count_type)(new_descr.dis[i].bit_length + extra) * cc;
    dynamic_format_bits_2 += (Count_type)(new_descr_2.dis[i].bit_length + extra) * cc;
    recycled_format_bits += (Count_type)(c->curr_descr.dis[i].bit_length + extra) * cc;
  }
  stored_format_bits += (1 + (stored_format_bits / 8) / 65535) * 5 * 8;                           /* :1193-1196 */
  cc = 1;
  if (c->block_to_finish && (c->last_block_type == bt_fixed || c->last_block_type == bt_dynamic))
    cc += (Count_type)c->curr_descr.lit_len[End_Of_Block].bit_length;
  stored_format_bits += cc;
  fixed_format_bits += cc + 2;
  dynamic_format_bits += cc + 2;
  dynamic_format_bits_2 += cc + 2;
  Put_Compression_Structure(c, &new_descr, 1, &dynamic_format_bits);
  Put_Compression_Structure(c, &new_descr_2, 1, &dynamic_format_bits_2);

  if (!stored_format_possible) stored_format_bits = Count_type_Last;                              /* :1228-1233 */
  if (!recycling_possible) recycled_format_bits = Count_type_Last;
  m1 = stored_format_bits < fixed_format_bits ? stored_format_bits : fixed_format_bits;
  m2 = dynamic_format_bits < dynamic_format_bits_2 ? dynamic_format_bits : dynamic_format_bits_2;
  m2 = m2 < recycled_format_bits ? m2 : recycled_format_bits;
  optimal_format_bits = m1 < m2 ? m1 : m2;

  if (fixed_format_bits == optimal_format_bits) {                                                 /* :1243-1268 */
    choice = 1;
    /* Send_fixed_block :1108-1121 */
    if (c->last_block_type != bt_fixed) {
      Mark_new_block(c, last_block);
      c->curr_descr = c->Deflate_fixed_descriptors;
      Put_Bits(c, 1, 2);
      c->last_block_type = bt_fixed;
    }
    Put_LZ_buffer(c, lzb);
  } else if (dynamic_format_bits == optimal_format_bits || dynamic_format_bits_2 == optimal_format_bits) {
    Count_type dummy = 0;
    choice = dynamic_format_bits == optimal_format_bits ? 2 : 3;
    /* Send_dynamic_block :1126-1135 */
    Mark_new_block(c, last_block);
    c->curr_descr = (choice == 2) ? new_descr : new_descr_2;
    Prepare_Huffman_Codes(&c->curr_descr);
    Put_Bits(c, 2, 2);
    Put_Compression_Structure(c, &c->curr_descr, 0, &dummy);
    Put_LZ_buffer(c, lzb);
    c->last_block_type = bt_dynamic;
  } else if (recycled_format_bits == optimal_format_bits) {
    choice = 4;
    Put_LZ_buffer(c, lzb);
  } else {
    choice = 0;
    Expand_LZ_buffer(c, lzb, last_block);
  }
  if (c->trace) c->trace(c->tr_user, ZO_TR_BLOCK, g_first, lzb->count, choice, (int64_t)optimal_format_bits);
}

/* Build_descriptors (lzb) :1319-1325 */
static void Build_descriptors_lzb(const lz_slice *lzb, Deflate_Huff_Descriptors *d) {
  Count_type stats_lit_len[288], stats_dis[32];
  Get_statistics(lzb, stats_lit_len, stats_dis);
  Build_descriptors_stats(stats_lit_len, stats_dis, d);
}

static int max_choice(int method) { return method == ZO_DEFLATE_1 ? 1 : method == ZO_DEFLATE_2 ? 2 : 3; }  /* :1310-1311 */

/* Scan_and_send_from_main_buffer :1327-1408 */
static void Scan_and_send_from_main_buffer(deflate_ctx *c, uint32_t from, uint32_t to, int last_flush) {
  Deflate_Huff_Descriptors initial_hd, sliding_hd;
  uint32_t start, slide_mid, send_from;
  int sliding_hd_computed;
  lz_slice sl;
  const int64_t g0 = (int64_t)c->atoms_flushed_before - (int64_t)from;   /* trace: global index of ring slot 0 */
  if (LZ_IDX(to - from) < slider_max) {
    sl.base = c->lz_buffer; sl.first = from; sl.count = to - from + 1;
    Send_as_block(c, &sl, last_flush, g0 + from);
    return;
  }
  if (c->past_lz_data) start = LZ_IDX(from - half_slider_size); else start = from;
  if (start > from) {                                                               /* :1343-1354 */
    static LZ_atom copy[slider_size];
    uint32_t copy_from = start;
    for (int i = 0; i <= slider_max; i++) { copy[i] = c->lz_buffer[copy_from]; copy_from = LZ_IDX(copy_from + 1); }
    sl.base = copy; sl.first = 0; sl.count = slider_size;
    Build_descriptors_lzb(&sl, &initial_hd);
  } else {
    sl.base = c->lz_buffer; sl.first = start; sl.count = slider_size;
    Build_descriptors_lzb(&sl, &initial_hd);
  }
  send_from = from;
  slide_mid = from + min_step;
  while ((int64_t)slide_mid + half_slider_size < (int64_t)to) {                     /* Scan_LZ_data */
    sliding_hd_computed = 0;
    for (int level = 1; level <= 3; level++) {                                      /* Browse_step_level */
      if (level > max_choice(c->method)) break;
      if (LZ_IDX(slide_mid - from) % step_choice[level - 1].slider_step == 0) {
        if (!sliding_hd_computed) {
          /* lz_buffer (slide_mid - half_slider_size .. slide_mid + half_slider_size): modular
             bounds; a lower bound that wraps above the upper bound is an Ada NULL SLICE. */
          uint32_t lo = LZ_IDX(slide_mid - half_slider_size), hi = LZ_IDX(slide_mid + half_slider_size);
          sl.base = c->lz_buffer; sl.first = lo; sl.count = (lo > hi) ? 0 : hi - lo + 1;
          Build_descriptors_lzb(&sl, &sliding_hd);
          sliding_hd_computed = 1;
        }
        if (!Similar(c, &initial_hd, &sliding_hd, step_choice[level - 1].cutting_threshold, g0 + slide_mid)) {
          if (c->trace) c->trace(c->tr_user, ZO_TR_CUT, g0 + slide_mid, level, 0, 0);
          sl.base = c->lz_buffer; sl.first = send_from; sl.count = slide_mid - send_from;
          Send_as_block(c, &sl, 0, g0 + send_from);
          send_from = slide_mid;
          initial_hd = sliding_hd;
          break;
        }
      }
    }
    if ((int64_t)slide_mid + min_step + half_slider_size >= (int64_t)to) break;
    slide_mid += min_step;
  }
  if (send_from <= to) {
    sl.base = c->lz_buffer; sl.first = send_from; sl.count = to - send_from + 1;
    Send_as_block(c, &sl, last_flush, g0 + send_from);
  }
}

/* Flush_half_buffer :1410-1422 */
static void Flush_half_buffer(deflate_ctx *c, int last_flush) {
  const uint32_t last_idx = LZ_IDX(c->lz_buffer_index - 1);
  const uint32_t n_div_2 = LZ_buffer_size / 2;
  if (last_idx < n_div_2) Scan_and_send_from_main_buffer(c, 0, last_idx, last_flush);
  else Scan_and_send_from_main_buffer(c, n_div_2, last_idx, last_flush);
  c->past_lz_data = 1;
  c->atoms_flushed_before += n_div_2;
}

/* Push :1424-1432 */
static inline void Push(deflate_ctx *c, const LZ_atom *a) {
  c->lz_buffer[c->lz_buffer_index] = *a;
  c->lz_buffer_index = LZ_IDX(c->lz_buffer_index + 1);
  if (LZ_IDX(c->lz_buffer_index * 2) == 0) Flush_half_buffer(c, 0);
}

/* Put_or_delay_literal_byte :1434-1443 */
static inline void Put_or_delay_literal_byte(deflate_ctx *c, uint8_t b) {
  if (c->method == ZO_DEFLATE_FIXED) { Put_literal_byte(c, b); return; }
  LZ_atom a; memset(&a, 0, sizeof a);
  a.kind = plain_byte; a.plain = b; a.lz_expanded[0] = b;
  Push(c, &a);
}

/* Put_or_delay_DL_code :1445-1454 */
static inline void Put_or_delay_DL_code(deflate_ctx *c, int distance, int length, const uint8_t *expand) {
  if (c->method == ZO_DEFLATE_FIXED) { Put_DL_code(c, distance, length); return; }
  LZ_atom a; memset(&a, 0, sizeof a);
  a.kind = distance_length; a.lz_distance = distance; a.lz_length = length;
  memcpy(a.lz_expanded, expand, max_expand);
  Push(c, &a);
}

/* ---- Encode :1460-1636 ---- */

/* Read_byte :1467-1495 */
static uint8_t Read_byte(void *u) {
  deflate_ctx *c = (deflate_ctx *)u;
  uint8_t b = c->InBuf[c->InBufIdx];
  c->InBufIdx++;
  c->CRC = zo_crc32_update(c->CRC, &b, 1);
  c->Bytes_in++;
  if (c->feedback != NULL) {
    int user_aborting = 0;
    if (c->Bytes_in == 1) user_aborting = c->feedback(0, 0, c->fb_user);
    if (c->feedback_milestone > 0 &&
        ((c->Bytes_in - 1) % c->feedback_milestone == 0 || c->Bytes_in == c->input_size)) {
      if (c->input_size_known) {
        int PctDone = (int)((100.0f * (float)c->Bytes_in) / (float)c->input_size);
        user_aborting = c->feedback(PctDone, 0, c->fb_user);
      } else {
        user_aborting = c->feedback(0, 0, c->fb_user);
      }
      if (user_aborting) { c->escape_code = ZO_ABORTED; longjmp(c->escape, 1); }   /* raise User_abort */
    }
  }
  return b;
}

/* More_bytes :1497-1503 */
static int More_bytes(void *u) {
  deflate_ctx *c = (deflate_ctx *)u;
  if (c->InBufIdx > c->MaxInBufIdx) Read_Block(c);
  return !c->InputEoF;
}

/* LZ77_emits_DL_code :1518-1554 */
static void LZ77_emits_DL_code(void *u, int distance, int length) {
  deflate_ctx *c = (deflate_ctx *)u;
  uint8_t b, expand[max_expand];
  uint32_t copy_start;
  int ie = 1;
  memset(expand, 0, sizeof expand);
  if (distance == 32768) copy_start = c->R; else copy_start = (c->R - (uint32_t)distance) & 32767;
  for (uint32_t K = 0; K <= (uint32_t)(length - 1); K++) {
    b = c->Text_Buf[(copy_start + K) & 32767];
    c->Text_Buf[c->R] = b;
    c->R = (c->R + 1) & 32767;
    if (ie <= max_expand) { expand[ie - 1] = b; ie++; }
  }
  if (distance >= 1 && distance <= 32768 && length >= 3 && length <= 258) {
    Put_or_delay_DL_code(c, distance, length, expand);
  } else {
    for (uint32_t K = 0; K <= (uint32_t)(length - 1); K++)
      Put_or_delay_literal_byte(c, c->Text_Buf[(copy_start + K) & 32767]);
  }
}

/* LZ77_emits_literal_byte :1556-1561 */
static void LZ77_emits_literal_byte(void *u, uint8_t b) {
  deflate_ctx *c = (deflate_ctx *)u;
  c->Text_Buf[c->R] = b;
  c->R = (c->R + 1) & 32767;
  Put_or_delay_literal_byte(c, b);
}

/* LZ77_choice :1573-1579 */
static int LZ77_choice(int method) {
  switch (method) {
    case ZO_DEFLATE_FIXED: return 4;
    case ZO_DEFLATE_0: return LZ_NO_LZ77;
    case ZO_DEFLATE_1: return 6;
    case ZO_DEFLATE_2: return 8;
    case ZO_DEFLATE_3: return 10;
    default: return -1;                 /* Deflate_R (LZ77.Rich) is out of scope */
  }
}

/* Encode :1593-1636 */
static void Encode(deflate_ctx *c) {
  zo_lz_io io = {Read_byte, More_bytes, LZ77_emits_literal_byte, LZ77_emits_DL_code, c};
  Read_Block(c);
  c->R = 32768 - 258;
  if (c->input_size_known) c->feedback_milestone = c->input_size / feedback_steps;
  if (c->method == ZO_DEFLATE_FIXED) { Put_Bits(c, 1, 1); Put_Bits(c, 1, 2); }

  if (c->replay == NULL) {
    LZ77_Encode(&io, LZ77_choice(c->method));                                       /* My_LZ77 :1611 */
  } else {
    /* as LZ77_from_Dump_File (lz77.adb:2148-2179): consume the stream, then replay tokens */
    while (More_bytes(c)) (void)Read_byte(c);
    for (uint64_t i = 0; i < c->nreplay; i++) {
      uint32_t t = c->replay[i];
      if (t & ZO_TOKEN_MATCH) LZ77_emits_DL_code(c, (int)(t & 0xFFFF), (int)((t >> 16) & 0x1FF));
      else LZ77_emits_literal_byte(c, (uint8_t)t);
    }
  }

  if (c->method == ZO_DEFLATE_FIXED) {
    Put_Huffman_Code(c, c->curr_descr.lit_len[End_Of_Block]);
  } else {
    if (LZ_IDX(c->lz_buffer_index * 2) == 0) {
      if (c->block_to_finish && (c->last_block_type == bt_fixed || c->last_block_type == bt_dynamic))
        Put_Huffman_Code(c, c->curr_descr.lit_len[End_Of_Block]);
    } else {
      Flush_half_buffer(c, 1);
      if (c->last_block_type == bt_fixed || c->last_block_type == bt_dynamic)
        Put_Huffman_Code(c, c->curr_descr.lit_len[End_Of_Block]);
    }
    if (!c->last_block_marked) {
      Put_Bits(c, 1, 1);
      Put_Bits(c, 1, 2);
      c->curr_descr = c->Deflate_fixed_descriptors;
      Put_Huffman_Code(c, c->curr_descr.lit_len[End_Of_Block]);
    }
  }
}

/* body :1644-1679 */
static int deflate_body(const uint8_t *in, uint64_t n, int method, uint8_t *out, uint64_t cap, uint64_t *out_len,
                        uint32_t *crc_inout, zo_feedback_fn fb, void *fb_user, zo_trace_fn tr, void *tr_user,
                        const uint32_t *replay, uint64_t nreplay) {
  deflate_ctx *c;
  int rc = ZO_OK;
  int bl_ll[288], bl_d[32];
  uint64_t calibration;
  if (LZ77_choice(method) < 0) return ZO_EINVAL;
  c = (deflate_ctx *)calloc(1, sizeof *c);
  if (!c) return ZO_ENOMEM;
  c->in = in; c->in_size = n; c->in_pos = 0;
  c->input_size_known = 1; c->input_size = n;                      /* zip-create.adb:256-257: always known */
  c->method = method; c->feedback = fb; c->fb_user = fb_user; c->trace = tr; c->tr_user = tr_user;
  c->CRC = crc_inout ? *crc_inout : zo_crc32_init();
  c->out = out; c->out_cap = cap; c->output_size = 0;
  c->replay = replay; c->nreplay = nreplay;
  /* Deflate_fixed_descriptors :718-719 ; curr_descr :722 */
  for (int i = 0; i < 288; i++) bl_ll[i] = default_lit_len_bl(i);
  for (int i = 0; i < 32; i++) bl_d[i] = default_dis_bl;
  Build_descriptors_bl(bl_ll, bl_d, &c->Deflate_fixed_descriptors);
  Prepare_Huffman_Codes(&c->Deflate_fixed_descriptors);
  c->curr_descr = c->Deflate_fixed_descriptors;
  c->block_to_finish = 0; c->last_block_marked = 0; c->last_block_type = bt_reserved;
  c->lz_buffer_index = 0; c->past_lz_data = 0;
  /* Allocate_Buffers zip-compress.adb:430-445 */
  calibration = n > 8 ? n : 8;
  if (calibration > default_byte_IO_buffer_size) calibration = default_byte_IO_buffer_size;
  c->InBuf_len = calibration;
  c->InBuf = (uint8_t *)malloc(calibration + 1);
  c->OutBuf = (uint8_t *)malloc(default_byte_IO_buffer_size + 1);
  c->OutBufIdx = 1;
  c->lz_buffer = (LZ_atom *)calloc(LZ_buffer_size, sizeof(LZ_atom));
  if (!c->InBuf || !c->OutBuf || !c->lz_buffer) { rc = ZO_ENOMEM; goto done; }
  if (setjmp(c->escape) == 0) {
    Encode(c);
    Flush_bit_buffer(c);
    Flush_byte_buffer(c);
    rc = ZO_OK;
  } else {
    rc = c->escape_code;                                             /* :1667-1669 */
  }
  if (out_len) *out_len = c->output_size;
  if (crc_inout) *crc_inout = c->CRC;
done:
  free(c->InBuf); free(c->OutBuf); free(c->lz_buffer); free(c);
  return rc;
}

int zo_deflate(const uint8_t *in, uint64_t n, int method, uint8_t *out, uint64_t cap, uint64_t *out_len,
               uint32_t *crc_inout, zo_feedback_fn fb, void *fb_user, zo_trace_fn tr, void *tr_user) {
  return deflate_body(in, n, method, out, cap, out_len, crc_inout, fb, fb_user, tr, tr_user, NULL, 0);
}

int zo_deflate_from_tokens(const uint8_t *in, uint64_t n, const uint32_t *tokens, uint64_t ntok, int method,
                           uint8_t *out, uint64_t cap, uint64_t *out_len, zo_trace_fn tr, void *tr_user) {
  static const uint32_t none = 0;
  return deflate_body(in, n, method, out, cap, out_len, NULL, NULL, NULL, tr, tr_user, tokens ? tokens : &none, ntok);
}

/* ------------------------------------------------------------------------- */
/*  Zip.Compress.Compress_Data, single method, clear  --  zip-compress.adb:142-241 */
/* ------------------------------------------------------------------------- */
int zo_compress_data(const uint8_t *in, uint64_t n, int method, uint8_t *out, uint64_t cap, uint64_t *out_len,
                     uint32_t *crc_out, uint16_t *zip_type) {
  uint32_t CRC = zo_crc32_init();                                    /* :144 */
  int rc;
  if (method == ZO_STORE) {                                          /* Store_data :107-141 */
    if (cap < n) return ZO_EINVAL;
    memcpy(out, in, n);
    CRC = zo_crc32_update(CRC, in, n);
    *out_len = n; *zip_type = 0; *crc_out = zo_crc32_final(CRC);
    return ZO_OK;
  }
  if (method >= ZO_BZIP2_1 && method <= ZO_BZIP2_3) {                /* :204-209 */
    rc = zo_bzip2(in, n, method, out, cap, out_len, &CRC);
    *zip_type = 12;                                                  /* bzip2_code, zip.ads:502 */
  } else if (method >= ZO_LZMA_0 && method <= ZO_LZMA_3) {           /* :211-216 */
    rc = zo_lzma(in, n, method, out, cap, out_len, &CRC);
    *zip_type = 14;                                                  /* lzma_code, zip.ads:503 */
  } else {
    rc = zo_deflate(in, n, method, out, cap, out_len, &CRC, NULL, NULL, NULL, NULL);   /* :197-202 */
    *zip_type = 8;
  }
  if (rc < 0 || rc == ZO_ABORTED) return rc;
  CRC = zo_crc32_final(CRC);                                         /* :218 */
  if (rc == ZO_INEFFICIENT) {                                        /* :224-237 */
    if (cap < n) return ZO_EINVAL;
    CRC = zo_crc32_init();
    memcpy(out, in, n);
    CRC = zo_crc32_update(CRC, in, n);
    CRC = zo_crc32_final(CRC);
    *out_len = n; *zip_type = 0;
  }
  *crc_out = CRC;
  return ZO_OK;
}
