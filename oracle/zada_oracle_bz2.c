/* zada_oracle_bz2.c -- CPU restatement of zip-ada's BZip2 encoder.
 *
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it; the product never does.
 *
 * Follows zip_lib/bzip2-encoding.adb (Encode :87-1431), zip_lib/bzip2-buffers.adb:8-45 (MSB-first bit buffer),
 * zip_lib/bzip2.adb (CRC :13-122), zip_lib/data_segmentation.adb:39-105, zip_lib/zip-compress-bzip2_e.adb:44-157
 * and reuses zo_llhc / zo_prepare_codes of zada_oracle.c (huffman-encoding*.adb).
 *
 * PARITY UNPINNED against the Ada build, like the Deflate half (no GNAT here).  What pins it: every stream must
 * decompress with libbz2 (Python bz2) to the input.  Two places depend on things outside /root/reference:
 *  1. Ranking_Sort (:566-570) is an instance of Ada.Containers.Generic_Constrained_Array_Sort with a key-only "<";
 *     the order of equal keys is whatever GNAT's run-time body does.  That body (libgnat a-cgcaso.adb, "adapted from
 *     GNAT.Heap_Sort_G") is not in this image; gnat_heap_sort() below restates its published algorithm from memory:
 *     Floyd's heap sort, 1-based, sift-down along the larger sons followed by a sift-up of the saved element.
 *  2. Data_Segmentation uses Log on a `digits 15` type; GNAT maps it to the C library's log().  The oracle calls libm.
 * The BWT itself is a total order (ties by offset, :245-246), so any correct rotation sort gives the reference's result;
 * the oracle sorts with cyclic prefix doubling instead of the reference's comparison sort (the reference's is far slower
 * on redundant data, see its own To-do note :37-40).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "zada_oracle.h"

/* ---- BZip2.CRC (bzip2.adb): MSB-first CRC-32, polynomial 0x04C11DB7 ---- */
static uint32_t bz_crc_table[256];
static int bz_crc_ready;
static void bz_crc_prepare(void) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i << 24;
    for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : c << 1;
    bz_crc_table[i] = c;
  }
  bz_crc_ready = 1;
}
static inline uint32_t bz_crc_update(uint32_t crc, uint8_t val) { return bz_crc_table[0xFF & ((crc >> 24) ^ val)] ^ (crc << 8); }

uint32_t zo_bz2_crc(const uint8_t *buf, uint64_t n) {          /* Init / Update / Final */
  uint32_t c = 0xFFFFFFFFu;
  if (!bz_crc_ready) bz_crc_prepare();
  for (uint64_t i = 0; i < n; i++) c = bz_crc_update(c, buf[i]);
  return ~c;
}

/* ---- BZip2.Buffers ---- */
typedef struct {
  uint8_t buffer;            /* partial byte */
  int bit_index;             /* 7 .. 0, next bit to fill */
  uint8_t *destination_data;
  int64_t destination_index; /* bytes stored */
  int64_t cap;
  int overflow;
} Bit_Buffer_Type;

static void Flush_Bit_Buffer(Bit_Buffer_Type *b) {
  if (b->destination_index < b->cap) b->destination_data[b->destination_index] = b->buffer; else b->overflow = 1;
  b->destination_index++;
  b->buffer = 0;
  b->bit_index = 7;
}
static void Put_Bits(Bit_Buffer_Type *b, uint32_t data, int amount) {
  for (int count = amount; count >= 1; count--) {
    if (data & (1u << (count - 1))) b->buffer |= (uint8_t)(1u << b->bit_index);
    if (b->bit_index == 0) Flush_Bit_Buffer(b); else b->bit_index--;
  }
}
static void Put_String(Bit_Buffer_Type *b, const char *s, int n) { for (int i = 0; i < n; i++) Put_Bits(b, (uint8_t)s[i], 8); }

/* ---- constants, bzip2.ads private part ---- */
enum { run_a = 0, run_b = 1, max_alphabet_size = 258, max_entropy_coders = 6, min_entropy_coders = 2, group_size = 50,
       sub_block_size = 100000 };
static const char block_header_magic[6] = {'1', 'A', 'Y', '&', 'S', 'Y'};
static const char stream_footer_magic[6] = {0x17, 'r', 'E', '8', 'P', (char)0x90};

enum { block_100k = 0, block_400k = 1, block_900k = 2 };

/* ---- GNAT's Ada.Containers.Generic_Constrained_Array_Sort (see the header note) ---- */
typedef struct { int32_t key; int32_t index; } Pair;
static void gnat_heap_sort(Pair *a1, int64_t n) {   /* a1[1 .. n] */
  int64_t Max = n;
  Pair Temp;
#define SIFT(S)                                                                      \
  do {                                                                               \
    int64_t C = (S), Son;                                                            \
    for (;;) {                                                                       \
      Son = 2 * C;                                                                   \
      if (Son > Max) break;                                                          \
      if (Son < Max && a1[Son].key < a1[Son + 1].key) Son++;                         \
      a1[C] = a1[Son];                                                               \
      C = Son;                                                                       \
    }                                                                                \
    while (C != (S)) {                                                               \
      int64_t Father = C / 2;                                                        \
      if (a1[Father].key < Temp.key) { a1[C] = a1[Father]; C = Father; } else break; \
    }                                                                                \
    a1[C] = Temp;                                                                    \
  } while (0)
  for (int64_t J = Max / 2; J >= 1; J--) { Temp = a1[J]; SIFT(J); }
  while (Max > 1) {
    Temp = a1[Max];
    a1[Max] = a1[1];
    Max--;
    SIFT(1);
  }
#undef SIFT
}

/* ---- rotation sort: sa[i] = 0-based start of the i-th smallest rotation, cls[s] = class of the rotation at s ---- */
static int rotation_sort(const uint8_t *t, int32_t n, int32_t *sa, int32_t *cls) {
  int32_t *sa2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *cls2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 256 ? n : 256));
  int32_t classes;
  if (!sa2 || !cls2 || !cnt) { free(sa2); free(cls2); free(cnt); return -1; }
  memset(cnt, 0, sizeof(int32_t) * 256);
  for (int32_t i = 0; i < n; i++) cnt[t[i]]++;
  for (int i = 1; i < 256; i++) cnt[i] += cnt[i - 1];
  for (int32_t i = n - 1; i >= 0; i--) sa[--cnt[t[i]]] = i;
  cls[sa[0]] = 0; classes = 1;
  for (int32_t i = 1; i < n; i++) { if (t[sa[i]] != t[sa[i - 1]]) classes++; cls[sa[i]] = classes - 1; }
  for (int64_t h = 1; h < n && classes < n; h <<= 1) {
    for (int32_t i = 0; i < n; i++) { int64_t s = (int64_t)sa[i] - h; if (s < 0) s += n; sa2[i] = (int32_t)s; }
    memset(cnt, 0, sizeof(int32_t) * (size_t)classes);
    for (int32_t i = 0; i < n; i++) cnt[cls[sa2[i]]]++;
    for (int32_t i = 1; i < classes; i++) cnt[i] += cnt[i - 1];
    for (int32_t i = n - 1; i >= 0; i--) sa[--cnt[cls[sa2[i]]]] = sa2[i];
    cls2[sa[0]] = 0; classes = 1;
    for (int32_t i = 1; i < n; i++) {
      int64_t a = sa[i] + h, b = sa[i - 1] + h;
      if (a >= n) a -= n;
      if (b >= n) b -= n;
      if (cls[sa[i]] != cls[sa[i - 1]] || cls[a] != cls[b]) classes++;
      cls2[sa[i]] = classes - 1;
    }
    memcpy(cls, cls2, sizeof(int32_t) * (size_t)n);
  }
  free(sa2); free(cls2); free(cnt);
  return 0;
}

/* ---- Encode_Block :148-1134 ---- */
typedef struct { int bit_length; int32_t code; } Length_Code_Pair;

typedef struct {
  int option;
  int32_t block_capacity;
  /* stage outputs (also what zo_bz2_block hands to the tests) */
  uint8_t *rle_1_data; int32_t rle_1_block_size;
  uint32_t block_crc;
  int in_use[256];
  uint8_t *bwt_data; int32_t bwt_index;
  uint16_t *mtf_data; int32_t mtf_last;
  int normal_symbols_in_use, last_symbol_in_use, EOB;
  Length_Code_Pair descr[max_entropy_coders + 1][max_alphabet_size];
  int entropy_coder_count;
  int32_t selector_count;
  uint8_t *selector;                 /* 1-based */
  int max_code_len;
  /* Multiple_Entropy_Coders state */
  int low_cluster_usage;
  int defector_groups;
  int best_sample_width;
  Pair *ranking;                     /* 1-based */
} block_ctx;

static int RLE_1(block_ctx *k, const uint8_t *raw, int32_t n) {                /* :167-213 */
  uint8_t b_prev = 0;
  int run = 0, start = 1;
  uint32_t crc = 0xFFFFFFFFu;
  k->rle_1_block_size = 0;
  memset(k->in_use, 0, sizeof k->in_use);
  /* allocation: the reference reserves block_capacity * 5 / 4; any fitting size will do */
  k->rle_1_data = (uint8_t *)malloc((size_t)n + (size_t)n / 4 + 8);
  if (!k->rle_1_data) return -1;
#define STORE(x) do { k->rle_1_data[k->rle_1_block_size++] = (uint8_t)(x); k->in_use[(uint8_t)(x)] = 1; } while (0)
#define STORE_RUN() do { for (int count = 1; count <= (run < 4 ? run : 4); count++) STORE(b_prev); if (run >= 4) STORE(run - 4); run = 1; } while (0)
  for (int32_t i = 0; i < n; i++) {
    const uint8_t b = raw[i];
    crc = bz_crc_update(crc, b);
    if (start || b != b_prev) { STORE_RUN(); start = 0; }
    else if (run == 259) { STORE_RUN(); }
    else run++;
    b_prev = b;
  }
  STORE_RUN();
#undef STORE_RUN
#undef STORE
  k->block_crc = crc;
  return 0;
}

static int BWT(block_ctx *k) {                                                    /* :222-300 */
  const int32_t n = k->rle_1_block_size;
  int32_t *sa, *cls;
  k->bwt_index = 0;
  k->bwt_data = (uint8_t *)malloc((size_t)n + 1);
  if (!k->bwt_data) return -1;
  if (n == 0) return 0;
  sa = (int32_t *)malloc(sizeof(int32_t) * (size_t)n); cls = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  if (!sa || !cls || rotation_sort(k->rle_1_data, n, sa, cls) < 0) { free(sa); free(cls); return -1; }
  /* offset o of the reference is the rotation starting at (n - o) mod n; equal rotations are ordered by offset, so the
   * original message (offset 0) is the first row of its class (:245-246, :272-275) */
  {
    int32_t first = -1;
    for (int32_t i = 0; i < n; i++) {
      const int32_t s = sa[i];
      k->bwt_data[i] = k->rle_1_data[s == 0 ? n - 1 : s - 1];
      if (first < 0 && cls[s] == cls[0]) first = i;
    }
    k->bwt_index = first;
  }
  free(sa); free(cls);
  return 0;
}

static int MTF_and_RLE_2(block_ctx *k) {                                          /* :320-412 */
  uint8_t unseq_to_seq[256];
  uint8_t mtf_symbol[256];
  int32_t run = 0;
  k->normal_symbols_in_use = 0;
  memset(unseq_to_seq, 0, sizeof unseq_to_seq);
  for (int i = 0; i < 256; i++) if (k->in_use[i]) unseq_to_seq[i] = (uint8_t)k->normal_symbols_in_use++;
  k->last_symbol_in_use = k->normal_symbols_in_use + 3 - 1 - 1;
  k->EOB = k->last_symbol_in_use;
  k->mtf_data = (uint16_t *)malloc(sizeof(uint16_t) * (2 * (size_t)k->rle_1_block_size + 2));   /* 1-based */
  if (!k->mtf_data) return -1;
  k->mtf_last = 0;
#define STORE(a) do { k->mtf_data[++k->mtf_last] = (uint16_t)(a); } while (0)
#define STORE_RUN() do { if (run > 0) { uint32_t rc = (uint32_t)(run + 1); do { STORE(rc & 1); rc >>= 1; } while (rc >= 2); run = 0; } } while (0)
  for (int i = 0; i < 256; i++) mtf_symbol[i] = (uint8_t)i;
  for (int32_t p = 0; p < k->rle_1_block_size; p++) {
    const uint8_t bt_seq = unseq_to_seq[k->bwt_data[p]];
    int idx = 0;
    while (mtf_symbol[idx] != bt_seq) idx++;
    for (int i = idx; i >= 1; i--) mtf_symbol[i] = mtf_symbol[i - 1];
    mtf_symbol[0] = bt_seq;
    if (idx == 0) run++;
    else { STORE_RUN(); STORE(1 + idx); }
  }
  STORE_RUN();
  STORE(k->EOB);
#undef STORE_RUN
#undef STORE
  return 0;
}

/* -- Entropy_Calculations :418-1010 -- */
static void Avoid_Zeros(uint32_t *freq, int n) {                                  /* :436-460 */
  int zeroes = 0;
  for (int i = 0; i < n; i++) if (freq[i] == 0) zeroes++;
  if (zeroes == 0) return;
  if (zeroes <= 100) { for (int i = 0; i < n; i++) if (freq[i] < 1) freq[i] = 1; }
  else { for (int i = 0; i < n; i++) freq[i] = freq[i] == 0 ? 1 : freq[i] * 2; }
}

static void Define_Descriptor(block_ctx *k, uint32_t *freq, int des) {            /* :497-517 */
  const int n = k->last_symbol_in_use + 1;
  uint64_t f64[max_alphabet_size];
  int len[max_alphabet_size];
  int32_t codes[max_alphabet_size];
  Avoid_Zeros(freq, n);
  for (int i = 0; i < n; i++) f64[i] = freq[i];
  zo_llhc(f64, n, k->max_code_len, len);
  zo_prepare_codes(len, n, k->max_code_len, 0, codes);
  for (int i = 0; i < n; i++) { k->descr[des][i].bit_length = len[i]; k->descr[des][i].code = codes[i]; }
}

static void Initial_Clustering_by_Rank(block_ctx *k, const int *attr, int na) {   /* :574-591 */
  const int32_t ns = k->selector_count;
  int32_t low = 1, high;
  for (int a32 = 1; a32 <= na; a32++) {
    high = (int32_t)((int64_t)a32 * ns / na);
    for (int32_t i = low; i <= high; i++) k->selector[k->ranking[i].index] = (uint8_t)attr[a32 - 1];
    low = 1 + high;
  }
}

static void Initial_Clustering_Ranking_Method(block_ctx *k, int sample_width) {   /* :555-635 */
  int pos_countdown = group_size;
  int32_t sel_idx = 1, key = 0;
  const int last_symbol_sampled = (k->EOB - 1 < run_a + sample_width - 1) ? k->EOB - 1 : run_a + sample_width - 1;
  static const int a2[] = {2, 1}, a3[] = {3, 1, 2}, a4[] = {4, 2, 1, 3}, a5[] = {5, 3, 1, 2, 4}, a6[] = {6, 4, 2, 1, 3, 5};
  for (int32_t m = 1; m <= k->mtf_last; m++) {
    const int symbol = k->mtf_data[m];
    if (symbol >= run_a && symbol <= last_symbol_sampled) key++;
    if (--pos_countdown == 0) {
      k->ranking[sel_idx].key = key; k->ranking[sel_idx].index = sel_idx;
      pos_countdown = group_size; sel_idx++; key = 0;
    }
  }
  if (pos_countdown < group_size) { k->ranking[sel_idx].key = key; k->ranking[sel_idx].index = sel_idx; }
  gnat_heap_sort(k->ranking, k->selector_count);
  switch (k->entropy_coder_count) {
    case 2: Initial_Clustering_by_Rank(k, a2, 2); break;
    case 3: Initial_Clustering_by_Rank(k, a3, 3); break;
    case 4: Initial_Clustering_by_Rank(k, a4, 4); break;
    case 5: Initial_Clustering_by_Rank(k, a5, 5); break;
    default: Initial_Clustering_by_Rank(k, a6, 6); break;
  }
}

static void Define_Descriptors(block_ctx *k) {                                    /* :637-660 */
  uint32_t freq_cluster[max_entropy_coders + 1][max_alphabet_size];
  int pos_countdown = group_size;
  int32_t selector_idx = 1;
  int cluster = k->selector[1];
  memset(freq_cluster, 0, sizeof freq_cluster);
  for (int32_t m = 1; m <= k->mtf_last; m++) {
    freq_cluster[cluster][k->mtf_data[m]]++;
    if (--pos_countdown == 0 && m < k->mtf_last) { pos_countdown = group_size; selector_idx++; cluster = k->selector[selector_idx]; }
  }
  for (int cl = 1; cl <= k->entropy_coder_count; cl++) Define_Descriptor(k, freq_cluster[cl], cl);
}

static void Simulate_Entropy_Coding_Variants_and_Reclassify(block_ctx *k) {       /* :664-752 */
  const int ec = k->entropy_coder_count;
  int pos_countdown = group_size;
  int32_t selector_idx = 1;
  int cluster = k->selector[1];
  int bit_count[max_entropy_coders + 1];
  int mtf_cluster_value[max_entropy_coders + 1];
  int mtf_cluster_index = 1;
  memset(bit_count, 0, sizeof bit_count);
#define OPTIMIZE_GROUP()                                                                              \
  do {                                                                                                \
    int min_bits = 0x7FFFFFFF, best = cluster, cost;                                                  \
    for (int cl = 1; cl <= ec; cl++) {                                                                \
      cost = bit_count[cl];                                                                           \
      for (int search = 1; search <= ec; search++) if (mtf_cluster_value[search] == cl) { mtf_cluster_index = search; break; } \
      cost += mtf_cluster_index;                                                                      \
      if (cost < min_bits) { min_bits = cost; best = cl; }                                            \
    }                                                                                                 \
    if (best != cluster) { k->selector[selector_idx] = (uint8_t)best; k->defector_groups++; }         \
    for (int search = 1; search <= ec; search++) if (mtf_cluster_value[search] == k->selector[selector_idx]) { mtf_cluster_index = search; break; } \
    for (int j = mtf_cluster_index; j >= 2; j--) mtf_cluster_value[j] = mtf_cluster_value[j - 1];     \
    mtf_cluster_value[1] = k->selector[selector_idx];                                                 \
  } while (0)
  for (int w = 1; w <= ec; w++) mtf_cluster_value[w] = w;
  k->defector_groups = 0;
  for (int32_t m = 1; m <= k->mtf_last; m++) {
    const int symbol = k->mtf_data[m];
    for (int cl = 1; cl <= ec; cl++) bit_count[cl] += k->descr[cl][symbol].bit_length;
    if (--pos_countdown == 0) {
      OPTIMIZE_GROUP();
      pos_countdown = group_size;
      if (m < k->mtf_last) {
        memset(bit_count, 0, sizeof bit_count);
        selector_idx++;
        cluster = k->selector[selector_idx];
      }
    }
  }
  if (pos_countdown < group_size) OPTIMIZE_GROUP();
#undef OPTIMIZE_GROUP
}

static void Cluster_Statistics(block_ctx *k) {                                    /* :756-779 */
  int32_t stat_cluster[max_entropy_coders + 1];
  const int32_t uniform_usage = k->selector_count / k->entropy_coder_count;
  memset(stat_cluster, 0, sizeof stat_cluster);
  k->low_cluster_usage = 0;
  for (int32_t i = 1; i <= k->selector_count; i++) stat_cluster[k->selector[i]]++;
  for (int c = 1; c <= k->entropy_coder_count; c++) if (stat_cluster[c] < uniform_usage / 2) k->low_cluster_usage = 1;
}

static void Construct(block_ctx *k, int sample_width) {                           /* :781-811 */
  Initial_Clustering_Ranking_Method(k, sample_width);
  for (int iteration = 1; iteration <= 10; iteration++) {
    Cluster_Statistics(k);
    Define_Descriptors(k);
    Simulate_Entropy_Coding_Variants_and_Reclassify(k);
    if (k->defector_groups == 0) break;
  }
  if (k->defector_groups > 0) Define_Descriptors(k);
  Cluster_Statistics(k);
}

static int32_t Compute_Total_Entropy_Cost(block_ctx *k) {                         /* :813-890 */
  const int ec = k->entropy_coder_count;
  int32_t bits = 0;
  {
    int pos_countdown = group_size;
    int32_t selector_idx = 1;
    int cluster = k->selector[1];
    for (int32_t m = 1; m <= k->mtf_last; m++) {
      bits += k->descr[cluster][k->mtf_data[m]].bit_length;
      if (--pos_countdown == 0 && m < k->mtf_last) { pos_countdown = group_size; selector_idx++; cluster = k->selector[selector_idx]; }
    }
  }
  {                                                                               /* Compute_Selectors_Cost */
    int v[max_entropy_coders + 1], idx = 1;
    for (int w = 1; w <= ec; w++) v[w] = w;
    for (int32_t i = 1; i <= k->selector_count; i++) {
      for (int s = 1; s <= ec; s++) if (v[s] == k->selector[i]) { idx = s; break; }
      for (int j = idx; j >= 2; j--) v[j] = v[j - 1];
      v[1] = k->selector[i];
      bits += idx;
    }
  }
  for (int coder = 1; coder <= ec; coder++) {                                     /* Compute_Huffman_Bit_Lengths_Cost */
    int cur = k->descr[coder][0].bit_length;
    bits += 5;
    for (int i = 0; i <= k->last_symbol_in_use; i++) {
      const int nw = k->descr[coder][i].bit_length;
      while (cur != nw) { bits += 2; cur += cur < nw ? 1 : -1; }
      bits += 1;
    }
  }
  return bits;
}

static int in_choices(const int *v, int n, int x) { for (int i = 0; i < n; i++) if (v[i] == x) return 1; return 0; }

static void Multiple_Entropy_Coders(block_ctx *k) {                               /* :541-980 */
  int32_t cost, best_cost = 0x7FFFFFFF;
  int best_ec_count = 2, best_max_code_len = 15, best_sample_width = 3;
  int mcl[2], nmcl, cc[4], ncc, sw[2], nsw;
  if (k->option == block_900k) {                                                  /* :900-925 */
    mcl[0] = 15; mcl[1] = 17; nmcl = 2;
    if (k->mtf_last <= 5000) { cc[0] = 2; cc[1] = 3; cc[2] = 6; ncc = 3; }
    else if (k->mtf_last <= 10000) { cc[0] = 3; cc[1] = 4; cc[2] = 6; ncc = 3; }
    else { cc[0] = 3; cc[1] = 4; cc[2] = 5; cc[3] = 6; ncc = 4; }
    sw[0] = 3; sw[1] = 4; nsw = 2;
  } else {
    mcl[0] = 16; nmcl = 1; cc[0] = 4; cc[1] = 6; ncc = 2; sw[0] = 4; nsw = 1;
  }
  k->low_cluster_usage = 0;
  for (int a = 0; a < nmcl; a++) {
    k->max_code_len = mcl[a];
    for (int b = 0; b < nsw; b++)
      for (int ec_test = max_entropy_coders; ec_test >= min_entropy_coders; ec_test--)
        if (k->low_cluster_usage || in_choices(cc, ncc, ec_test)) {
          k->entropy_coder_count = ec_test;
          Construct(k, sw[b]);
          cost = Compute_Total_Entropy_Cost(k);
          if (cost < best_cost) { best_cost = cost; best_ec_count = ec_test; best_max_code_len = k->max_code_len; best_sample_width = sw[b]; }
        }
  }
  k->max_code_len = best_max_code_len;
  k->entropy_coder_count = best_ec_count;
  k->best_sample_width = best_sample_width;
  Construct(k, best_sample_width);
}

static void Put_Block(block_ctx *k, Bit_Buffer_Type *o, uint32_t *combined_crc) {  /* :1014-1116 */
  int in_use_16[16];
  Put_String(o, block_header_magic, 6);
  k->block_crc = ~k->block_crc;
  Put_Bits(o, k->block_crc, 32);
  *combined_crc = ((*combined_crc << 1) | (*combined_crc >> 31)) ^ k->block_crc;
  Put_Bits(o, 0, 1);
  Put_Bits(o, (uint32_t)k->bwt_index, 24);
  /* Put_Mapping_Table */
  for (int i = 0; i < 16; i++) { in_use_16[i] = 0; for (int j = 0; j < 16; j++) if (k->in_use[i * 16 + j]) in_use_16[i] = 1; }
  for (int i = 0; i < 16; i++) Put_Bits(o, (uint32_t)in_use_16[i], 1);
  for (int i = 0; i < 16; i++) if (in_use_16[i]) for (int j = 0; j < 16; j++) Put_Bits(o, (uint32_t)k->in_use[i * 16 + j], 1);
  Put_Bits(o, (uint32_t)k->entropy_coder_count, 3);
  {                                                                               /* Put_Selectors */
    int v[max_entropy_coders + 1], idx = 1;
    Put_Bits(o, (uint32_t)k->selector_count, 15);
    for (int w = 1; w <= k->entropy_coder_count; w++) v[w] = w;
    for (int32_t i = 1; i <= k->selector_count; i++) {
      for (int s = 1; s <= k->entropy_coder_count; s++) if (v[s] == k->selector[i]) { idx = s; break; }
      for (int j = idx; j >= 2; j--) v[j] = v[j - 1];
      v[1] = k->selector[i];
      for (int bar = 1; bar <= idx - 1; bar++) Put_Bits(o, 1, 1);
      Put_Bits(o, 0, 1);
    }
  }
  for (int coder = 1; coder <= k->entropy_coder_count; coder++) {                 /* Put_Huffman_Bit_Lengths */
    int cur = k->descr[coder][0].bit_length;
    Put_Bits(o, (uint32_t)cur, 5);
    for (int i = 0; i <= k->last_symbol_in_use; i++) {
      const int nw = k->descr[coder][i].bit_length;
      while (cur != nw) {
        Put_Bits(o, 1, 1);
        if (cur < nw) { cur++; Put_Bits(o, 0, 1); } else { cur--; Put_Bits(o, 1, 1); }
      }
      Put_Bits(o, 0, 1);
    }
  }
  {                                                                               /* Entropy_Output */
    int pos_countdown = group_size;
    int32_t selector_idx = 1;
    int cluster = k->selector[1];
    for (int32_t m = 1; m <= k->mtf_last; m++) {
      const int symbol = k->mtf_data[m];
      Put_Bits(o, (uint32_t)k->descr[cluster][symbol].code, k->descr[cluster][symbol].bit_length);
      if (--pos_countdown == 0 && m < k->mtf_last) { pos_countdown = group_size; selector_idx++; cluster = k->selector[selector_idx]; }
    }
  }
}

static void block_free(block_ctx *k) {
  free(k->rle_1_data); free(k->bwt_data); free(k->mtf_data); free(k->selector); free(k->ranking);
  k->rle_1_data = NULL; k->bwt_data = NULL; k->mtf_data = NULL; k->selector = NULL; k->ranking = NULL;
}

/* Runs the data transformation and entropy calculation of Encode_Block; output is left to the caller. */
static int block_compute(block_ctx *k, const uint8_t *raw, int32_t n) {
  if (!bz_crc_ready) bz_crc_prepare();
  if (RLE_1(k, raw, n) < 0 || BWT(k) < 0 || MTF_and_RLE_2(k) < 0) return ZO_ENOMEM;
  k->selector_count = 1 + (k->mtf_last - 1) / group_size;                         /* :996 */
  k->selector = (uint8_t *)calloc((size_t)k->selector_count + 2, 1);
  k->ranking = (Pair *)calloc((size_t)k->selector_count + 2, sizeof(Pair));
  if (!k->selector || !k->ranking) return ZO_ENOMEM;
  Multiple_Entropy_Coders(k);
  return ZO_OK;
}

static int Encode_Block(const uint8_t *raw, int32_t n, int option, Bit_Buffer_Type *o, uint32_t *combined_crc) {
  block_ctx k;
  int rc;
  memset(&k, 0, sizeof k);
  k.option = option;
  rc = block_compute(&k, raw, n);
  if (rc == ZO_OK) Put_Block(&k, o, combined_crc);
  block_free(&k);
  return rc;
}

/* ---- Data_Segmentation.Segment_by_Entropy, data_segmentation.adb:39-105; seg[] receives 1-based segment ends ---- */
static int32_t Segment_by_Entropy(const uint8_t *buffer0 /* buffer (i) = buffer0[i - 1] */, int32_t len, float discrepancy_threshold,
                                  int32_t index_threshold, int32_t window_size, int32_t *seg, int32_t cap) {
  const double inv_window_size = 1.0 / (double)window_size;
  int32_t nseg = 0, seg_point, index_mark = 1;
  int32_t freq[256];
  double elem[256];
  double entropy = 0.0, entropy_mark = 0.0, p;
  memset(freq, 0, sizeof freq);
  for (int i = 0; i < 256; i++) elem[i] = 0.0;
  if (len > window_size + index_threshold) {
    for (int32_t i = 1; i <= len; i++) {
      uint8_t bt = buffer0[i - 1];
      freq[bt]++;
      if (i == window_size) {
        for (int b = 0; b < 256; b++) {
          p = (double)freq[b] * inv_window_size;
          if (p > 0.0) { elem[b] = -(p * log(p)); entropy = entropy + elem[b]; }
        }
        entropy_mark = entropy;
      } else if (i > window_size) {
        entropy = entropy - elem[bt];
        p = (double)freq[bt] * inv_window_size;
        elem[bt] = -(p * log(p));
        entropy = entropy + elem[bt];
        bt = buffer0[i - window_size - 1];
        entropy = entropy - elem[bt];
        freq[bt]--;
        p = (double)freq[bt] * inv_window_size;
        if (p > 0.0) { elem[bt] = -(p * log(p)); entropy = entropy + elem[bt]; }
        else elem[bt] = 0.0;
        if (fabs(entropy - entropy_mark) > (double)discrepancy_threshold) {
          seg_point = i - window_size;
          if (seg_point - index_mark > index_threshold) {
            if (nseg < cap) seg[nseg] = seg_point;
            nseg++;
            index_mark = seg_point;
            entropy_mark = entropy;
          }
        }
      }
    }
  }
  if (len > 0) { if (nseg < cap) seg[nseg] = len; nseg++; }
  return nseg;
}

/* ---- Encode :87-1431 ---- */
typedef struct {
  const uint8_t *in; uint64_t n, pos;
  uint8_t *out; uint64_t cap, out_len;
  int option;
  int32_t block_capacity;
  int64_t stream_rest;
  uint32_t combined_crc;
  zo_bz2_trace_fn tr; void *tr_user;
} enc_ctx;

static void Write_Byte(enc_ctx *e, uint8_t b) { if (e->out_len < e->cap) e->out[e->out_len] = b; e->out_len++; }

enum { tactic_single = 0, tactic_parts_4, tactic_segmented_1, tactic_segmented_2, n_tactics };

static int Read_and_Split_Block(enc_ctx *e, Bit_Buffer_Type *out_bit_buf, int32_t dyn_block_capacity) {   /* :1144-1378 */
  const int64_t raw_last = 10 * (int64_t)dyn_block_capacity;
  const uint8_t *raw_buf = e->in + e->pos;                                        /* raw_buf (1 ..) = raw_buf[0 ..] */
  int32_t raw_buf_index = 0;
  int64_t out_size;
  int rc = ZO_OK;
  {                                                                               /* Data_Acquisition :1161-1209 */
    int32_t rle_1_block_size = 0;
    uint8_t b, b_prev = 0;
    int run = 0, start = 1;
#define SIMULATE_STORE_RUN() do { rle_1_block_size += run < 4 ? run : 4; if (run >= 4) rle_1_block_size++; run = 1; } while (0)
    while (e->pos < e->n && rle_1_block_size + 5 < dyn_block_capacity && raw_buf_index < raw_last) {
      b = e->in[e->pos++];
      raw_buf_index++;
      if (e->stream_rest != -1) e->stream_rest--;
      if (start || b != b_prev) { SIMULATE_STORE_RUN(); start = 0; }
      else if (run == 259) { SIMULATE_STORE_RUN(); }
      else run++;
      b_prev = b;
    }
#undef SIMULATE_STORE_RUN
  }
  out_size = (int64_t)raw_buf_index * 2 + 1000000;
  if (e->option != block_900k) {                                                  /* :1357-1363 */
    out_bit_buf->destination_data = (uint8_t *)malloc((size_t)out_size);
    out_bit_buf->destination_index = 0; out_bit_buf->cap = out_size;
    if (!out_bit_buf->destination_data) return ZO_ENOMEM;
    rc = Encode_Block(raw_buf, raw_buf_index, e->option, out_bit_buf, &e->combined_crc);
    if (e->tr) e->tr(e->tr_user, (int64_t)(raw_buf - e->in), raw_buf_index, tactic_single, 1);
    for (int64_t i = 0; i < out_bit_buf->destination_index; i++) Write_Byte(e, out_bit_buf->destination_data[i]);
    free(out_bit_buf->destination_data); out_bit_buf->destination_data = NULL;
    return rc;
  }
  {                                                                               /* Block_Split_Parallel :1214-1345 */
    Bit_Buffer_Type v[n_tactics];
    uint32_t crc_v[n_tactics];
    int32_t nsub[n_tactics];
    int best = tactic_single;
    for (int t = 0; t < n_tactics; t++) {
      v[t] = *out_bit_buf; crc_v[t] = e->combined_crc; nsub[t] = 0;
      v[t].destination_data = (uint8_t *)malloc((size_t)out_size);
      v[t].destination_index = 0; v[t].cap = out_size; v[t].overflow = 0;
      if (!v[t].destination_data) rc = ZO_ENOMEM;
    }
    for (int t = tactic_single; t <= tactic_parts_4 && rc == ZO_OK; t++) {        /* Do_Simple_Cut_Type :1237-1253 */
      const int32_t slices = t == tactic_single ? 1 : 4, size = raw_buf_index / slices;
      int32_t start, stop = 0;
      for (int32_t count = 1; count <= slices && rc == ZO_OK; count++) {
        start = stop + 1;
        stop = count == slices ? raw_buf_index : count * size;
        rc = Encode_Block(raw_buf + (start - 1), stop - start + 1, e->option, &v[t], &crc_v[t]);
        nsub[t]++;
      }
    }
    for (int t = tactic_segmented_1; t <= tactic_segmented_2 && rc == ZO_OK; t++) {   /* Do_Segmented_Block_Type :1265-1297 */
      const float thr = t == tactic_segmented_1 ? 0.6f : 0.4f;
      const int32_t index_threshold = t == tactic_segmented_1 ? 4000 : 8000, window = 16000;
      const int32_t cap = raw_buf_index / (index_threshold > 0 ? index_threshold : 1) + 2;
      int32_t *seg = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
      int32_t nseg, index_start = 1;
      if (!seg) { rc = ZO_ENOMEM; break; }
      nseg = Segment_by_Entropy(raw_buf, raw_buf_index, thr, index_threshold, window, seg, cap);
      if (nseg == 0) { rc = Encode_Block(raw_buf, 0, e->option, &v[t], &crc_v[t]); nsub[t] = 1; }
      else for (int32_t s = 0; s < nseg && rc == ZO_OK; s++) {
        rc = Encode_Block(raw_buf + (index_start - 1), seg[s] - index_start + 1, e->option, &v[t], &crc_v[t]);
        index_start = seg[s] + 1;
        nsub[t]++;
      }
      free(seg);
    }
    if (rc == ZO_OK) {
      for (int t = 0; t < n_tactics; t++) if (v[t].destination_index < v[best].destination_index) best = t;   /* :1312-1318 */
      if (e->tr) e->tr(e->tr_user, (int64_t)(raw_buf - e->in), raw_buf_index, best, nsub[best]);
      for (int64_t i = 0; i < v[best].destination_index; i++) Write_Byte(e, v[best].destination_data[i]);
      out_bit_buf->bit_index = v[best].bit_index;
      out_bit_buf->buffer = v[best].buffer;
      out_bit_buf->destination_data = NULL;
      out_bit_buf->destination_index = 0;
      e->combined_crc = crc_v[best];
    }
    for (int t = 0; t < n_tactics; t++) free(v[t].destination_data);
  }
  return rc;
}

int zo_bzip2_encode(const uint8_t *in, uint64_t n, int option, int64_t size_hint, uint8_t *out, uint64_t cap, uint64_t *out_len,
                    zo_bz2_trace_fn tr, void *tr_user) {
  enc_ctx e;
  Bit_Buffer_Type main_bit_buffer;
  const int level = option == block_100k ? 1 : option == block_400k ? 4 : 9;
  int rc = ZO_OK;
  if (option < block_100k || option > block_900k) return ZO_EINVAL;
  if (!bz_crc_ready) bz_crc_prepare();
  memset(&e, 0, sizeof e);
  memset(&main_bit_buffer, 0, sizeof main_bit_buffer);
  main_bit_buffer.bit_index = 7;
  e.in = in; e.n = n; e.out = out; e.cap = cap; e.option = option;
  e.block_capacity = sub_block_size * level;
  e.stream_rest = size_hint;
  e.tr = tr; e.tr_user = tr_user;
  Write_Byte(&e, 'B'); Write_Byte(&e, 'Z'); Write_Byte(&e, 'h'); Write_Byte(&e, (uint8_t)('0' + level));   /* :1380-1387 */
  for (;;) {                                                                      /* :1411-1428 */
    const float fr = (float)e.stream_rest, fc = (float)e.block_capacity;
    if (fr >= fc * 1.05f && fr <= fc * 1.30f) rc = Read_and_Split_Block(&e, &main_bit_buffer, (int32_t)(e.stream_rest / 2));
    else rc = Read_and_Split_Block(&e, &main_bit_buffer, e.block_capacity);
    if (rc != ZO_OK) return rc;
    if (!(e.pos < e.n)) break;
  }
  {                                                                               /* Write_Stream_Footer :1391-1403 */
    uint8_t foot[11];
    main_bit_buffer.destination_data = foot; main_bit_buffer.destination_index = 0; main_bit_buffer.cap = 11;
    Put_String(&main_bit_buffer, stream_footer_magic, 6);
    Put_Bits(&main_bit_buffer, e.combined_crc, 32);
    if (main_bit_buffer.bit_index < 7) Flush_Bit_Buffer(&main_bit_buffer);
    for (int64_t i = 0; i < main_bit_buffer.destination_index; i++) Write_Byte(&e, foot[i]);
  }
  if (out_len) *out_len = e.out_len;
  return e.out_len > cap ? ZO_EINVAL : ZO_OK;
}

/* Zip.Compress.BZip2_E (zip-compress-bzip2_e.adb:44-157): size always known on this path (zip-create.adb:256-257);
 * Compression_inefficient (zip-compress.adb:479-486) is decided on the final size, as for Deflate. */
int zo_bzip2(const uint8_t *in, uint64_t n, int method, uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout) {
  uint64_t len = 0;
  int rc;
  if (method < ZO_BZIP2_1 || method > ZO_BZIP2_3) return ZO_EINVAL;
  rc = zo_bzip2_encode(in, n, method - ZO_BZIP2_1, (int64_t)n, out, cap, &len, NULL, NULL);
  if (out_len) *out_len = len;
  if (rc == ZO_EINVAL && len > cap && len >= n) rc = ZO_INEFFICIENT;   /* did not fit a buffer of the input's size: inefficient anyway */
  if (rc < 0) return rc;
  if (crc_inout) *crc_inout = zo_crc32_update(*crc_inout, in, n);
  return len >= n ? ZO_INEFFICIENT : ZO_OK;
}

/* Test hook: the stages of one Encode_Block. Any out pointer may be NULL. lens: 6 x 258 bit lengths, selectors: 1 per group. */
int zo_bz2_block(const uint8_t *raw, int32_t n, int option, uint8_t *rle_out, uint8_t *bwt_out, uint16_t *mtf_out,
                 uint8_t *selectors_out, uint8_t *lens_out, zo_bz2_block_info *info, uint8_t *bits_out, uint64_t bits_cap) {
  block_ctx k;
  int rc;
  memset(&k, 0, sizeof k);
  k.option = option;
  rc = block_compute(&k, raw, n);
  if (rc == ZO_OK) {
    if (rle_out) memcpy(rle_out, k.rle_1_data, (size_t)k.rle_1_block_size);
    if (bwt_out) memcpy(bwt_out, k.bwt_data, (size_t)k.rle_1_block_size);
    if (mtf_out) memcpy(mtf_out, k.mtf_data + 1, sizeof(uint16_t) * (size_t)k.mtf_last);
    if (selectors_out) memcpy(selectors_out, k.selector + 1, (size_t)k.selector_count);
    if (lens_out) {
      memset(lens_out, 0, 6 * 258);
      for (int c = 1; c <= k.entropy_coder_count; c++) for (int i = 0; i <= k.last_symbol_in_use; i++) lens_out[(c - 1) * 258 + i] = (uint8_t)k.descr[c][i].bit_length;
    }
    if (info) {
      info->rle_n = k.rle_1_block_size; info->bwt_index = k.bwt_index; info->mtf_n = k.mtf_last; info->selector_count = k.selector_count;
      info->coders = k.entropy_coder_count; info->max_code_len = k.max_code_len; info->sample_width = k.best_sample_width;
      info->alphabet = k.last_symbol_in_use + 1; info->block_crc = ~k.block_crc; info->bits = 0;
    }
    if (bits_out || info) {
      Bit_Buffer_Type o;
      uint32_t cc = 0;
      uint8_t *tmp = bits_out;
      uint64_t cap = bits_cap;
      memset(&o, 0, sizeof o);
      if (!tmp) { cap = (uint64_t)n * 2 + 1000000; tmp = (uint8_t *)malloc(cap); }
      if (tmp) {
        o.bit_index = 7; o.destination_data = tmp; o.cap = (int64_t)cap;
        Put_Block(&k, &o, &cc);
        if (info) info->bits = (uint64_t)o.destination_index * 8 + (uint64_t)(7 - o.bit_index);
        if (o.bit_index < 7) Flush_Bit_Buffer(&o);
        if (!bits_out) free(tmp);
      }
    }
  }
  block_free(&k);
  return rc;
}

/* Test hook: Segment_by_Entropy on buf[0 .. len) with the profile of `tactic` (2 or 3). Returns the number of segments. */
int32_t zo_bz2_segments(const uint8_t *buf, int32_t len, int tactic, int32_t *seg, int32_t cap) {
  return Segment_by_Entropy(buf, len, tactic == tactic_segmented_1 ? 0.6f : 0.4f, tactic == tactic_segmented_1 ? 4000 : 8000, 16000, seg, cap);
}
