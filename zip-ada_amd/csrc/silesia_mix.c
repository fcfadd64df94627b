/*
 * silesia_mix.c -- deterministic synthetic corpora "silesia_mix_v1" / "silesia_mix_v2" (SURVEY.md section 8d).
 *
 * v1 and v2 differ ONLY in how a segment's generator is seeded.  v1 starts segment i at the state
 * (seed + i) * gamma + c with SplitMix64's own increment gamma, so segment i+1's draw k-1 is segment
 * i's draw k: consecutive segments are shifted copies of ONE stream of draws, not independent ones
 * (round-4 review: 48-byte windows of the text class recur at distance ~65 534, xz ratio 0.0055).
 * Deflate's 32 KiB window cannot see that; BZip2's 900 k blocks and LZMA's dictionary can.  v2 seeds
 * every segment with the FINALISED value mix(seed ^ mix(index + 1)), so segments are i.i.d. as the
 * recipe says.  v1 stays because the committed golden digests were taken on v1 inputs; everything
 * that is measured on data reaching beyond 32 KiB (BZip2, LZMA legs) uses v2.
 *
 * No corpus is available offline, so the benchmark input is generated: 64 KiB segments,
 * each drawn independently (SplitMix64 seeded with seed + segment index) from five
 * classes that imitate the members of the Silesia corpus:
 *     45 %  Zipf(1.1) word text over a fixed 4096-word vocabulary, punctuation, newlines
 *     20 %  XML-like nested tagged records
 *     15 %  source-code-like lines (indentation, identifiers, operators)
 *     10 %  fixed-width decimal / CSV database rows
 *     10 %  noisy 16-bit measurement samples (Silesia's nearly incompressible members)
 * The generator is seekable per segment, so any byte range can be produced
 * independently (each rank of a multi-GPU run materialises only its own range).
 * C1 of BASELINE.md ("text class only") uses class_mask = 1.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>

#define SEG 65536u
#define VOCAB 4096

static inline uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static char vocab[VOCAB][12];
static uint8_t vocab_len[VOCAB];
static uint32_t zipf_cdf[VOCAB];   /* cumulative, scaled to 2^32 */
static uint16_t phrases[512][6];
static uint8_t phrase_len[512];
static char idents[512][14];
static uint8_t ident_len[512];
static int tables_ready = 0;

static void build_tables(void) {
  static const char letters[] = "eeeeeeeeeeeetttttttttaaaaaaaaooooooooiiiiiiinnnnnnnsssssshhhhhhrrrrrrddddlllluuucccmmmwwffggyyppbbvk";
  uint64_t s = 0x51E51Aull;
  double tot = 0, acc = 0;
  for (int w = 0; w < VOCAB; w++) {
    /* frequent words are short */
    int len = 2 + (int)(splitmix64(&s) % (uint64_t)(w < 64 ? 3 : w < 512 ? 6 : 9));
    for (int k = 0; k < len; k++) vocab[w][k] = letters[splitmix64(&s) % (sizeof letters - 1)];
    vocab_len[w] = (uint8_t)len;
  }
  for (int w = 0; w < VOCAB; w++) tot += 1.0 / pow((double)(w + 1), 1.1);
  for (int w = 0; w < VOCAB; w++) {
    acc += 1.0 / pow((double)(w + 1), 1.1) / tot;
    double v = acc * 4294967296.0;
    zipf_cdf[w] = v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
  }
  zipf_cdf[VOCAB - 1] = 0xFFFFFFFFu;
  for (int i = 0; i < 512; i++) {      /* stock phrases: natural text repeats word groups */
    int len = 2 + (int)(splitmix64(&s) % 4);
    for (int k = 0; k < len; k++) {
      uint32_t u = (uint32_t)(splitmix64(&s) >> 32);
      int lo = 0, hi = VOCAB - 1;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (zipf_cdf[mid] < u) lo = mid + 1; else hi = mid; }
      phrases[i][k] = (uint16_t)lo;
    }
    phrase_len[i] = (uint8_t)len;
  }
  for (int i = 0; i < 512; i++) {
    int len = 3 + (int)(splitmix64(&s) % 10);
    for (int k = 0; k < len; k++) idents[i][k] = (char)((k && splitmix64(&s) % 7 == 0) ? '_' : 'a' + splitmix64(&s) % 26);
    ident_len[i] = (uint8_t)len;
  }
  tables_ready = 1;
}

static inline int zipf_word(uint64_t *s) {
  uint32_t u = (uint32_t)(splitmix64(s) >> 32);
  int lo = 0, hi = VOCAB - 1;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (zipf_cdf[mid] < u) lo = mid + 1; else hi = mid; }
  return lo;
}

typedef struct { uint8_t *p; uint32_t n; } obuf;
static inline void put(obuf *o, const char *s, int len) {
  for (int i = 0; i < len && o->n < SEG; i++) o->p[o->n++] = (uint8_t)s[i];
}
static inline void putc1(obuf *o, char c) { if (o->n < SEG) o->p[o->n++] = (uint8_t)c; }
static void putnum(obuf *o, uint64_t v, int width) {
  char t[24]; int k = 0;
  do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v && k < 20);
  while (k < width) t[k++] = '0';
  while (k) putc1(o, t[--k]);
}

static void gen_text(obuf *o, uint64_t *s) {
  int sentence = 0, line = 0, ph = -1, phk = 0;
  while (o->n < SEG) {
    int w;
    if (ph < 0 && splitmix64(s) % 100 < 45) { ph = (int)(splitmix64(s) % 512); ph = ph * (int)(1 + splitmix64(s) % 8) / 8; phk = 0; }
    if (ph >= 0) { w = phrases[ph][phk++]; if (phk >= phrase_len[ph]) ph = -1; }
    else w = zipf_word(s);
    int cap = (sentence == 0);
    uint32_t start = o->n;
    put(o, vocab[w], vocab_len[w]);
    if (cap && start < SEG) o->p[start] = (uint8_t)(o->p[start] - 32);
    sentence++;
    line += vocab_len[w] + 1;
    uint64_t r = splitmix64(s) % 100;
    if (sentence > 4 && r < 12) { putc1(o, '.'); sentence = 0; }
    else if (r < 20) putc1(o, ',');
    if (line > 72) { putc1(o, '\n'); line = 0; if (splitmix64(s) % 9 == 0) putc1(o, '\n'); }
    else putc1(o, ' ');
  }
}

static void gen_xml(obuf *o, uint64_t *s) {
  static const char *tags[] = {"record", "name", "title", "author", "date", "value", "item", "description", "id", "ref"};
  uint64_t id = splitmix64(s) % 100000;
  while (o->n < SEG) {
    put(o, "  <record id=\"", 14); putnum(o, id++, 6); put(o, "\">\n", 3);
    int nf = 3 + (int)(splitmix64(s) % 5);
    for (int f = 0; f < nf; f++) {
      const char *t = tags[1 + splitmix64(s) % 9];
      int tl = (int)strlen(t);
      put(o, "    <", 5); put(o, t, tl); putc1(o, '>');
      int nw = 1 + (int)(splitmix64(s) % 4);
      for (int k = 0; k < nw; k++) { int w = zipf_word(s); if (k) putc1(o, ' '); put(o, vocab[w], vocab_len[w]); }
      put(o, "</", 2); put(o, t, tl); put(o, ">\n", 2);
    }
    put(o, "  </record>\n", 12);
  }
}

static void gen_code(obuf *o, uint64_t *s) {
  static const char *kw[] = {"if", "then", "else", "end", "loop", "for", "while", "return", "begin", "procedure", "function", "declare", "constant", "in", "out", "type", "is", "null", "case", "when"};
  static const char *ops[] = {" := ", " = ", " + ", " - ", " * ", " /= ", " <= ", " and ", " or ", " (", ")", ", ", ";"};
  int indent = 0;
  while (o->n < SEG) {
    for (int k = 0; k < indent; k++) put(o, "  ", 2);
    int nt = 2 + (int)(splitmix64(s) % 8);
    for (int k = 0; k < nt; k++) {
      uint64_t r = splitmix64(s) % 10;
      if (r < 3) { const char *t = kw[splitmix64(s) % 20]; put(o, t, (int)strlen(t)); putc1(o, ' '); }
      else if (r < 8) { int id = (int)(splitmix64(s) % 512); id = id * (int)(splitmix64(s) % 4 + 1) / 4; put(o, idents[id], ident_len[id]); }
      else putnum(o, splitmix64(s) % 1000, 1);
      const char *op = ops[splitmix64(s) % 13]; put(o, op, (int)strlen(op));
    }
    putc1(o, '\n');
    uint64_t r = splitmix64(s) % 8;
    if (r == 0 && indent < 8) indent++; else if (r == 1 && indent > 0) indent--;
    if (splitmix64(s) % 11 == 0) { put(o, "  --  ", 6); for (int k = 0; k < 5; k++) { int w = zipf_word(s); put(o, vocab[w], vocab_len[w]); putc1(o, ' '); } putc1(o, '\n'); }
  }
}

static void gen_db(obuf *o, uint64_t *s) {
  uint64_t key = splitmix64(s) % 1000000, t = 1500000000ull + splitmix64(s) % 100000000ull;
  while (o->n < SEG) {
    putnum(o, key, 8); putc1(o, ',');
    key += 1 + splitmix64(s) % 3;
    putnum(o, t, 10); putc1(o, ','); t += splitmix64(s) % 600;
    putnum(o, (splitmix64(s) % 200) * 25, 5); putc1(o, '.'); putnum(o, (splitmix64(s) % 4) * 25, 2); putc1(o, ',');
    int w = (int)(splitmix64(s) % 64); w = w * (int)(1 + splitmix64(s) % 4) / 4; put(o, vocab[w], vocab_len[w]); putc1(o, ',');
    putnum(o, splitmix64(s) % 4, 1); putc1(o, ',');
    putnum(o, splitmix64(s) % 3000, 7); putc1(o, '\n');
  }
}

/* "measurement" data, as Silesia's sao / x-ray: 16-bit little-endian samples whose low byte is
 * noise and whose high byte follows a slow random walk -- nearly incompressible for LZ77. */
static void gen_random(obuf *o, uint64_t *s) {
  unsigned hi = 0x80;
  while (o->n + 8 <= SEG) {
    uint64_t v = splitmix64(s);
    for (int k = 0; k < 4; k++) {
      unsigned step = (unsigned)(v >> (48 + 4 * k)) & 15;
      if (step < 3) hi = (hi + step - 1) & 0xFF;
      o->p[o->n++] = (uint8_t)(v >> (8 * k));
      o->p[o->n++] = (uint8_t)hi;
    }
  }
}

/* class_mask: bit0 text, bit1 xml, bit2 code, bit3 db, bit4 random; 0x1F = the full mix. */
static inline uint64_t mix64(uint64_t z) {   /* SplitMix64's finaliser alone */
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static void gen_segment(int version, uint64_t seed, uint64_t index, unsigned class_mask, uint8_t *dst) {
  static const int weight[5] = {45, 20, 15, 10, 10};
  uint64_t s = version >= 2 ? mix64(seed ^ mix64(index + 1))
                            : (seed + index) * 0x9E3779B97F4A7C15ull + 0x5A1E51Aull;
  obuf o = {dst, 0};
  int tot = 0, cls = 0;
  for (int c = 0; c < 5; c++) if (class_mask & (1u << c)) tot += weight[c];
  if (tot == 0) { class_mask = 0x1F; tot = 100; }
  int u = (int)(splitmix64(&s) % (uint64_t)tot);
  for (int c = 0; c < 5; c++) if (class_mask & (1u << c)) { if (u < weight[c]) { cls = c; break; } u -= weight[c]; }
  switch (cls) {
    case 0: gen_text(&o, &s); break;
    case 1: gen_xml(&o, &s); break;
    case 2: gen_code(&o, &s); break;
    case 3: gen_db(&o, &s); break;
    default: gen_random(&o, &s); break;
  }
}

/* Fill dst[0..len) with bytes [offset, offset+len) of the stream.  Thread-safe after the
 * first call has built the tables (call zada_silesia_mix(…, len = 0) once up front). */
static void mix_range(int version, uint64_t seed, unsigned class_mask, uint64_t offset, uint64_t len, uint8_t *dst) {
  uint8_t seg[SEG];
  if (!tables_ready) build_tables();
  uint64_t pos = offset, end = offset + len;
  while (pos < end) {
    uint64_t idx = pos / SEG, o = pos % SEG;
    uint64_t take = SEG - o; if (take > end - pos) take = end - pos;
    if (o == 0 && take == SEG) gen_segment(version, seed, idx, class_mask, dst + (pos - offset));
    else { gen_segment(version, seed, idx, class_mask, seg); memcpy(dst + (pos - offset), seg + o, take); }
    pos += take;
  }
}

void zada_silesia_mix(uint64_t seed, unsigned class_mask, uint64_t offset, uint64_t len, uint8_t *dst) {
  mix_range(1, seed, class_mask, offset, len, dst);
}

/* silesia_mix_v2: the same classes and weights, segments seeded independently (see the header). */
void zada_silesia_mix_v2(uint64_t seed, unsigned class_mask, uint64_t offset, uint64_t len, uint8_t *dst) {
  mix_range(2, seed, class_mask, offset, len, dst);
}
