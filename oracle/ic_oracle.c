/* TEST INFRASTRUCTURE ONLY -- see ic_oracle.h for what this file is and how it
 * is pinned.  Plain-C scalar restatement of the reference's block encoders,
 * written from SURVEY.md section 8 / Appendix A; every function cites the
 * reference lines it follows (paths relative to /root/reference/image_compression).
 * All arithmetic is 32-bit integer, like the reference (no floats anywhere).
 */
#include "ic_oracle.h"

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int r, g, b; } rgb_t;

/* One gathered 4x4 block: internal/pixel4x4.h:44-67 + pixel4x4.cc:23-59. */
typedef struct {
  rgb_t px[16];  /* raster order, p = 4*y + x; channels in *memory* order (no swap yet) */
  int alpha[16];
  int one_pixel; /* block entirely right-of AND below the image (pixel4x4.cc:58) */
} block4x4_t;

static const uint8_t kConstTable[256][8] = {
#include "dxtc_const_table.inc"
};

static int imax(int a, int b) { return a > b ? a : b; }
static int iabs(int a) { return a < 0 ? -a : a; }

/* pixel4x4.h:44-67 / pixel4x4.cc:23-59: clamp-to-edge gather. */
static void gather_block(const uint8_t *src, int comps, uint32_t h, uint32_t w, uint32_t stride,
                         uint32_t row, uint32_t col, block4x4_t *b) {
  for (int y = 0; y < 4; ++y) {
    uint32_t sy = row + (uint32_t)y < h ? row + (uint32_t)y : h - 1;
    const uint8_t *line = src + (size_t)sy * stride;
    for (int x = 0; x < 4; ++x) {
      uint32_t sx = col + (uint32_t)x < w ? col + (uint32_t)x : w - 1;
      const uint8_t *p = line + (size_t)sx * (size_t)comps;
      b->px[4 * y + x].r = p[0];
      b->px[4 * y + x].g = p[1];
      b->px[4 * y + x].b = p[2];
      b->alpha[4 * y + x] = comps == 4 ? p[3] : 255;
    }
  }
  b->one_pixel = (col >= w && row >= h);
}

/* ------------------------------------------------------------------ DXT --- */

static rgb_t maybe_swap(rgb_t c, int swap) { /* internal/color_util.h:118-120 */
  if (swap) { int t = c.r; c.r = c.b; c.b = t; }
  return c;
}
static int lum(rgb_t c) { return 4 * c.r + 8 * c.g + c.b; } /* color_util.h:383-395 */

static int quant8(int v, int bits) { /* color_util.h:156-164 */
  int i = v * ((1 << bits) - 1) + 128;
  return (i + (i >> 8)) >> 8;
}
typedef struct { int r, g, b; } c565_t;
static c565_t quant565(rgb_t c) { /* color_util.h:185-189 */
  c565_t q = { quant8(c.r, 5), quant8(c.g, 6), quant8(c.b, 5) };
  return q;
}
static int pack565(c565_t c) { return c.r << 11 | c.g << 5 | c.b; } /* color_util.h:91-95 */
static rgb_t expand565(c565_t c) { /* color_util.h:226-230 */
  rgb_t o = { (c.r << 3) | (c.r >> 2), (c.g << 2) | (c.g >> 4), (c.b << 3) | (c.b >> 2) };
  return o;
}
static int lum_of_absdiff_sq(rgb_t a, rgb_t b) { /* color_util.h:410-417 */
  rgb_t d = { iabs(a.r - b.r), iabs(a.g - b.g), iabs(a.b - b.b) };
  int l = lum(d);
  return l * l;
}
static rgb_t blend(rgb_t a, rgb_t b, int wa, int wb) { /* color_util.h:282-286,315-321 */
  rgb_t o = { (wa * a.r + wb * b.r) / (wa + wb), (wa * a.g + wb * b.g) / (wa + wb),
              (wa * a.b + wb * b.b) / (wa + wb) };
  return o;
}

/* internal/dxtc_const_color_table.cc:322-392.  Returns the 2-bit index every pixel gets. */
static int best_const_colors(rgb_t target, c565_t *c0, c565_t *c1, int always4) {
  c565_t single = quant565(target);
  int which = 0;
  int min_err = lum_of_absdiff_sq(target, expand565(single));
  *c0 = single;
  *c1 = single;
  if (!always4) {
    c565_t h0 = { kConstTable[target.r][2], kConstTable[target.g][6], kConstTable[target.b][2] };
    c565_t h1 = { kConstTable[target.r][3], kConstTable[target.g][7], kConstTable[target.b][3] };
    int err = lum_of_absdiff_sq(target, blend(expand565(h0), expand565(h1), 1, 1));
    if (err < min_err) {
      which = 2;
      if (pack565(h0) < pack565(h1)) { *c0 = h0; *c1 = h1; } else { *c0 = h1; *c1 = h0; }
      min_err = err;
    }
  }
  {
    c565_t t0 = { kConstTable[target.r][0], kConstTable[target.g][4], kConstTable[target.b][0] };
    c565_t t1 = { kConstTable[target.r][1], kConstTable[target.g][5], kConstTable[target.b][1] };
    int err = lum_of_absdiff_sq(target, blend(expand565(t0), expand565(t1), 2, 1));
    if (err < min_err) {
      if (pack565(t0) > pack565(t1)) { which = 2; *c0 = t0; *c1 = t1; }
      else { which = 3; *c0 = t1; *c1 = t0; }
    }
  }
  return which;
}

/* internal/dxtc_compressor.cc:482-513 (EncodeDxt1Block) with :284-311, :315-349, :353-369. */
static void encode_dxt1_block(const block4x4_t *blk, int swap, int always4, uint8_t out[8]) {
  rgb_t low = maybe_swap(blk->px[0], swap), high = low;
  if (!blk->one_pixel) {
    int low_l = INT_MAX, high_l = 0;
    for (int p = 0; p < 16; ++p) {
      rgb_t c = maybe_swap(blk->px[p], swap);
      int l = lum(c);
      if (l < low_l) { low_l = l; low = c; }
      if (l > high_l) { high_l = l; high = c; }
    }
  }
  rgb_t base0 = low, base1 = high;
  c565_t c0 = quant565(base0), c1 = quant565(base1);
  uint8_t bits[4];
  if (pack565(c0) == pack565(c1)) {
    /* dxtc.cc:353-369 -- note the colour is R/B-swapped a second time (:360). */
    int which = best_const_colors(maybe_swap(base0, swap), &c0, &c1, always4);
    uint8_t byte = (uint8_t)(which * 0x55);
    bits[0] = bits[1] = bits[2] = bits[3] = byte;
  } else {
    if (pack565(c0) < pack565(c1)) {
      rgb_t t = base0; base0 = base1; base1 = t;
      c565_t u = c0; c0 = c1; c1 = u;
    }
    /* dxtc.cc:315-349: luminance-only metric on the UNQUANTISED endpoints. */
    int tl[4];
    tl[0] = lum(base0);
    tl[1] = lum(base1);
    tl[2] = lum(blend(base0, base1, 2, 1));
    tl[3] = lum(blend(base0, base1, 1, 2));
    for (int y = 0; y < 4; ++y) {
      bits[y] = 0;
      for (int x = 0; x < 4; ++x) {
        int l = lum(maybe_swap(blk->px[4 * y + x], swap));
        int which = 0, best = (tl[0] - l) * (tl[0] - l);
        for (int k = 1; k < 4; ++k) {
          int e = (tl[k] - l) * (tl[k] - l);
          if (e < best) { best = e; which = k; }
        }
        bits[y] |= (uint8_t)(which << (2 * x));
      }
    }
  }
  int p0 = pack565(c0), p1 = pack565(c1);
  out[0] = (uint8_t)(p0 & 0xff); out[1] = (uint8_t)(p0 >> 8);
  out[2] = (uint8_t)(p1 & 0xff); out[3] = (uint8_t)(p1 >> 8);
  memcpy(out + 4, bits, 4);
}

/* dxtc.cc:374-424 (ComputeBaseAlphas) + :427-479 (ComputeAlphaBits) + :103-158 packing. */
static void encode_dxt5_alpha(const block4x4_t *blk, uint8_t out[8]) {
  if (blk->one_pixel) {
    out[0] = out[1] = (uint8_t)blk->alpha[0];
    memset(out + 2, 0, 6);
    return;
  }
  int n0 = 0, n255 = 0, lo = 255, hi = 0;
  for (int p = 0; p < 16; ++p) {
    int a = blk->alpha[p];
    if (a == 0) ++n0;
    else if (a == 255) ++n255;
    else { if (a < lo) lo = a; if (a > hi) hi = a; }
  }
  if (lo > hi) { lo = 0; hi = 255; }
  int a0, a1;
  if (n0 > 1 || n255 > 1) { a0 = lo; a1 = hi; }
  else {
    if (n0 > 0) lo = 0;
    if (n255 > 0) hi = 255;
    a0 = hi; a1 = lo;
  }
  int t[8];
  t[0] = a0; t[1] = a1;
  if (a0 <= a1) {
    t[2] = (4 * a0 + a1) / 5; t[3] = (3 * a0 + 2 * a1) / 5;
    t[4] = (2 * a0 + 3 * a1) / 5; t[5] = (a0 + 4 * a1) / 5;
    t[6] = 0; t[7] = 255;
  } else {
    t[2] = (6 * a0 + a1) / 7; t[3] = (5 * a0 + 2 * a1) / 7; t[4] = (4 * a0 + 3 * a1) / 7;
    t[5] = (3 * a0 + 4 * a1) / 7; t[6] = (2 * a0 + 5 * a1) / 7; t[7] = (a0 + 6 * a1) / 7;
  }
  uint64_t codes = 0;
  for (int p = 0; p < 16; ++p) {
    int a = blk->alpha[p];
    int which = 0, best = (t[0] - a) * (t[0] - a);
    for (int k = 1; k < 8; ++k) {
      int e = (t[k] - a) * (t[k] - a);
      if (e < best) { best = e; which = k; }
    }
    codes |= (uint64_t)which << (3 * p);
  }
  out[0] = (uint8_t)a0; out[1] = (uint8_t)a1;
  for (int i = 0; i < 6; ++i) out[2 + i] = (uint8_t)(codes >> (8 * i));
}

/* ------------------------------------------------------------------ ETC1 -- */

static const int kEtcModifiers[8][4] = { /* OES_compressed_ETC1_RGB8_texture table; etc.cc:101-110 */
  { 2, 8, -2, -8 },     { 5, 17, -5, -17 },   { 9, 29, -9, -29 },    { 13, 42, -13, -42 },
  { 18, 60, -18, -60 }, { 24, 80, -24, -80 }, { 33, 106, -33, -106 }, { 47, 183, -47, -183 },
};
static int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); } /* color_util.h:248-265 */

/* Sub-block pixel lists (etc.cc:464-467): flip=0 -> columns {0,1} | {2,3}; flip=1 -> rows {0,1} | {2,3}. */
static int in_subblock(int flip, int second, int y, int x) {
  int v = flip ? y : x;
  return second ? v >= 2 : v < 2;
}

/* etc.cc:350-385: per-pixel argmin over the 4 modifiers, ties to the lowest index. */
static uint32_t codeword_error(const block4x4_t *b, int flip, int second, int cw, rgb_t base,
                               uint32_t *indices) {
  uint32_t total = 0, idx = 0;
  rgb_t cand[4];
  for (int k = 0; k < 4; ++k) {
    int m = kEtcModifiers[cw][k];
    cand[k].r = clamp255(base.r + m); cand[k].g = clamp255(base.g + m); cand[k].b = clamp255(base.b + m);
  }
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) {
      if (!in_subblock(flip, second, y, x)) continue;
      rgb_t t = b->px[4 * y + x];
      int best_k = 0;
      uint32_t best = 0;
      for (int k = 0; k < 4; ++k) {
        int dr = cand[k].r - t.r, dg = cand[k].g - t.g, db = cand[k].b - t.b;
        uint32_t e = (uint32_t)(dr * dr + dg * dg + db * db);
        if (k == 0 || e < best) { best = e; best_k = k; }
      }
      int p = 4 * x + y; /* etc.cc:131-137 column-major bit position */
      idx |= (uint32_t)(best_k & 1) << p;
      idx |= (uint32_t)(best_k >> 1) << (p + 16);
      total += best;
    }
  *indices = idx;
  return total;
}

/* etc.cc:391-409 (exhaustive) and :415-455 (heuristic). */
static int pick_codeword(const block4x4_t *b, int flip, int second, rgb_t base, int heuristic,
                         uint32_t *indices, uint32_t *err) {
  if (heuristic) {
    int sr = 0, sg = 0, sb = 0;
    for (int y = 0; y < 4; ++y)
      for (int x = 0; x < 4; ++x) {
        if (!in_subblock(flip, second, y, x)) continue;
        rgb_t t = b->px[4 * y + x];
        sr += iabs(base.r - t.r); sg += iabs(base.g - t.g); sb += iabs(base.b - t.b);
      }
    int dev = imax(sr / 8, imax(sg / 8, sb / 8));
    int cw = dev > 144 ? 7 : dev > 93 ? 6 : dev > 70 ? 5 : dev > 51 ? 4 : dev > 35 ? 3
           : dev > 23 ? 2 : dev > 12 ? 1 : 0;
    *err = codeword_error(b, flip, second, cw, base, indices);
    return cw;
  }
  int best_cw = -1;
  *err = 0xffffffffu;
  for (int cw = 0; cw < 8; ++cw) {
    uint32_t idx, e = codeword_error(b, flip, second, cw, base, &idx);
    if (e < *err) { best_cw = cw; *indices = idx; *err = e; }
  }
  return best_cw;
}

static int ext5(int v) { return (v << 3) | ((v >> 2) & 7); } /* color_util.h:200-202 */
static int ext4(int v) { return (v << 4) | v; }              /* color_util.h:193-195 */

/* etc.cc:460-542.  Returns hi/lo words (before the big-endian byte order of etc.cc:172-180). */
static uint32_t encode_etc1_flip(const block4x4_t *b, int flip, int heuristic, uint32_t *hi_out,
                                 uint32_t *lo_out) {
  int sum[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) {
      int s = in_subblock(flip, 1, y, x);
      sum[s][0] += b->px[4 * y + x].r; sum[s][1] += b->px[4 * y + x].g; sum[s][2] += b->px[4 * y + x].b;
    }
  int avg[2][3], q5[2][3], d[3];
  int diff_mode = 1;
  for (int c = 0; c < 3; ++c) {
    avg[0][c] = sum[0][c] / 8; avg[1][c] = sum[1][c] / 8; /* etc.cc:299-312 */
    q5[0][c] = avg[0][c] >> 3; q5[1][c] = avg[1][c] >> 3; /* QuantizeRgbFast<5> */
    d[c] = q5[1][c] - q5[0][c];
    if (d[c] < -4 || d[c] > 3) diff_mode = 0;
  }
  uint32_t hi = (uint32_t)flip;
  rgb_t base[2];
  if (diff_mode) {
    hi |= 2u;
    hi |= (uint32_t)q5[0][0] << 27 | (uint32_t)q5[0][1] << 19 | (uint32_t)q5[0][2] << 11;
    hi |= ((uint32_t)d[0] & 7u) << 24 | ((uint32_t)d[1] & 7u) << 16 | ((uint32_t)d[2] & 7u) << 8;
    for (int s = 0; s < 2; ++s) { base[s].r = ext5(q5[s][0]); base[s].g = ext5(q5[s][1]); base[s].b = ext5(q5[s][2]); }
  } else {
    int q4[2][3];
    for (int s = 0; s < 2; ++s) for (int c = 0; c < 3; ++c) q4[s][c] = avg[s][c] >> 4;
    hi |= (uint32_t)q4[0][0] << 28 | (uint32_t)q4[0][1] << 20 | (uint32_t)q4[0][2] << 12;
    hi |= (uint32_t)q4[1][0] << 24 | (uint32_t)q4[1][1] << 16 | (uint32_t)q4[1][2] << 8;
    for (int s = 0; s < 2; ++s) { base[s].r = ext4(q4[s][0]); base[s].g = ext4(q4[s][1]); base[s].b = ext4(q4[s][2]); }
  }
  uint32_t idx1, idx2, e1, e2;
  int cw1 = pick_codeword(b, flip, 0, base[0], heuristic, &idx1, &e1);
  int cw2 = pick_codeword(b, flip, 1, base[1], heuristic, &idx2, &e2);
  hi |= (uint32_t)cw1 << 5 | (uint32_t)cw2 << 2;
  *hi_out = hi;
  *lo_out = idx1 | idx2;
  return e1 + e2;
}

/* etc.cc:545-586 (EncodeEtc1Block). */
static void encode_etc1_block(const block4x4_t *b, int strategy, uint8_t out[8]) {
  uint32_t hi, lo;
  if (strategy == ICO_ETC_SPLIT_HORIZONTALLY) {
    encode_etc1_flip(b, 1, 0, &hi, &lo);
  } else if (strategy == ICO_ETC_SPLIT_VERTICALLY) {
    encode_etc1_flip(b, 0, 0, &hi, &lo);
  } else if (strategy == ICO_ETC_HEURISTIC) {
    /* etc.cc:553-574, including the (2,2)-twice / (3,3)-never quirk at :563-564. */
    static const int quad[4][4] = { { 0, 1, 4, 5 }, { 8, 9, 12, 13 }, { 2, 3, 6, 7 }, { 10, 11, 14, 10 } };
    int s[4][3];
    for (int q = 0; q < 4; ++q) {
      s[q][0] = s[q][1] = s[q][2] = 0;
      for (int i = 0; i < 4; ++i) {
        s[q][0] += b->px[quad[q][i]].r; s[q][1] += b->px[quad[q][i]].g; s[q][2] += b->px[quad[q][i]].b;
      }
    }
    uint32_t e_lr = 0, e_tb = 0;
    for (int c = 0; c < 3; ++c) {
      int left = (s[0][c] + s[1][c]) / 8, right = (s[2][c] + s[3][c]) / 8;
      int top = (s[0][c] + s[2][c]) / 8, bottom = (s[1][c] + s[3][c]) / 8;
      e_lr += (uint32_t)((right - left) * (right - left));
      e_tb += (uint32_t)((bottom - top) * (bottom - top));
    }
    encode_etc1_flip(b, e_lr > e_tb ? 0 : 1, 1, &hi, &lo);
  } else { /* kSmallerError and any other value (default: label, etc.cc:575) */
    uint32_t hi_tb, lo_tb;
    uint32_t e_lr = encode_etc1_flip(b, 0, 0, &hi, &lo);
    uint32_t e_tb = encode_etc1_flip(b, 1, 0, &hi_tb, &lo_tb);
    if (!(e_lr <= e_tb)) { hi = hi_tb; lo = lo_tb; }
  }
  /* etc.cc:172-180: hi word then lo word, each big-endian. */
  out[0] = (uint8_t)(hi >> 24); out[1] = (uint8_t)(hi >> 16); out[2] = (uint8_t)(hi >> 8); out[3] = (uint8_t)hi;
  out[4] = (uint8_t)(lo >> 24); out[5] = (uint8_t)(lo >> 16); out[6] = (uint8_t)(lo >> 8); out[7] = (uint8_t)lo;
}

/* --------------------------------------------------------------- 4x4 grid -- */

static size_t block_bytes(int codec) { return codec == ICO_DXT5 ? 16 : 8; }

typedef struct {
  int codec, etc_strategy, comps, swap;
  uint32_t h, w, grid_cols, stride;
  uint32_t row_begin, row_end; /* block rows */
  const uint8_t *src;
  uint8_t *out;
} slab_t;

/* helper.h:202-214 / :504-518: raster loop over the block grid. */
static void *encode_slab(void *arg) {
  const slab_t *s = (const slab_t *)arg;
  size_t bb = block_bytes(s->codec);
  block4x4_t blk;
  for (uint32_t br = s->row_begin; br < s->row_end; ++br)
    for (uint32_t bc = 0; bc < s->grid_cols; ++bc) {
      uint8_t *o = s->out + ((size_t)br * s->grid_cols + bc) * bb;
      gather_block(s->src, s->comps, s->h, s->w, s->stride, br * 4, bc * 4, &blk);
      if (s->codec == ICO_DXT1) {
        encode_dxt1_block(&blk, s->swap, 0, o);
      } else if (s->codec == ICO_DXT5) { /* dxtc.cc:516-528: alpha block then colour block */
        encode_dxt5_alpha(&blk, o);
        encode_dxt1_block(&blk, s->swap, 1, o + 8);
      } else {
        encode_etc1_block(&blk, s->etc_strategy, o);
      }
    }
  return NULL;
}

/* ---------------------------------------------------------------- PVRTC --- */
/* internal/pvrtc_compressor.cc; 2bpp, 8x4 blocks, three whole-image passes. */

typedef struct { uint8_t r, g, b, a; } rgba_t;

static uint32_t color_diff(rgba_t a, rgba_t b) { /* pvrtc.cc:74-77 */
  return (uint32_t)(iabs(a.r - b.r) + iabs(a.g - b.g) + iabs(a.b - b.b) + iabs(a.a - b.a));
}
static uint8_t bit_depth_reduce(uint8_t v, unsigned depth) { /* pvrtc.cc:93-106 */
  uint8_t mask = (uint8_t)(((1u << depth) - 1u) << (8 - depth));
  uint8_t e = v & mask;
  uint8_t r = (uint8_t)(e | (e >> depth));
  if (depth <= 3) r |= (uint8_t)(e >> (depth * 2));
  return r;
}
static rgba_t channel_reduce(rgba_t c, int is_b) { /* pvrtc.cc:337-349 */
  if (c.a == 255) {
    c.r = bit_depth_reduce(c.r, 5); c.g = bit_depth_reduce(c.g, 5);
    c.b = bit_depth_reduce(c.b, is_b ? 5 : 4);
  } else {
    c.r = bit_depth_reduce(c.r, 4); c.g = bit_depth_reduce(c.g, 4);
    c.b = bit_depth_reduce(c.b, is_b ? 4 : 3);
    c.a = bit_depth_reduce(c.a, 3);
  }
  return c;
}
static rgba_t load_rgba(const uint8_t *img, uint32_t index) {
  rgba_t c = { img[4 * (size_t)index], img[4 * (size_t)index + 1], img[4 * (size_t)index + 2],
               img[4 * (size_t)index + 3] };
  return c;
}

/* pvrtc.cc:255-329.  NOTE the reference initialises every "max" candidate to
 * image index 0 (pvrtc.cc:268-269) and only replaces it when a fitness value is
 * > 0, so an all-zero axis compares against -- and may select -- the image's
 * first pixel rather than a pixel of this block. */
static void pvrtc_extremes(const uint8_t *img, uint32_t w, uint32_t x0, uint32_t y0,
                           uint32_t *ia, uint32_t *ib) {
  uint32_t fit[5][2], idx[5][2];
  for (int i = 0; i < 5; ++i) { fit[i][0] = 0xffffffffu; fit[i][1] = 0; idx[i][0] = idx[i][1] = 0; }
  for (uint32_t y = y0; y < y0 + 4; ++y)
    for (uint32_t x = x0; x < x0 + 8; ++x) {
      uint32_t index = y * w + x; /* callers never pass out-of-range blocks, so no wrap needed */
      rgba_t c = load_rgba(img, index);
      uint32_t v[5] = { (77u * c.r + 150u * c.g + 28u * c.b) / 256u, c.r, c.g, c.b, c.a };
      for (int i = 0; i < 5; ++i) {
        if (v[i] < fit[i][0]) { fit[i][0] = v[i]; idx[i][0] = index; }
        if (v[i] > fit[i][1]) { fit[i][1] = v[i]; idx[i][1] = index; }
      }
    }
  uint32_t best_diff = 0, best = 0;
  for (uint32_t i = 0; i < 5; ++i) {
    uint32_t d = color_diff(load_rgba(img, idx[i][0]), load_rgba(img, idx[i][1]));
    if (d > best_diff) { best = i; best_diff = d; }
  }
  uint32_t a = idx[best][0], b = idx[best][1];
  rgba_t ca = load_rgba(img, a), cb = load_rgba(img, b);
  if ((uint32_t)cb.r + cb.g + cb.b + cb.a < (uint32_t)ca.r + ca.g + ca.b + ca.a) { uint32_t t = a; a = b; b = t; }
  *ia = a; *ib = b;
}

/* pvrtc.cc:208-237 + :173-192: bilinear up-sampling of the low-res A or B image with wrap. */
static rgba_t pvrtc_interp(const rgba_t *low, uint32_t w, uint32_t h, uint32_t x, uint32_t y) {
  uint32_t left = ((x - 4) & (w - 1)) >> 3, top = ((y - 2) & (h - 1)) >> 2;
  uint32_t right = (left + 1) & ((w >> 3) - 1), bottom = (top + 1) & ((h >> 2) - 1);
  uint32_t px = (x + 4) & 7, py = (y + 2) & 3, lw = w / 8;
  rgba_t c00 = low[top * lw + left], c01 = low[top * lw + right];
  rgba_t c10 = low[bottom * lw + left], c11 = low[bottom * lw + right];
  uint32_t a = (4 - py) * (8 - px), b = (4 - py) * px, c = py * (8 - px), d = py * px;
  rgba_t o;
  o.r = (uint8_t)((a * c00.r + b * c01.r + c * c10.r + d * c11.r) / 32);
  o.g = (uint8_t)((a * c00.g + b * c01.g + c * c10.g + d * c11.g) / 32);
  o.b = (uint8_t)((a * c00.b + b * c01.b + c * c10.b + d * c11.b) / 32);
  o.a = (uint8_t)((a * c00.a + b * c01.a + c * c10.a + d * c11.a) / 32);
  return o;
}
static rgba_t pvrtc_apply_mod(rgba_t a, rgba_t b, int mod) { /* pvrtc.cc:120-144 */
  rgba_t o = a;
  int wa = mod == 1 ? 5 : 3, wb = 8 - wa;
  if (mod == 3) return b;
  if (mod == 0) return a;
  o.r = (uint8_t)((wa * a.r + wb * b.r) / 8); o.g = (uint8_t)((wa * a.g + wb * b.g) / 8);
  o.b = (uint8_t)((wa * a.b + wb * b.b) / 8); o.a = (uint8_t)((wa * a.a + wb * b.a) / 8);
  return o;
}
static uint8_t pvrtc_best_mod(rgba_t c, rgba_t a, rgba_t b) { /* pvrtc.cc:148-166: early exit */
  uint32_t best = color_diff(c, a);
  uint8_t best_mod = 0;
  for (int m = 1; m < 4; ++m) {
    uint32_t d = color_diff(c, pvrtc_apply_mod(a, b, m));
    if (d < best) { best = d; best_mod = (uint8_t)m; } else return best_mod;
  }
  return best_mod;
}
static uint32_t pvrtc_pack_colors(rgba_t a, rgba_t b, int mode_is_1bpp) { /* pvrtc.cc:356-388 */
  uint32_t v = 0;
  if (a.a == 255) v |= 1u << 15 | (uint32_t)(a.b >> 4) << 1 | (uint32_t)(a.g >> 3) << 5 | (uint32_t)(a.r >> 3) << 10;
  else v |= (uint32_t)(a.b >> 5) << 1 | (uint32_t)(a.g >> 4) << 4 | (uint32_t)(a.r >> 4) << 8 | (uint32_t)(a.a >> 5) << 12;
  if (b.a == 255) v |= 1u << 31 | (uint32_t)(b.b >> 3) << 16 | (uint32_t)(b.g >> 3) << 21 | (uint32_t)(b.r >> 3) << 26;
  else v |= (uint32_t)(b.b >> 4) << 16 | (uint32_t)(b.g >> 4) << 20 | (uint32_t)(b.r >> 4) << 24 | (uint32_t)(b.a >> 5) << 28;
  if (!mode_is_1bpp) v |= 1u;
  return v;
}

enum { PV_1BPP = 0, PV_AVG4 = 1, PV_VERT = 2, PV_HORZ = 3 };

static int pvrtc_encode_image(const uint8_t *img, uint32_t w, uint32_t h, uint8_t *out) {
  uint32_t nblocks = w * h / 32, lw = w / 8;
  rgba_t *la = (rgba_t *)malloc(sizeof(rgba_t) * nblocks);
  rgba_t *lb = (rgba_t *)malloc(sizeof(rgba_t) * nblocks);
  uint8_t *mod = (uint8_t *)malloc((size_t)w * h);
  if (!la || !lb || !mod) { free(la); free(lb); free(mod); return 0; }
  /* Morph, pvrtc.cc:506-521 */
  for (uint32_t y = 0; y < h; y += 4)
    for (uint32_t x = 0; x < w; x += 8) {
      uint32_t ia, ib;
      pvrtc_extremes(img, w, x, y, &ia, &ib);
      la[(y / 4) * lw + x / 8] = channel_reduce(load_rgba(img, ia), 0);
      lb[(y / 4) * lw + x / 8] = channel_reduce(load_rgba(img, ib), 1);
    }
  /* Modulate, pvrtc.cc:527-540 */
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x)
      mod[(size_t)y * w + x] = pvrtc_best_mod(load_rgba(img, y * w + x), pvrtc_interp(la, w, h, x, y),
                                              pvrtc_interp(lb, w, h, x, y));
  /* Encode, pvrtc.cc:551-580 */
  for (uint32_t i = 0; i < nblocks; ++i) {
    uint32_t bx = 0, by = 0;
    for (int j = 0; j < 16; ++j) { /* pvrtc.cc:80-86 */
      bx |= ((i >> (2 * j + 1)) & 1u) << j;
      by |= ((i >> (2 * j)) & 1u) << j;
    }
    /* mode, pvrtc.cc:395-447 (counter names are swapped in the source; kept as-is) */
    uint32_t inter = 0, hc = 0, vc = 0;
    for (uint32_t y = 0; y < 4; ++y)
      for (uint32_t x = 0; x < 8; ++x) {
        uint32_t yy = by * 4 + y, xx = bx * 8 + x;
        int m = mod[(size_t)yy * w + xx];
        if (m == 1 || m == 2) ++inter;
        int m_right = mod[(size_t)yy * w + ((xx + 1) & (w - 1))];
        int m_down = mod[(size_t)((yy + 1) & (h - 1)) * w + xx];
        hc += (uint32_t)iabs(m - m_down);
        vc += (uint32_t)iabs(m - m_right);
      }
    int mode = PV_AVG4;
    if (inter <= 4) mode = PV_1BPP;
    else if (vc > 10 && vc > hc * 2) mode = PV_VERT;
    else if (hc > 10 && hc > vc * 2) mode = PV_HORZ;
    /* data, pvrtc.cc:456-496 */
    uint32_t data = 0, bitpos = 0;
    for (uint32_t y = 0; y < 4; ++y)
      for (uint32_t x = 0; x < 8; ++x) {
        uint32_t m = mod[(size_t)(by * 4 + y) * w + (bx * 8 + x)];
        if (mode == PV_1BPP) { data |= (m / 2) << bitpos; ++bitpos; continue; }
        if ((x ^ y) & 1) continue;
        if (bitpos == 0) { if (mode == PV_AVG4) m &= 2; else m |= 1; }
        else if (bitpos == 20) { if (mode == PV_VERT) m |= 1; else m &= 2; }
        data |= (m & 3u) << bitpos;
        bitpos += 2;
      }
    uint32_t colors = pvrtc_pack_colors(la[by * lw + bx], lb[by * lw + bx], mode == PV_1BPP);
    uint8_t *o = out + (size_t)i * 8;
    o[0] = (uint8_t)data; o[1] = (uint8_t)(data >> 8); o[2] = (uint8_t)(data >> 16); o[3] = (uint8_t)(data >> 24);
    o[4] = (uint8_t)colors; o[5] = (uint8_t)(colors >> 8); o[6] = (uint8_t)(colors >> 16); o[7] = (uint8_t)(colors >> 24);
  }
  free(la); free(lb); free(mod);
  return 1;
}

/* ---- PVRTC1 4 bpp encoder: EXTENSION, PARITY UNPINNED.  BASELINE.json's config 5 names "PVRTC 4bpp"; the reference only
 * implements 2 bpp (public/pvrtc_compressor.h:15-18), so there is nothing to pin this against.  It is the
 * reference's 2 bpp encoder with the block shape changed, rule by rule:
 *   - blocks are 4 x 4 pixels; Morph (pvrtc.cc:506-521): GetExtremesFast (:255-329, incl. the image-pixel-0 initialisation of
 *     the maxima, :268-269) over the block's 16 pixels in raster order, ApplyColorChannelReduction (:337-349) unchanged;
 *   - Modulate (pvrtc.cc:527-540): BestModulation (:148-166, early exit) against the A / B images up-sampled like
 *     GetInterpolatedColor2BPP (:208-237) with the 4 bpp geometry -- block centres at (2, 2), weights (x + 2) & 3 and
 *     (y + 2) & 3 out of 4 in BOTH directions, toroidal wrap, sum / 16;
 *   - Encode: every pixel keeps its 2-bit value, pixel (x, y) at bits 2 (4 y + x) of the modulation word (PVRTC1 4 bpp block
 *     layout); colour word = EncodeColors (pvrtc.cc:356-388) with bit 0 (the modulation-mode flag of a 4 bpp block) clear:
 *     standard weights 0, 3/8, 5/8, 1 -- the only ones ApplyModulation (:120-144) has; blocks in the same Z order
 *     (pvrtc.cc:80-86: x in the odd bits). */
static void pvrtc4_extremes(const uint8_t *img, uint32_t w, uint32_t x0, uint32_t y0, uint32_t *ia, uint32_t *ib) {
  uint32_t fit[5][2], idx[5][2];
  for (int i = 0; i < 5; ++i) { fit[i][0] = 0xffffffffu; fit[i][1] = 0; idx[i][0] = idx[i][1] = 0; }
  for (uint32_t y = y0; y < y0 + 4; ++y)
    for (uint32_t x = x0; x < x0 + 4; ++x) {
      uint32_t index = y * w + x;
      rgba_t c = load_rgba(img, index);
      uint32_t v[5] = { (77u * c.r + 150u * c.g + 28u * c.b) / 256u, c.r, c.g, c.b, c.a };
      for (int i = 0; i < 5; ++i) {
        if (v[i] < fit[i][0]) { fit[i][0] = v[i]; idx[i][0] = index; }
        if (v[i] > fit[i][1]) { fit[i][1] = v[i]; idx[i][1] = index; }
      }
    }
  uint32_t best_diff = 0, best = 0;
  for (uint32_t i = 0; i < 5; ++i) {
    uint32_t d = color_diff(load_rgba(img, idx[i][0]), load_rgba(img, idx[i][1]));
    if (d > best_diff) { best = i; best_diff = d; }
  }
  uint32_t a = idx[best][0], b = idx[best][1];
  rgba_t ca = load_rgba(img, a), cb = load_rgba(img, b);
  if ((uint32_t)cb.r + cb.g + cb.b + cb.a < (uint32_t)ca.r + ca.g + ca.b + ca.a) { uint32_t t = a; a = b; b = t; }
  *ia = a; *ib = b;
}
static rgba_t pvrtc4_interp(const rgba_t *low, uint32_t n, uint32_t x, uint32_t y) {
  uint32_t lw = n / 4, left = ((x - 2) & (n - 1)) >> 2, top = ((y - 2) & (n - 1)) >> 2;
  uint32_t right = (left + 1) & (lw - 1), bottom = (top + 1) & (lw - 1);
  uint32_t px = (x + 2) & 3, py = (y + 2) & 3;
  rgba_t c00 = low[top * lw + left], c01 = low[top * lw + right], c10 = low[bottom * lw + left], c11 = low[bottom * lw + right];
  uint32_t a = (4 - py) * (4 - px), b = (4 - py) * px, c = py * (4 - px), d = py * px;
  rgba_t o;
  o.r = (uint8_t)((a * c00.r + b * c01.r + c * c10.r + d * c11.r) / 16);
  o.g = (uint8_t)((a * c00.g + b * c01.g + c * c10.g + d * c11.g) / 16);
  o.b = (uint8_t)((a * c00.b + b * c01.b + c * c10.b + d * c11.b) / 16);
  o.a = (uint8_t)((a * c00.a + b * c01.a + c * c10.a + d * c11.a) / 16);
  return o;
}
static int pvrtc4_encode_image(const uint8_t *img, uint32_t n, uint8_t *out) {
  uint32_t lw = n / 4, nblocks = lw * lw;
  rgba_t *la = (rgba_t *)malloc(sizeof(rgba_t) * nblocks), *lb = (rgba_t *)malloc(sizeof(rgba_t) * nblocks);
  if (!la || !lb) { free(la); free(lb); return 0; }
  for (uint32_t by = 0; by < lw; ++by)
    for (uint32_t bx = 0; bx < lw; ++bx) {
      uint32_t ia, ib;
      pvrtc4_extremes(img, n, bx * 4, by * 4, &ia, &ib);
      la[by * lw + bx] = channel_reduce(load_rgba(img, ia), 0);
      lb[by * lw + bx] = channel_reduce(load_rgba(img, ib), 1);
    }
  for (uint32_t i = 0; i < nblocks; ++i) {
    uint32_t bx = 0, by = 0;
    for (int j = 0; j < 16; ++j) { /* pvrtc.cc:80-86 */
      bx |= ((i >> (2 * j + 1)) & 1u) << j;
      by |= ((i >> (2 * j)) & 1u) << j;
    }
    uint32_t data = 0;
    for (uint32_t y = 0; y < 4; ++y)
      for (uint32_t x = 0; x < 4; ++x) {
        uint32_t xx = bx * 4 + x, yy = by * 4 + y;
        uint32_t m = pvrtc_best_mod(load_rgba(img, yy * n + xx), pvrtc4_interp(la, n, xx, yy), pvrtc4_interp(lb, n, xx, yy));
        data |= m << (2 * (4 * y + x));
      }
    uint32_t colors = pvrtc_pack_colors(la[by * lw + bx], lb[by * lw + bx], 1 /* bit 0 clear: standard modulation */);
    uint8_t *o = out + (size_t)i * 8;
    o[0] = (uint8_t)data; o[1] = (uint8_t)(data >> 8); o[2] = (uint8_t)(data >> 16); o[3] = (uint8_t)(data >> 24);
    o[4] = (uint8_t)colors; o[5] = (uint8_t)(colors >> 8); o[6] = (uint8_t)(colors >> 16); o[7] = (uint8_t)(colors >> 24);
  }
  free(la); free(lb);
  return 1;
}

/* ------------------------------------------------------------ public API -- */

static uint32_t nblk(uint32_t n) { return (n + 3) / 4; } /* helper.h:86-88 */
static int is_pow2(uint32_t x) { return x != 0 && !(x & (x - 1)); }

size_t ico_encoded_size(int codec, uint32_t gh, uint32_t gw) {
  if (codec == ICO_PVRTC2) return (size_t)gw * gh / 4;
  if (codec == ICO_PVRTC4) return (size_t)gw * gh / 2;
  return (size_t)nblk(gh) * nblk(gw) * block_bytes(codec);
}

int ico_encode(int codec, int etc_strategy, int comps, int swap, uint32_t h, uint32_t w, uint32_t gh,
               uint32_t gw, uint32_t stride, const uint8_t *src, uint8_t *out, int threads) {
  if (!src || !out || h == 0 || w == 0 || (comps != 3 && comps != 4)) return 0;
  if (codec == ICO_PVRTC2) {
    /* pvrtc.cc:636-667 preconditions; source is always read as RGBA8888 */
    if (!is_pow2(w) || !is_pow2(h) || w != h || w % 8 || h % 4 || comps != 4 || stride != w * 4) return 0;
    return pvrtc_encode_image(src, w, h, out);
  }
  if (codec == ICO_PVRTC4) { /* extension: the same preconditions (square power of two, at least 8, RGBA8, no row padding) */
    if (!is_pow2(w) || w != h || w < 8 || comps != 4 || stride != w * 4) return 0;
    return pvrtc4_encode_image(src, w, out);
  }
  if (codec != ICO_DXT1 && codec != ICO_DXT5 && codec != ICO_ETC1) return 0;
  if (codec == ICO_DXT5 && comps != 4) return 0;
  if (gh < h) gh = h;
  if (gw < w) gw = w;
  uint32_t rows = nblk(gh), cols = nblk(gw);
  if (threads < 1) threads = 1;
  if ((uint32_t)threads > rows) threads = (int)rows;
  slab_t *slabs = (slab_t *)malloc(sizeof(slab_t) * (size_t)threads);
  pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  if (!slabs || !tids) { free(slabs); free(tids); return 0; }
  for (int t = 0; t < threads; ++t) {
    slab_t s = { codec, etc_strategy, comps, swap, h, w, cols, stride,
                 (uint32_t)((uint64_t)rows * (uint32_t)t / (uint32_t)threads),
                 (uint32_t)((uint64_t)rows * (uint32_t)(t + 1) / (uint32_t)threads), src, out };
    slabs[t] = s;
  }
  if (threads == 1) encode_slab(&slabs[0]);
  else {
    for (int t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, encode_slab, &slabs[t]);
    for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
  }
  free(slabs); free(tids);
  return 1;
}

size_t ico_compute_compressed_data_size(int compressor, int format, uint32_t h, uint32_t w) {
  int comps = (format == ICO_RGB || format == ICO_BGR) ? 3 : (format == ICO_RGBA || format == ICO_BGRA) ? 4 : 0;
  if (compressor == ICO_COMPRESSOR_PVRTC) return (size_t)(w * h / 4); /* pvrtc.cc:631-634, 32-bit product */
  if (h == 0 || w == 0) return 0;
  size_t blocks = (size_t)imax(1, (int)nblk(h)) * (size_t)imax(1, (int)nblk(w));
  if (compressor == ICO_COMPRESSOR_DXTC) return blocks * (comps == 3 ? 8u : 16u); /* dxtc.cc:276-280,725-733 */
  if (compressor == ICO_COMPRESSOR_ETC) return format == ICO_RGB ? blocks * 8u : 0; /* etc.cc:734-745 */
  return 0;
}

static int compress_common(int compressor, int etc_strategy, int format, uint32_t h, uint32_t w, uint32_t gh,
                           uint32_t gw, uint32_t pad, const uint8_t *buf, uint8_t *out, size_t out_size) {
  if (!buf || !out || h == 0 || w == 0) return 0;
  int comps = (format == ICO_RGB || format == ICO_BGR) ? 3 : 4;
  int swap = (format == ICO_BGR || format == ICO_BGRA); /* compressed_image.h:202-204 */
  int codec;
  if (compressor == ICO_COMPRESSOR_DXTC) codec = comps == 3 ? ICO_DXT1 : ICO_DXT5; /* dxtc.cc:741-749 */
  else if (compressor == ICO_COMPRESSOR_ETC) { if (format != ICO_RGB) return 0; codec = ICO_ETC1; } /* etc.cc:751-754 */
  else return 0;
  if (gh < h) gh = h; /* helper.h:487-488 */
  if (gw < w) gw = w;
  if (out_size != ico_encoded_size(codec, gh, gw)) return 0; /* compressor4x4_helper.cc:34-41 */
  return ico_encode(codec, etc_strategy, comps, swap, h, w, gh, gw, w * (uint32_t)comps + pad, buf, out, 1);
}

int ico_compress(int compressor, int etc_strategy, int format, uint32_t h, uint32_t w, uint32_t pad,
                 const uint8_t *buf, uint8_t *out, size_t out_size) {
  if (compressor == ICO_COMPRESSOR_PVRTC) { /* pvrtc.cc:636-667: format is NOT validated */
    if (!buf || !out || h == 0 || w == 0) return 0;
    if (!is_pow2(w) || !is_pow2(h) || w != h || pad != 0 || w % 8 || h % 4) return 0;
    if (out_size != (size_t)(w * h / 4)) return 0;
    return pvrtc_encode_image(buf, w, h, out);
  }
  return compress_common(compressor, etc_strategy, format, h, w, h, w, pad, buf, out, out_size);
}

int ico_compress_and_pad(int compressor, int etc_strategy, int format, uint32_t h, uint32_t w, uint32_t ph,
                         uint32_t pw, uint32_t pad, const uint8_t *buf, uint8_t *out, size_t out_size) {
  if (compressor == ICO_COMPRESSOR_PVRTC) return 0; /* pvrtc.cc:684-691 */
  return compress_common(compressor, etc_strategy, format, h, w, ph, pw, pad, buf, out, out_size);
}

/* --------------------------------------------------------------- decoders -- */

static void decode_dxt_colors(const uint8_t *blk, int swap, int always4, uint8_t colors[4][3]) {
  /* dxtc.cc:167-192 */
  int c0 = blk[0] + blk[1] * 256, c1 = blk[2] + blk[3] * 256;
  int v[2] = { c0, c1 };
  for (int i = 0; i < 2; ++i) {
    c565_t q = { v[i] >> 11, (v[i] >> 5) & 0x3f, v[i] & 0x1f };
    rgb_t e = maybe_swap(expand565(q), swap);
    colors[i][0] = (uint8_t)e.r; colors[i][1] = (uint8_t)e.g; colors[i][2] = (uint8_t)e.b;
  }
  for (int c = 0; c < 3; ++c) {
    if (c0 == c1) { colors[2][c] = colors[3][c] = colors[1][c]; }
    else if (always4 || c0 > c1) {
      colors[2][c] = (uint8_t)((2 * colors[0][c] + colors[1][c]) / 3);
      colors[3][c] = (uint8_t)((colors[0][c] + 2 * colors[1][c]) / 3);
    } else {
      colors[2][c] = (uint8_t)((colors[0][c] + colors[1][c]) / 2);
      colors[3][c] = 0;
    }
  }
}

static void decode_block(int codec, int swap, const uint8_t *blk, uint8_t px[16][4]) {
  if (codec == ICO_DXT1 || codec == ICO_DXT5) {
    const uint8_t *cb = codec == ICO_DXT5 ? blk + 8 : blk;
    uint8_t colors[4][3], alpha[8];
    uint64_t acodes = 0;
    decode_dxt_colors(cb, swap, codec == ICO_DXT5, colors);
    if (codec == ICO_DXT5) { /* dxtc.cc:195-217 */
      int a0 = blk[0], a1 = blk[1];
      alpha[0] = (uint8_t)a0; alpha[1] = (uint8_t)a1;
      if (a0 > a1) for (int k = 1; k <= 6; ++k) alpha[1 + k] = (uint8_t)(((7 - k) * a0 + k * a1) / 7);
      else { for (int k = 1; k <= 4; ++k) alpha[1 + k] = (uint8_t)(((5 - k) * a0 + k * a1) / 5); alpha[6] = 0; alpha[7] = 255; }
      for (int i = 0; i < 6; ++i) acodes |= (uint64_t)blk[2 + i] << (8 * i);
    }
    for (int p = 0; p < 16; ++p) {
      int code = (cb[4 + p / 4] >> (2 * (p % 4))) & 3;
      px[p][0] = colors[code][0]; px[p][1] = colors[code][1]; px[p][2] = colors[code][2];
      px[p][3] = codec == ICO_DXT5 ? alpha[(acodes >> (3 * p)) & 7] : 255;
    }
    return;
  }
  /* ETC1: etc.cc:198-289 */
  uint32_t hi = (uint32_t)blk[0] << 24 | (uint32_t)blk[1] << 16 | (uint32_t)blk[2] << 8 | blk[3];
  uint32_t lo = (uint32_t)blk[4] << 24 | (uint32_t)blk[5] << 16 | (uint32_t)blk[6] << 8 | blk[7];
  int flip = hi & 1, diff = (hi >> 1) & 1, cw[2] = { (int)((hi >> 5) & 7), (int)((hi >> 2) & 7) };
  rgb_t base[2];
  if (diff) {
    int b5[3] = { (int)((hi >> 27) & 31), (int)((hi >> 19) & 31), (int)((hi >> 11) & 31) };
    int d3[3] = { (int)((hi >> 24) & 7), (int)((hi >> 16) & 7), (int)((hi >> 8) & 7) };
    int s5[3];
    for (int c = 0; c < 3; ++c) s5[c] = b5[c] + (d3[c] >= 4 ? d3[c] - 8 : d3[c]);
    base[0].r = ext5(b5[0]); base[0].g = ext5(b5[1]); base[0].b = ext5(b5[2]);
    base[1].r = ext5(s5[0]); base[1].g = ext5(s5[1]); base[1].b = ext5(s5[2]);
  } else {
    base[0].r = ext4((int)((hi >> 28) & 15)); base[0].g = ext4((int)((hi >> 20) & 15)); base[0].b = ext4((int)((hi >> 12) & 15));
    base[1].r = ext4((int)((hi >> 24) & 15)); base[1].g = ext4((int)((hi >> 16) & 15)); base[1].b = ext4((int)((hi >> 8) & 15));
  }
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) {
      int p = 4 * x + y;
      int k = (int)((lo >> p) & 1) | (int)(((lo >> (p + 16)) & 1) << 1);
      int s = in_subblock(flip, 1, y, x);
      int m = kEtcModifiers[cw[s]][k];
      px[4 * y + x][0] = (uint8_t)clamp255(base[s].r + m);
      px[4 * y + x][1] = (uint8_t)clamp255(base[s].g + m);
      px[4 * y + x][2] = (uint8_t)clamp255(base[s].b + m);
      px[4 * y + x][3] = 255;
    }
}

/* ---- PVRTC1 2bpp decoder: EXTENSION, PARITY UNPINNED.  The reference has no PVRTC decoder
 * (PvrtcCompressor::Decompress returns false, pvrtc.cc:669-672), so there is nothing to pin this against; it is
 * written from the encoder's own rules so that decode(encode(x)) reproduces exactly what the encoder optimised for:
 *   - block word layout and Z order: pvrtc.cc:551-580, :80-86;  colour fields: EncodeColors, pvrtc.cc:356-388,
 *     expanded to 8 bits by bit replication exactly like ApplyBitDepthReduction (pvrtc.cc:93-106);
 *   - A / B images up-sampled with GetInterpolatedColor2BPP / Interpolate4_2BPP (pvrtc.cc:173-237: centre of a block
 *     = the stored colour, toroidal wrap, truncating / 32);
 *   - modulation 0..3 = A, (5A+3B)/8, (3A+5B)/8, B (ApplyModulation, pvrtc.cc:120-144), i.e. weights 0, 3, 5, 8 of B;
 *     1BPP blocks: one bit per pixel, 0 -> A, 1 -> B (CalculateBlockModulationData stores m / 2, pvrtc.cc:465-468);
 *     2BPP blocks: 2 bits for the checkerboard pixels ((x ^ y) & 1 == 0) in raster order; the samples at bit 0
 *     (pixel (0,0)) and bit 20 (pixel (4,2)) keep only their high bit (-> weight 0 or 8), their low bits select the
 *     sub-mode (pvrtc.cc:474-487): bit 0 clear = average of the 4 orthogonal neighbours, else bit 20 set = average of
 *     the vertical pair, clear = of the horizontal pair.  The encoder does not say how the skipped pixels are
 *     reconstructed beyond the format comment (public/pvrtc_compressor.h:31-37); the PVRTC1 rule is used: the
 *     neighbours' weights are averaged with rounding ((a+b+1)/2, (a+b+c+d+2)/4), neighbours wrap toroidally and may
 *     belong to a block of the other kind.  Final colour = ((8 - w) A + w B) / 8 per channel, truncating. */
static const int kPvWeight[4] = { 0, 3, 5, 8 };
static rgba_t pvrtc_unpack_a(uint32_t c) {
  rgba_t o;
  if (c & (1u << 15)) {
    o.r = (uint8_t)ext5((c >> 10) & 31); o.g = (uint8_t)ext5((c >> 5) & 31); o.b = (uint8_t)ext4((c >> 1) & 15); o.a = 255;
  } else {
    uint32_t b3 = (c >> 1) & 7, a3 = (c >> 12) & 7;
    o.r = (uint8_t)ext4((c >> 8) & 15); o.g = (uint8_t)ext4((c >> 4) & 15);
    o.b = (uint8_t)(b3 << 5 | b3 << 2 | b3 >> 1); o.a = (uint8_t)(a3 << 5 | a3 << 2 | a3 >> 1);
  }
  return o;
}
static rgba_t pvrtc_unpack_b(uint32_t c) {
  rgba_t o;
  if (c & (1u << 31)) {
    o.r = (uint8_t)ext5((c >> 26) & 31); o.g = (uint8_t)ext5((c >> 21) & 31); o.b = (uint8_t)ext5((c >> 16) & 31); o.a = 255;
  } else {
    uint32_t a3 = (c >> 28) & 7;
    o.r = (uint8_t)ext4((c >> 24) & 15); o.g = (uint8_t)ext4((c >> 20) & 15); o.b = (uint8_t)ext4((c >> 16) & 15);
    o.a = (uint8_t)(a3 << 5 | a3 << 2 | a3 >> 1);
  }
  return o;
}
static uint32_t pvrtc_z_of(uint32_t bx, uint32_t by) { /* inverse of FromZOrder, pvrtc.cc:80-86 */
  uint32_t z = 0;
  for (int i = 0; i < 16; ++i) z |= ((by >> i) & 1u) << (2 * i) | ((bx >> i) & 1u) << (2 * i + 1);
  return z;
}
/* explicit weight of pixel (x, y) of a block, or -1 where a 2BPP block stores nothing for it */
static int pvrtc_stored_weight(uint32_t data, int two_bpp, uint32_t x, uint32_t y) {
  if (!two_bpp) return ((data >> (8 * y + x)) & 1u) ? 8 : 0;
  if ((x ^ y) & 1u) return -1;
  uint32_t pos = 2 * (4 * y + (x >> 1)), s = (data >> pos) & 3u;
  if (pos == 0 || pos == 20) return (s & 2u) ? 8 : 0;
  return kPvWeight[s];
}
static int pvrtc_decode_image(const uint8_t *blocks, uint32_t n, uint8_t *out) {
  uint32_t bw = n / 8, bh = n / 4, nblocks = bw * bh;
  rgba_t *la = (rgba_t *)malloc(sizeof(rgba_t) * nblocks), *lb = (rgba_t *)malloc(sizeof(rgba_t) * nblocks);
  uint32_t *data = (uint32_t *)malloc(sizeof(uint32_t) * nblocks);
  uint8_t *two = (uint8_t *)malloc(nblocks);
  if (!la || !lb || !data || !two) { free(la); free(lb); free(data); free(two); return 0; }
  for (uint32_t by = 0; by < bh; ++by)
    for (uint32_t bx = 0; bx < bw; ++bx) {
      const uint8_t *p = blocks + 8 * (size_t)pvrtc_z_of(bx, by);
      uint32_t d = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
      uint32_t c = (uint32_t)p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
      data[by * bw + bx] = d; two[by * bw + bx] = (uint8_t)(c & 1u);
      la[by * bw + bx] = pvrtc_unpack_a(c); lb[by * bw + bx] = pvrtc_unpack_b(c);
    }
#define PV_W(X, Y) pvrtc_stored_weight(data[(((Y) & (n - 1)) >> 2) * bw + (((X) & (n - 1)) >> 3)], \
                                       two[(((Y) & (n - 1)) >> 2) * bw + (((X) & (n - 1)) >> 3)], (X) & 7u, (Y) & 3u)
  for (uint32_t y = 0; y < n; ++y)
    for (uint32_t x = 0; x < n; ++x) {
      uint32_t blk = (y >> 2) * bw + (x >> 3);
      int w = PV_W(x, y);
      if (w < 0) {
        int l = PV_W(x - 1, y), r = PV_W(x + 1, y), u = PV_W(x, y - 1), d = PV_W(x, y + 1);
        if (!(data[blk] & 1u)) w = (l + r + u + d + 2) >> 2;
        else if (data[blk] & (1u << 20)) w = (u + d + 1) >> 1;
        else w = (l + r + 1) >> 1;
      }
      rgba_t a = pvrtc_interp(la, n, n, x, y), b = pvrtc_interp(lb, n, n, x, y);
      uint8_t *o = out + 4 * ((size_t)y * n + x);
      o[0] = (uint8_t)(((8 - w) * a.r + w * b.r) >> 3); o[1] = (uint8_t)(((8 - w) * a.g + w * b.g) >> 3);
      o[2] = (uint8_t)(((8 - w) * a.b + w * b.b) >> 3); o[3] = (uint8_t)(((8 - w) * a.a + w * b.a) >> 3);
    }
#undef PV_W
  free(la); free(lb); free(data); free(two);
  return 1;
}

/* PVRTC1 4 bpp decoder (extension, parity unpinned like the encoder above): the inverse of pvrtc4_encode_image -- colours
 * unpacked like the 2 bpp decoder's, A / B up-sampled with pvrtc4_interp, pixel = ((8 - w) A + w B) >> 3 with w = 0, 3, 5, 8
 * (ApplyModulation, pvrtc.cc:120-144).  A block whose mode flag (colour word bit 0) is set uses PVRTC1's punch-through
 * weights 0, 4, 4, 8 with alpha 0 for value 2; the encoder never produces it. */
static int pvrtc4_decode_image(const uint8_t *blocks, uint32_t n, uint8_t *out) {
  uint32_t lw = n / 4, nblocks = lw * lw;
  rgba_t *la = (rgba_t *)malloc(sizeof(rgba_t) * nblocks), *lb = (rgba_t *)malloc(sizeof(rgba_t) * nblocks);
  uint32_t *data = (uint32_t *)malloc(sizeof(uint32_t) * nblocks);
  uint8_t *punch = (uint8_t *)malloc(nblocks);
  if (!la || !lb || !data || !punch) { free(la); free(lb); free(data); free(punch); return 0; }
  for (uint32_t by = 0; by < lw; ++by)
    for (uint32_t bx = 0; bx < lw; ++bx) {
      const uint8_t *p = blocks + 8 * (size_t)pvrtc_z_of(bx, by);
      uint32_t c = (uint32_t)p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
      data[by * lw + bx] = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
      punch[by * lw + bx] = (uint8_t)(c & 1u);
      la[by * lw + bx] = pvrtc_unpack_a(c); lb[by * lw + bx] = pvrtc_unpack_b(c);
    }
  for (uint32_t y = 0; y < n; ++y)
    for (uint32_t x = 0; x < n; ++x) {
      uint32_t blk = (y >> 2) * lw + (x >> 2), m = (data[blk] >> (2 * (4 * (y & 3) + (x & 3)))) & 3u;
      int w = punch[blk] ? (m == 0 ? 0 : m == 3 ? 8 : 4) : kPvWeight[m];
      rgba_t a = pvrtc4_interp(la, n, x, y), b = pvrtc4_interp(lb, n, x, y);
      uint8_t *o = out + 4 * ((size_t)y * n + x);
      o[0] = (uint8_t)(((8 - w) * a.r + w * b.r) >> 3); o[1] = (uint8_t)(((8 - w) * a.g + w * b.g) >> 3);
      o[2] = (uint8_t)(((8 - w) * a.b + w * b.b) >> 3); o[3] = (uint8_t)(((8 - w) * a.a + w * b.a) >> 3);
      if (punch[blk] && m == 2) o[3] = 0;
    }
  free(la); free(lb); free(data); free(punch);
  return 1;
}

/* helper.h:218-262 */
int ico_decode(int codec, int swap, uint32_t h, uint32_t w, uint32_t pad, const uint8_t *blocks, uint8_t *out) {
  if (!blocks || !out || h == 0 || w == 0) return 0;
  if (codec == ICO_PVRTC2) { /* extension, see above; same size rules as PvrtcCompressor::Compress, pvrtc.cc:636-650 */
    if (h != w || !is_pow2(w) || w < 8 || pad != 0) return 0;
    return pvrtc_decode_image(blocks, w, out);
  }
  if (codec == ICO_PVRTC4) { /* extension */
    if (h != w || !is_pow2(w) || w < 8 || pad != 0) return 0;
    return pvrtc4_decode_image(blocks, w, out);
  }
  if (codec != ICO_DXT1 && codec != ICO_DXT5 && codec != ICO_ETC1) return 0;
  int comps = codec == ICO_DXT5 ? 4 : 3;
  size_t bb = block_bytes(codec), stride = (size_t)w * (size_t)comps + pad;
  uint32_t rows = nblk(h), cols = nblk(w);
  for (uint32_t br = 0; br < rows; ++br)
    for (uint32_t bc = 0; bc < cols; ++bc) {
      uint8_t px[16][4];
      decode_block(codec, swap, blocks + ((size_t)br * cols + bc) * bb, px);
      for (uint32_t y = 0; y < 4 && br * 4 + y < h; ++y)
        for (uint32_t x = 0; x < 4 && bc * 4 + x < w; ++x)
          memcpy(out + (size_t)(br * 4 + y) * stride + (size_t)(bc * 4 + x) * (size_t)comps, px[4 * y + x], (size_t)comps);
    }
  return 1;
}

/* --------------------------------------- compressed-domain operations (8f rows 2-4) -- */

static int fmt_comps(int format) { return (format == ICO_RGB || format == ICO_BGR) ? 3 : 4; }
static int block_codec(int compressor, int format) {
  if (compressor == ICO_COMPRESSOR_ETC) return ICO_ETC1;
  return fmt_comps(format) == 3 ? ICO_DXT1 : ICO_DXT5;
}

/* etc.cc:595-617 (CreateSolidBlock): differential mode, 5-bit base = color >> 3, zero difference, codewords 0,
 * indices 0 (the "adjusted_color" computed there is unused). */
static void etc_solid_block(const uint8_t color[3], uint8_t out[8]) {
  uint32_t hi = 2u | (uint32_t)(color[0] >> 3) << 27 | (uint32_t)(color[1] >> 3) << 19 | (uint32_t)(color[2] >> 3) << 11;
  out[0] = (uint8_t)(hi >> 24); out[1] = (uint8_t)(hi >> 16); out[2] = (uint8_t)(hi >> 8); out[3] = (uint8_t)hi;
  out[4] = out[5] = out[6] = out[7] = 0;
}

int ico_create_solid(int compressor, int format, uint32_t h, uint32_t w, const uint8_t *color, uint8_t *out) {
  if (compressor == ICO_COMPRESSOR_PVRTC) return 0;                        /* pvrtc.cc:693-698 */
  if (compressor == ICO_COMPRESSOR_ETC && format != ICO_RGB) return 0;     /* etc.cc:805-806 */
  uint8_t blk[16];
  size_t bb;
  if (compressor == ICO_COMPRESSOR_ETC) { etc_solid_block(color, blk); bb = 8; }
  else {
    /* dxtc.cc:42-49,77-82: c0 = c1 = Quantize565(color) with NO red/blue swap, all index bits 0 */
    rgb_t c = { color[0], color[1], color[2] };
    int p = pack565(quant565(c));
    uint8_t d1[8] = { (uint8_t)(p & 0xff), (uint8_t)(p >> 8), (uint8_t)(p & 0xff), (uint8_t)(p >> 8), 0, 0, 0, 0 };
    if (fmt_comps(format) == 3) { memcpy(blk, d1, 8); bb = 8; }
    else { memset(blk, 0, 8); blk[0] = blk[1] = color[3]; memcpy(blk + 8, d1, 8); bb = 16; }
  }
  size_t n = (size_t)nblk(h) * nblk(w);
  for (size_t i = 0; i < n; ++i) memcpy(out + i * bb, blk, bb);
  return 1;
}

int ico_copy_subimage(int compressor, int format, uint32_t ch, uint32_t cw, const uint8_t *blocks, uint32_t row,
                      uint32_t col, uint32_t h, uint32_t w, uint8_t *out) {
  if (compressor == ICO_COMPRESSOR_PVRTC) return 0;
  if (compressor == ICO_COMPRESSOR_ETC && format != ICO_RGB) return 0;     /* etc.cc:719-722 (IsValidCompressedImage) */
  if (row % 4 || col % 4 || h % 4 || w % 4 || row > ch || col > cw || row + h > ch || col + w > cw) return 0;
  size_t bb = block_bytes(block_codec(compressor, format));
  uint32_t ocols = nblk(cw), scols = nblk(w), srows = nblk(h);
  for (uint32_t r = 0; r < srows; ++r)
    memcpy(out + (size_t)r * scols * bb, blocks + ((size_t)(row / 4 + r) * ocols + col / 4) * bb, (size_t)scols * bb);
  return 1;
}

/* DXT5 alpha codes as a 48-bit integer */
static uint64_t get_codes48(const uint8_t *b) { uint64_t v = 0; for (int i = 0; i < 6; ++i) v |= (uint64_t)b[2 + i] << (8 * i); return v; }
static void put_codes48(uint8_t *b, uint64_t v) { for (int i = 0; i < 6; ++i) b[2 + i] = (uint8_t)(v >> (8 * i)); }

/* kind: 0 = column pad (replicate pixel column 3), 1 = row pad (replicate pixel row 3), 2 = corner (pixel (3,3)) */
static void pad_block(int codec, int etc_strategy, int kind, const uint8_t *src, uint8_t *dst) {
  if (codec == ICO_ETC1) { /* etc.cc:645-698: decode, replicate, re-encode (corner: solid block) */
    uint8_t px[16][4];
    decode_block(ICO_ETC1, 0, src, px);
    if (kind == 2) { etc_solid_block(px[15], dst); return; }
    block4x4_t b;
    b.one_pixel = 0;
    for (int y = 0; y < 4; ++y)
      for (int x = 0; x < 4; ++x) {
        const uint8_t *s = kind == 0 ? px[4 * y + 3] : px[12 + x];
        b.px[4 * y + x].r = s[0]; b.px[4 * y + x].g = s[1]; b.px[4 * y + x].b = s[2];
        b.alpha[4 * y + x] = 255;
      }
    encode_etc1_block(&b, etc_strategy, dst);
    return;
  }
  /* dxtc.cc:594-696: same endpoints, index bits edited */
  size_t bb = block_bytes(codec);
  memcpy(dst, src, bb);
  uint8_t *cb = dst + (codec == ICO_DXT5 ? 8 : 0);
  const uint8_t *sb = src + (codec == ICO_DXT5 ? 8 : 0);
  for (int r = 0; r < 4; ++r) {
    if (kind == 0) cb[4 + r] = (uint8_t)(((sb[4 + r] >> 6) & 3) * 0x55);
    else if (kind == 1) cb[4 + r] = sb[7];
    else cb[4 + r] = (uint8_t)(((sb[7] >> 6) & 3) * 0x55);
  }
  if (codec == ICO_DXT5) {
    uint64_t c = get_codes48(src), o = 0;
    for (int p = 0; p < 16; ++p) {
      int sp = kind == 0 ? 4 * (p / 4) + 3 : kind == 1 ? 12 + (p % 4) : 15;
      o |= ((c >> (3 * sp)) & 7u) << (3 * p);
    }
    put_codes48(dst, o);
  }
}

int ico_pad(int compressor, int etc_strategy, int format, uint32_t ch, uint32_t cw, const uint8_t *blocks,
            uint32_t ph, uint32_t pw, uint8_t *out) {
  if (compressor == ICO_COMPRESSOR_PVRTC) return 0;
  int codec = block_codec(compressor, format);
  size_t bb = block_bytes(codec);
  uint32_t orows = nblk(ch), ocols = nblk(cw);
  if (ch >= ph && cw >= pw) { memcpy(out, blocks, (size_t)orows * ocols * bb); return 2; } /* helper.h:404-408 */
  uint32_t prows = nblk(ph), pcols = nblk(pw);
  if (prows < orows || pcols < ocols) return 0; /* the reference overruns its buffer here */
  for (uint32_t r = 0; r < prows; ++r)
    for (uint32_t c = 0; c < pcols; ++c) {
      uint8_t *o = out + ((size_t)r * pcols + c) * bb;
      if (r < orows && c < ocols) memcpy(o, blocks + ((size_t)r * ocols + c) * bb, bb);
      else if (r < orows) pad_block(codec, etc_strategy, 0, blocks + ((size_t)r * ocols + ocols - 1) * bb, o);
      else if (c < ocols) pad_block(codec, etc_strategy, 1, blocks + ((size_t)(orows - 1) * ocols + c) * bb, o);
      else pad_block(codec, etc_strategy, 2, blocks + ((size_t)(orows - 1) * ocols + ocols - 1) * bb, o);
    }
  return 1;
}

static void encode_any_block(int codec, int etc_strategy, const block4x4_t *b, uint8_t *o) {
  if (codec == ICO_DXT1) encode_dxt1_block(b, 0, 0, o);
  else if (codec == ICO_DXT5) { encode_dxt5_alpha(b, o); encode_dxt1_block(b, 0, 1, o + 8); }
  else encode_etc1_block(b, etc_strategy, o);
}

/* StoreDownsampledPixels4x4 (pixel4x4.h:152-162): 2x2 averages of src (a decoded 4x4) into quadrant (tr, tc) of dst */
static void store_downsampled(uint8_t src[16][4], int tr, int tc, block4x4_t *dst) {
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 2; ++c) {
      int v[4];
      for (int ch = 0; ch < 4; ++ch)
        v[ch] = (src[4 * (2 * r) + 2 * c][ch] + src[4 * (2 * r) + 2 * c + 1][ch] + src[4 * (2 * r + 1) + 2 * c][ch] +
                 src[4 * (2 * r + 1) + 2 * c + 1][ch]) / 4;
      int p = 4 * (tr + r) + tc + c;
      dst->px[p].r = v[0]; dst->px[p].g = v[1]; dst->px[p].b = v[2]; dst->alpha[p] = v[3];
    }
}

int ico_downsample(int compressor, int etc_strategy, int format, uint32_t uh, uint32_t uw, const uint8_t *blocks,
                   uint8_t *out) {
  if (compressor == ICO_COMPRESSOR_PVRTC) return 0;
  int codec = block_codec(compressor, format);
  size_t bb = block_bytes(codec);
  int orows = (int)nblk(uh), ocols = (int)nblk(uw);
  if ((orows > 1 && orows % 2) || (ocols > 1 && ocols % 2)) return 0; /* helper.h:281-284 */
  int drows = orows / 2, dcols = ocols / 2;
  uint8_t px[16][4];
  block4x4_t b;
  b.one_pixel = 0;
  if (orows > 1 && ocols > 1) {
    for (int r = 0; r < drows; ++r)
      for (int c = 0; c < dcols; ++c) {
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j) {
            decode_block(codec, 0, blocks + ((size_t)(2 * r + i) * ocols + 2 * c + j) * bb, px);
            store_downsampled(px, 2 * i, 2 * j, &b);
          }
        encode_any_block(codec, etc_strategy, &b, out + ((size_t)r * dcols + c) * bb);
      }
  } else if (orows > 1) { /* one block column: each source block fills two quadrant columns */
    for (int r = 0; r < drows; ++r) {
      for (int i = 0; i < 2; ++i) {
        decode_block(codec, 0, blocks + (size_t)(2 * r + i) * bb, px);
        store_downsampled(px, 2 * i, 0, &b);
        store_downsampled(px, 2 * i, 2, &b);
      }
      encode_any_block(codec, etc_strategy, &b, out + (size_t)r * bb);
    }
  } else if (ocols > 1) {
    for (int c = 0; c < dcols; ++c) {
      for (int j = 0; j < 2; ++j) {
        decode_block(codec, 0, blocks + (size_t)(2 * c + j) * bb, px);
        store_downsampled(px, 0, 2 * j, &b);
        store_downsampled(px, 2, 2 * j, &b);
      }
      encode_any_block(codec, etc_strategy, &b, out + (size_t)c * bb);
    }
  } else { /* a single block: replicate up to 4x4 first (helper.h:338-387) */
    if (uh == 3 || uw == 3) return 0;
    decode_block(codec, 0, blocks, px);
    if (uw == 1) for (int r = 0; r < 4; ++r) for (int x = 1; x < 4; ++x) memcpy(px[4 * r + x], px[4 * r], 4);
    else if (uw == 2) for (int r = 0; r < 4; ++r) { memcpy(px[4 * r + 2], px[4 * r], 4); memcpy(px[4 * r + 3], px[4 * r + 1], 4); }
    if (uh == 1) for (int c = 0; c < 4; ++c) for (int y = 1; y < 4; ++y) memcpy(px[4 * y + c], px[c], 4);
    else if (uh == 2) for (int c = 0; c < 4; ++c) { memcpy(px[8 + c], px[c], 4); memcpy(px[12 + c], px[4 + c], 4); }
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) store_downsampled(px, 2 * r, 2 * c, &b);
    encode_any_block(codec, etc_strategy, &b, out);
  }
  return 1;
}

void ico_transcode_dxt1_to_etc1(uint8_t *blocks, size_t n_bytes) {
  for (size_t i = 0; i + 8 <= n_bytes; i += 8) {
    uint8_t px[16][4];
    block4x4_t b;
    decode_block(ICO_DXT1, 0, blocks + i, px);
    b.one_pixel = 0;
    for (int p = 0; p < 16; ++p) { b.px[p].r = px[p][0]; b.px[p].g = px[p][1]; b.px[p].b = px[p][2]; b.alpha[p] = 255; }
    encode_etc1_block(&b, ICO_ETC_HEURISTIC, blocks + i);
  }
}
