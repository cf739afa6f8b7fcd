// TEST INFRASTRUCTURE ONLY.  Compiles the per-block device math of
// image-compression_amd/csrc/*_block.h for the HOST (g++ -DICAMD_HOST_EMULATION: the gfx950
// instruction wrappers in ic_device.h are replaced by plain-C equivalents) so that the exact
// kernel arithmetic can be checked against the oracle in the CPU-only test tier.
// Never linked into libic_amd.so; the product has no CPU path.
#ifndef ICAMD_HOST_EMULATION
#error "build with -DICAMD_HOST_EMULATION"
#endif
#include <algorithm>
#include <cstring>

#include "dxt_block.h"
#include "etc1_block.h"
#include "pvrtc_block.h"
#include "decode_block.h"
#include "blockops_block.h"

using namespace icamd;

extern "C" int emul_encode(int codec, int strategy, int comps, int swap, uint32_t h, uint32_t w, uint32_t gh,
                           uint32_t gw, uint32_t stride, const uint8_t *src, uint8_t *out) {
  if (codec == 3) return emul_pvrtc2(src, w, out);
  if (codec == 4) return emul_pvrtc4(src, w, out);
  const uint32_t rows = (std::max(h, gh) + 3) / 4, cols = (std::max(w, gw) + 3) / 4;
  for (uint32_t br = 0; br < rows; ++br)
    for (uint32_t bc = 0; bc < cols; ++bc) {
      uint32_t px[16];
      if (comps == 4) load_block<4>(src, h, w, stride, br * 4, bc * 4, px);
      else load_block<3>(src, h, w, stride, br * 4, bc * 4, px);
      uint8_t *o = out + ((size_t)br * cols + bc) * (codec == 1 ? 16 : 8);
      BlockStash stash;
      if (codec == 0) {
        Out8 c = encode_dxt_color_block(px, swap != 0, false, stash);
        memcpy(o, &c, 8);
      } else if (codec == 1) {
        const bool one_pixel = bc * 4 >= w && br * 4 >= h;
        Out8 a = encode_dxt5_alpha_block(px, one_pixel);
        Out8 c = encode_dxt_color_block(px, swap != 0, true, stash);
        memcpy(o, &a, 8);
        memcpy(o + 8, &c, 8);
      } else {
        // like the kernel: the instantiation with the mixed tier for "busy" blocks, without it otherwise; bits 8 / 9 of
        // `strategy` force one of the two for every block (both must produce the reference's bytes on any content)
        const uint32_t st = (uint32_t)strategy & 0xffu;
        const bool tier = (strategy & 0x100) ? true : (strategy & 0x200) ? false : (st != 3u && etc1_busy_wave(px));
        // the instantiations the kernels use: <tier, no pruning> for busy waves, <tier, pruning> for calm ones; bit 10
        // additionally selects the plain <no tier, pruning> form (what the block operations use)
        // (a "wave" is one block here: the one-colour form is taken for every constant block unless an instantiation is forced)
        const bool constant = (strategy & 0x700) == 0 && st != 3u && etc1_constant_block(px, etc1_block_spread(px));
        Out8 c = constant ? encode_etc1_constant_block(px[0], st)
                 : (strategy & 0x400) ? encode_etc1_block<false, true>(px, st)
                 : tier ? encode_etc1_block<true, false>(px, st) : encode_etc1_block<true, true>(px, st);
        if (st == 2u && !constant) {  // the four-lanes-per-block form (small launches, Pad border) must give the same bytes
          const Out8 q = encode_etc1_block_quad(px);
          if (q.lo != c.lo || q.hi != c.hi) { c.lo = 0xbad0bad0u; c.hi = q.lo ^ q.hi; }
        }
        memcpy(o, &c, 8);
      }
    }
  return 1;
}

static int emul_decode_pvrtc2(uint32_t n, const uint8_t *blocks, uint8_t *out) {
  const uint32_t bw = n / 8, bh = n / 4;
  const uint32_t *words = reinterpret_cast<const uint32_t *>(blocks);
  for (uint32_t by = 0; by < bh; ++by)
    for (uint32_t bx = 0; bx < bw; ++bx) {
      uint32_t mod[9], col[9], px[32];
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const uint32_t z = pvrtc_z_index((bx + bw + dx) % bw, (by + bh + dy) % bh);
          mod[3 * (dy + 1) + dx + 1] = words[2 * z];
          col[3 * (dy + 1) + dx + 1] = words[2 * z + 1];
        }
      decode_pvrtc2_block(mod, col, px);
      for (int i = 0; i < 32; ++i) memcpy(out + 4 * ((size_t)(by * 4 + i / 8) * n + bx * 8 + i % 8), &px[i], 4);
    }
  return 1;
}

static int emul_decode_pvrtc4(uint32_t n, const uint8_t *blocks, uint8_t *out) {
  const uint32_t bw = n / 4, bh = n / 4;
  const uint32_t *words = reinterpret_cast<const uint32_t *>(blocks);
  for (uint32_t by = 0; by < bh; ++by)
    for (uint32_t bx = 0; bx < bw; ++bx) {
      uint32_t col[9], px[16];
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
          col[3 * (dy + 1) + dx + 1] = words[2 * pvrtc_z_index((bx + bw + dx) % bw, (by + bh + dy) % bh) + 1];
      decode_pvrtc4_block(words[2 * pvrtc_z_index(bx, by)], col, px);
      for (int i = 0; i < 16; ++i) memcpy(out + 4 * ((size_t)(by * 4 + i / 4) * n + bx * 4 + i % 4), &px[i], 4);
    }
  return 1;
}

extern "C" int emul_decode(int codec, int swap, uint32_t h, uint32_t w, uint32_t pad, const uint8_t *blocks, uint8_t *out) {
  if (codec == 3) return emul_decode_pvrtc2(w, blocks, out);
  if (codec == 4) return emul_decode_pvrtc4(w, blocks, out);
  const int comps = codec == 1 ? 4 : 3;
  const uint32_t rows = (h + 3) / 4, cols = (w + 3) / 4;
  const size_t stride = (size_t)w * comps + pad;
  for (uint32_t br = 0; br < rows; ++br)
    for (uint32_t bc = 0; bc < cols; ++bc) {
      const uint32_t *b = reinterpret_cast<const uint32_t *>(blocks + ((size_t)br * cols + bc) * (codec == 1 ? 16 : 8));
      uint32_t px[16];
      if (codec == 1) { decode_dxt_colors(b[2], b[3], swap != 0, true, px); decode_dxt5_alpha(b[0], b[1], px); }
      else if (codec == 0) decode_dxt_colors(b[0], b[1], swap != 0, false, px);
      else decode_etc1(b[0], b[1], px);
      {  // the row decoders the kernels use for whole blocks (palette planes) must give the same bytes
        uint32_t rows[4][4];
        if (codec == 1) decode_block_rows<1>(b, swap != 0, rows);
        else if (codec == 0) decode_block_rows<0>(b, swap != 0, rows);
        else decode_block_rows<2>(b, false, rows);
        for (int y = 0; y < 4; ++y) {
          uint8_t want[16];
          for (int x = 0; x < 4; ++x) memcpy(want + x * comps, &px[4 * y + x], comps);
          if (memcmp(want, rows[y], 4 * comps) != 0) return 0;
        }
      }
      for (uint32_t y = 0; y < 4 && br * 4 + y < h; ++y)
        for (uint32_t x = 0; x < 4 && bc * 4 + x < w; ++x)
          memcpy(out + (br * 4 + y) * stride + (size_t)(bc * 4 + x) * comps, &px[4 * y + x], comps);
    }
  return 1;
}

// ---- compressed-domain operations through the device per-block math (blockops_block.h)
template <int CODEC>
static void emul_pad_t(int strategy, uint32_t orows, uint32_t ocols, uint32_t prows, uint32_t pcols, const uint8_t *in, uint8_t *out) {
  const int W = CODEC == 1 ? 4 : 2;
  for (uint32_t r = 0; r < prows; ++r)
    for (uint32_t c = 0; c < pcols; ++c) {
      const bool in_rows = r < orows, in_cols = c < ocols;
      const uint32_t *s = reinterpret_cast<const uint32_t *>(in) + ((size_t)(in_rows ? r : orows - 1) * ocols + (in_cols ? c : ocols - 1)) * W;
      uint32_t *d = reinterpret_cast<uint32_t *>(out) + ((size_t)r * pcols + c) * W;
      if (in_rows && in_cols) { memcpy(d, s, W * 4); continue; }
      const int kind = in_rows ? kPadColumn : (in_cols ? kPadRow : kPadCorner);
      if (CODEC == 2) {
        Out8 o = etc1_pad_block(s[0], s[1], kind, (uint32_t)strategy); d[0] = o.lo; d[1] = o.hi;
        if (strategy == 2) {  // the four-lanes-per-block form of the Pad border kernel must give the same bytes
          const Out8 q = etc1_pad_block_quad(s[0], s[1], kind);
          if (q.lo != o.lo || q.hi != o.hi) { d[0] = 0xbad0bad0u; d[1] = q.lo ^ q.hi; }
        }
      }
      else if (CODEC == 0) { d[0] = s[0]; d[1] = dxt_pad_color_bits(s[1], kind); }
      else {
        uint32_t lo24 = s[0] >> 16 | (s[1] & 0xffu) << 16, hi24 = s[1] >> 8;
        dxt5_pad_alpha_codes(lo24, hi24, kind);
        d[0] = (s[0] & 0xffffu) | lo24 << 16; d[1] = lo24 >> 16 | hi24 << 8; d[2] = s[2]; d[3] = dxt_pad_color_bits(s[3], kind);
      }
    }
}
extern "C" int emul_pad(int codec, int strategy, uint32_t ch, uint32_t cw, uint32_t ph, uint32_t pw, const uint8_t *in, uint8_t *out) {
  const uint32_t orows = (ch + 3) / 4, ocols = (cw + 3) / 4, prows = (ph + 3) / 4, pcols = (pw + 3) / 4;
  if (prows < orows || pcols < ocols) return 0;
  if (codec == 0) emul_pad_t<0>(strategy, orows, ocols, prows, pcols, in, out);
  else if (codec == 1) emul_pad_t<1>(strategy, orows, ocols, prows, pcols, in, out);
  else emul_pad_t<2>(strategy, orows, ocols, prows, pcols, in, out);
  return 1;
}

template <int CODEC>
static int emul_downsample_t(int strategy, uint32_t uh, uint32_t uw, const uint8_t *in, uint8_t *out) {
  const int W = CODEC == 1 ? 4 : 2;
  const uint32_t orows = (uh + 3) / 4, ocols = (uw + 3) / 4;
  if ((orows > 1 && orows % 2) || (ocols > 1 && ocols % 2)) return 0;
  if (orows == 1 && ocols == 1 && (uh == 3 || uw == 3)) return 0;
  const uint32_t drows = ((uh + 1) / 2 + 3) / 4, dcols = ((uw + 1) / 2 + 3) / 4;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(in);
  for (uint32_t r = 0; r < drows; ++r)
    for (uint32_t c = 0; c < dcols; ++c) {
      uint32_t px[16], tmp[16];
      if (orows > 1 && ocols > 1) {
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
          decode_any<CODEC>(src + ((size_t)(2 * r + i) * ocols + 2 * c + j) * W, tmp);
          store_downsampled(tmp, 2 * i, 2 * j, px);
        }
        {  // what the kernel runs for 2x2 grids (palette planes + quad selectors) must give the same pixels
          const uint32_t *const s4[2][2] = {
              { src + ((size_t)(2 * r) * ocols + 2 * c) * W, src + ((size_t)(2 * r) * ocols + 2 * c + 1) * W },
              { src + ((size_t)(2 * r + 1) * ocols + 2 * c) * W, src + ((size_t)(2 * r + 1) * ocols + 2 * c + 1) * W } };
          uint32_t fast[16];
          if (CODEC == 2) etc1_downsample_2x2(s4, fast);
          else dxt_downsample_2x2<CODEC == 1 ? 1 : 0>(s4, fast);
          for (int i = 0; i < 16; ++i)
            if ((fast[i] ^ px[i]) & (CODEC == 1 ? 0xffffffffu : 0x00ffffffu)) return 0;
          memcpy(px, fast, sizeof(px));
        }
      } else if (orows > 1) {
        for (int i = 0; i < 2; ++i) { decode_any<CODEC>(src + (size_t)(2 * r + i) * W, tmp); store_downsampled(tmp, 2 * i, 0, px); store_downsampled(tmp, 2 * i, 2, px); }
      } else if (ocols > 1) {
        for (int j = 0; j < 2; ++j) { decode_any<CODEC>(src + (size_t)(2 * c + j) * W, tmp); store_downsampled(tmp, 0, 2 * j, px); store_downsampled(tmp, 2, 2 * j, px); }
      } else {
        decode_any<CODEC>(src, tmp);
        if (uw == 1) for (int y = 0; y < 4; ++y) tmp[4 * y + 1] = tmp[4 * y + 2] = tmp[4 * y + 3] = tmp[4 * y];
        else if (uw == 2) for (int y = 0; y < 4; ++y) { tmp[4 * y + 2] = tmp[4 * y]; tmp[4 * y + 3] = tmp[4 * y + 1]; }
        if (uh == 1) for (int x = 0; x < 4; ++x) tmp[4 + x] = tmp[8 + x] = tmp[12 + x] = tmp[x];
        else if (uh == 2) for (int x = 0; x < 4; ++x) { tmp[8 + x] = tmp[x]; tmp[12 + x] = tmp[4 + x]; }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) store_downsampled(tmp, 2 * i, 2 * j, px);
      }
      BlockStash stash;
      uint32_t o[4];
      encode_any<CODEC>(px, (uint32_t)strategy, stash, o);
      if (CODEC == 2 && strategy == 2) {  // small grids re-encode with four lanes per block: the same bytes
        const Out8 q = encode_etc1_block_quad(px);
        if (q.lo != o[0] || q.hi != o[1]) return 0;
      }
      memcpy(reinterpret_cast<uint32_t *>(out) + ((size_t)r * dcols + c) * W, o, W * 4);
    }
  return 1;
}
extern "C" int emul_downsample(int codec, int strategy, uint32_t uh, uint32_t uw, const uint8_t *in, uint8_t *out) {
  if (codec == 0) return emul_downsample_t<0>(strategy, uh, uw, in, out);
  if (codec == 1) return emul_downsample_t<1>(strategy, uh, uw, in, out);
  return emul_downsample_t<2>(strategy, uh, uw, in, out);
}
extern "C" void emul_transcode(uint8_t *blocks, size_t n_bytes) {
  for (size_t i = 0; i + 8 <= n_bytes; i += 8) {
    uint32_t *b = reinterpret_cast<uint32_t *>(blocks + i), px[16];
    decode_dxt_colors(b[0], b[1], false, false, px);
    const Out8 generic = encode_etc1_block(px, 3u);
    Out8 o = transcode_dxt1_block_to_etc1(b[0], b[1]);  // what the kernel runs (palette domain) must agree block by block
    if (o.lo != generic.lo || o.hi != generic.hi) { o.lo = 0xdeadbeefu; o.hi = (uint32_t)i; }
    b[0] = o.lo; b[1] = o.hi;
  }
}
