// TEST INFRASTRUCTURE ONLY ("sanitising = run the host/CPU restatement under ASan/UBSan").
// One executable, built by tests/test_sanitizers.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all -DICAMD_HOST_EMULATION ...
// from oracle/ic_oracle.c (the plain-C restatement of the reference) and tests/host_emul/emul.cc (the device per-block
// math compiled for the host).  It drives every entry point of both over ragged geometries with exactly-sized heap
// buffers -- so any out-of-bounds read / write, misaligned or overflowing arithmetic aborts the run -- and cross-checks
// the two restatements against each other on the way.  Exit code 0 = clean.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ic_oracle.h"

extern "C" {
int emul_encode(int codec, int strategy, int comps, int swap, uint32_t h, uint32_t w, uint32_t gh, uint32_t gw,
                uint32_t stride, const uint8_t *src, uint8_t *out);
int emul_decode(int codec, int swap, uint32_t h, uint32_t w, uint32_t pad, const uint8_t *blocks, uint8_t *out);
int emul_pad(int codec, int strategy, uint32_t ch, uint32_t cw, uint32_t ph, uint32_t pw, const uint8_t *in, uint8_t *out);
int emul_downsample(int codec, int strategy, uint32_t uh, uint32_t uw, const uint8_t *in, uint8_t *out);
void emul_transcode(uint8_t *blocks, size_t n_bytes);
}

static uint32_t rng_state = 0x1234abcdu;
static uint32_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
  return rng_state;
}
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

// content kinds: 0 noise, 1 smooth, 2 flat tiles, 3 extremes (0/1/254/255), 4 constant
static std::vector<uint8_t> make_image(uint32_t h, uint32_t w, int comps, uint32_t stride, int kind) {
  // exactly the bytes the reference addresses: no padding after the last row (pixel4x4.h:47-48)
  std::vector<uint8_t> img((size_t)(h - 1) * stride + (size_t)w * comps);
  for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)rnd();  // padding bytes are random too
  const uint8_t solid[4] = { (uint8_t)rnd(), (uint8_t)rnd(), (uint8_t)rnd(), (uint8_t)rnd() };
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x)
      for (int c = 0; c < comps; ++c) {
        uint8_t v;
        switch (kind) {
          case 0: v = (uint8_t)rnd(); break;
          case 1: v = (uint8_t)((c == 3 ? (rnd() % 8 ? 255 : rnd()) : (255u * (c == 1 ? y : x) / (c == 1 ? h : w) + rnd() % 32)) & 255u); break;
          case 2: v = (uint8_t)((x / 8 * 37u + y / 8 * 91u + c * 53u) & 255u); break;
          case 3: { static const uint8_t e[4] = { 0, 1, 254, 255 }; v = e[rnd() & 3]; break; }
          default: v = solid[c]; break;
        }
        img[(size_t)y * stride + (size_t)x * comps + c] = v;
      }
  return img;
}

static size_t block_bytes(int codec) { return codec == ICO_DXT5 ? 16 : 8; }

int main() {
  static const uint32_t dims[][2] = { { 1, 1 }, { 4, 4 }, { 5, 3 }, { 9, 2 }, { 7, 13 }, { 16, 16 }, { 30, 30 }, { 33, 17 }, { 64, 8 } };
  // ---- DXT1 / DXT5 / ETC1: encode (all formats / strategies / pad grids / row paddings), decode, pad, downsample
  for (size_t d = 0; d < sizeof dims / sizeof dims[0]; ++d)
    for (int codec = 0; codec < 3; ++codec)
      for (int comps = 3; comps <= 4; ++comps) {
        if (codec == ICO_DXT5 && comps == 3) continue;
        for (int kind = 0; kind < 5; ++kind) {
          const uint32_t h = dims[d][0], w = dims[d][1];
          const uint32_t pad = (kind & 1) ? 0u : 1u + rnd() % 7u, stride = w * comps + pad;
          const std::vector<uint8_t> img = make_image(h, w, comps, stride, kind);
          const int swap = (int)(rnd() & 1u) && codec != ICO_ETC1;
          const int nstrat = codec == ICO_ETC1 ? 4 : 1;
          for (int st = 0; st < nstrat; ++st) {
            const int strategy = codec == ICO_ETC1 ? st : 2;
            const uint32_t gh = h + (kind == 2 ? rnd() % 9u : 0u), gw = w + (kind == 2 ? rnd() % 9u : 0u);
            const size_t n = ico_encoded_size(codec, gh, gw);
            std::vector<uint8_t> a(n), b(n);
            CHECK(ico_encode(codec, strategy, comps, swap, h, w, gh, gw, stride, img.data(), a.data(), 1) == 1, "ico_encode");
            CHECK(emul_encode(codec, strategy, comps, swap, h, w, gh, gw, stride, img.data(), b.data()) == 1, "emul_encode");
            CHECK(a == b, "encode mismatch codec %d comps %d kind %d %ux%u grid %ux%u strategy %d", codec, comps, kind, h, w, gh, gw, strategy);
            if (st != nstrat - 1) continue;
            // decode the (gh, gw) grid back, with and without row padding
            const int dcomps = codec == ICO_DXT5 ? 4 : 3;
            for (uint32_t dpad = 0; dpad <= 5; dpad += 5) {
              std::vector<uint8_t> pa((size_t)gh * ((size_t)gw * dcomps + dpad), 0), pb(pa.size(), 0);
              CHECK(ico_decode(codec, swap, gh, gw, dpad, a.data(), pa.data()) == 1, "ico_decode");
              CHECK(emul_decode(codec, swap, gh, gw, dpad, a.data(), pb.data()) == 1, "emul_decode");
              CHECK(pa == pb, "decode mismatch codec %d %ux%u pad %u", codec, gh, gw, dpad);
            }
            // the compressed-domain operations of the reference's own formats only (DXT1 <- 3 bytes, DXT5, ETC1 <- kRGB)
            const bool native = (codec == ICO_DXT1 && comps == 3) || codec == ICO_DXT5 || (codec == ICO_ETC1 && comps == 3);
            if (!native) continue;
            const int compressor = codec == ICO_ETC1 ? ICO_COMPRESSOR_ETC : ICO_COMPRESSOR_DXTC;
            const int format = codec == ICO_DXT5 ? (swap ? ICO_BGRA : ICO_RGBA) : (swap ? ICO_BGR : ICO_RGB);
            const uint32_t ch = (gh + 3) / 4 * 4, cw = (gw + 3) / 4 * 4;
            const uint32_t ph = ch + 4 * (rnd() % 3u), pw = cw + 4 * (rnd() % 3u);
            if (ph > ch || pw > cw) {
              const size_t pn = (size_t)(ph / 4) * (pw / 4) * block_bytes(codec);
              std::vector<uint8_t> oa(pn), ob(pn);
              const int ra = ico_pad(compressor, strategy, format, ch, cw, a.data(), ph, pw, oa.data());
              const int rb = emul_pad(codec, strategy, ch, cw, ph, pw, a.data(), ob.data());
              CHECK(ra == 1 && rb == 1 && oa == ob, "pad mismatch codec %d %ux%u -> %ux%u (%d %d)", codec, ch, cw, ph, pw, ra, rb);
            }
            {
              const uint32_t dh = (gh + 1) / 2, dw = (gw + 1) / 2;
              const size_t dn = (size_t)((dh + 3) / 4) * ((dw + 3) / 4) * block_bytes(codec);
              std::vector<uint8_t> oa(dn), ob(dn);
              const int ra = ico_downsample(compressor, strategy, format, gh, gw, a.data(), oa.data());
              const int rb = emul_downsample(codec, strategy, gh, gw, a.data(), ob.data());
              CHECK(ra == rb && (ra == 0 || oa == ob), "downsample mismatch codec %d %ux%u (%d %d)", codec, gh, gw, ra, rb);
            }
            {  // sub-image and solid image through the oracle (host byte shuffling; the bounds are what is exercised)
              const uint32_t sr = 4 * (rnd() % (ch / 4 + 1)), sc = 4 * (rnd() % (cw / 4 + 1));
              const uint32_t sh = 4 * (rnd() % ((ch - sr) / 4 + 1)), sw = 4 * (rnd() % ((cw - sc) / 4 + 1));
              std::vector<uint8_t> sub((size_t)(sh / 4) * (sw / 4) * block_bytes(codec) + 1);
              CHECK(ico_copy_subimage(compressor, format, ch, cw, a.data(), sr, sc, sh, sw, sub.data()) == 1, "copy_subimage");
              std::vector<uint8_t> solid(ico_encoded_size(codec, h, w));
              const uint8_t color[4] = { (uint8_t)rnd(), (uint8_t)rnd(), (uint8_t)rnd(), (uint8_t)rnd() };
              CHECK(ico_create_solid(compressor, format, h, w, color, solid.data()) == 1, "create_solid");
            }
            if (codec == ICO_DXT1) {  // in-place DXT1 -> ETC1
              std::vector<uint8_t> ta(a), tb(a);
              ico_transcode_dxt1_to_etc1(ta.data(), ta.size());
              emul_transcode(tb.data(), tb.size());
              CHECK(ta == tb, "transcode mismatch %ux%u", gh, gw);
            }
          }
        }
      }
  // ---- the public wrappers with their validation (wrong sizes, formats, null-free paths)
  {
    const std::vector<uint8_t> img = make_image(12, 20, 4, 20 * 4 + 3, 0);
    for (int compressor = 0; compressor < 3; ++compressor)
      for (int format = 0; format < 4; ++format) {
        const size_t n = ico_compute_compressed_data_size(compressor, format, 12, 20);
        std::vector<uint8_t> out(n + 1);
        (void)ico_compress(compressor, 2, format, 12, 20, 3, img.data(), out.data(), n);
        (void)ico_compress(compressor, 2, format, 12, 20, 3, img.data(), out.data(), n + 1);
        const size_t np = ico_compute_compressed_data_size(compressor, format, 16, 28);
        std::vector<uint8_t> outp(np + 1);
        (void)ico_compress_and_pad(compressor, 2, format, 12, 20, 16, 28, 3, img.data(), outp.data(), np);
      }
  }
  // ---- PVRTC 2bpp: encode (oracle vs the device math incl. its internal cross-checks) and the decoder extension
  for (uint32_t n = 8; n <= 64; n *= 2)
    for (int kind = 0; kind < 5; ++kind) {
      const std::vector<uint8_t> img = make_image(n, n, 4, n * 4, kind);
      const size_t bytes = (size_t)n * n / 4;
      std::vector<uint8_t> a(bytes), b(bytes);
      CHECK(ico_encode(ICO_PVRTC2, 0, 4, 0, n, n, n, n, n * 4, img.data(), a.data(), 1) == 1, "ico_encode pvrtc");
      CHECK(emul_encode(ICO_PVRTC2, 0, 4, 0, n, n, n, n, n * 4, img.data(), b.data()) == 1, "emul_encode pvrtc %u kind %d", n, kind);
      CHECK(a == b, "pvrtc mismatch %u kind %d", n, kind);
      std::vector<uint8_t> pa((size_t)n * n * 4), pb(pa.size());
      CHECK(ico_decode(ICO_PVRTC2, 0, n, n, 0, a.data(), pa.data()) == 1, "ico_decode pvrtc");
      CHECK(emul_decode(ICO_PVRTC2, 0, n, n, 0, a.data(), pb.data()) == 1, "emul_decode pvrtc");
      CHECK(pa == pb, "pvrtc decode mismatch %u", n);
    }
  if (failures) {
    fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  printf("sanitizer driver: all checks passed\n");
  return 0;
}
