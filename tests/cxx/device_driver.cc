// TEST PROGRAM for the device-resident EXTENSION of the C++ classes (compressor.h: CompressDevice / CompressAndPadDevice /
// CompressBatchDevice -- not part of the reference's interface, so this driver only builds against this repo's classes).
// For every class, format and shape: the bytes the device forms leave in HBM must equal the bytes of the host-buffer drop-in
// (Compress / CompressAndPad), whose transcript tests/cxx/api_driver.cc pins against the reference.  Prints one line per
// case and "device extension: N cases, all equal" at the end; exit code 1 on any difference.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "image_compression/public/compressed_image.h"
#include "image_compression/public/dxtc_compressor.h"
#include "image_compression/public/etc_compressor.h"
#include "image_compression/public/pvrtc_compressor.h"

using namespace image_codec_compression;

static int g_cases = 0, g_bad = 0;

static std::vector<uint8> MakeImage(uint32 h, uint32 w, uint32 comps, uint32 pad, uint32 seed) {
  std::vector<uint8> v((size_t)h * (w * comps + pad));
  uint32 x = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < v.size(); ++i) {
    x = x * 1664525u + 1013904223u;
    const uint32 yy = (uint32)(i / (w * comps + pad)), xx = (uint32)(i % (w * comps + pad)) / comps;
    v[i] = (uint8)(((x >> 24) & 63u) + 3u * (xx / 4) + 2u * (yy / 4));  // coarse gradient + noise: varied blocks
  }
  return v;
}

struct DeviceBuf {
  void *p;
  explicit DeviceBuf(size_t n) : p(nullptr) { if (hipMalloc(&p, n ? n : 1) != hipSuccess) p = nullptr; }
  ~DeviceBuf() { if (p) (void)hipFree(p); }
};

static void Check(const char *what, bool ok_host, bool ok_dev, const uint8 *host, const std::vector<uint8> &dev, size_t n) {
  ++g_cases;
  const bool same = ok_host == ok_dev && (!ok_host || std::memcmp(host, dev.data(), n) == 0);
  if (!same) ++g_bad;
  std::printf("%s: host %s, device %s -> %s\n", what, ok_host ? "true" : "false", ok_dev ? "true" : "false", same ? "equal" : "DIFFERENT");
}

template <typename C>
static void Run(C *c, const char *name, CompressedImage::Format format, uint32 comps, uint32 h, uint32 w, uint32 pad, uint32 ph, uint32 pw,
                hipStream_t stream) {
  char what[160];
  const std::vector<uint8> img = MakeImage(h, w, comps, pad, h * 131u + w);
  DeviceBuf d_in(img.size());
  (void)hipMemcpy(d_in.p, img.data(), img.size(), hipMemcpyHostToDevice);
  {  // Compress
    CompressedImage host;
    const bool ok_host = c->Compress(format, h, w, pad, img.data(), &host);
    const size_t n = c->ComputeCompressedDataSize(format, h, w);
    DeviceBuf d_out(n);
    const bool ok_dev = c->CompressDevice(format, h, w, pad, d_in.p, d_out.p, n, stream);
    (void)hipStreamSynchronize(stream);
    std::vector<uint8> got(n ? n : 1);
    (void)hipMemcpy(got.data(), d_out.p, n, hipMemcpyDeviceToHost);
    std::snprintf(what, sizeof what, "%s CompressDevice fmt=%d %ux%u pad=%u", name, (int)format, h, w, pad);
    Check(what, ok_host, ok_dev, ok_host ? host.GetData() : nullptr, got, ok_host ? host.GetDataSize() : 0);
    if (ok_host) {  // a wrong out_size is refused like external storage of the wrong size
      ++g_cases;
      const bool refused = !c->CompressDevice(format, h, w, pad, d_in.p, d_out.p, n + 8, stream);
      if (!refused) ++g_bad;
      std::printf("%s CompressDevice wrong out_size -> %s\n", name, refused ? "false" : "TRUE (should be refused)");
    }
  }
  if (ph || pw) {  // CompressAndPad
    CompressedImage host;
    const bool ok_host = c->CompressAndPad(format, h, w, ph, pw, pad, img.data(), &host);
    const size_t n = c->ComputeCompressedDataSize(format, h > ph ? h : ph, w > pw ? w : pw);
    DeviceBuf d_out(n);
    const bool ok_dev = c->CompressAndPadDevice(format, h, w, ph, pw, pad, d_in.p, d_out.p, n, stream);
    (void)hipStreamSynchronize(stream);
    std::vector<uint8> got(n ? n : 1);
    (void)hipMemcpy(got.data(), d_out.p, n, hipMemcpyDeviceToHost);
    std::snprintf(what, sizeof what, "%s CompressAndPadDevice fmt=%d %ux%u -> %ux%u pad=%u", name, (int)format, h, w, ph, pw, pad);
    Check(what, ok_host, ok_dev, ok_host ? host.GetData() : nullptr, got, ok_host ? host.GetDataSize() : 0);
  }
}

template <typename C>
static void RunBatch(C *c, const char *name, CompressedImage::Format format, uint32 comps, uint32 h, uint32 w, uint32 n_images,
                     hipStream_t stream) {
  const size_t per_in = (size_t)h * w * comps, per_out = c->ComputeCompressedDataSize(format, h, w);
  const size_t in_stride = per_in + 64, out_stride = per_out + 32;  // padded strides on both sides
  std::vector<uint8> all(in_stride * n_images);
  std::vector<std::vector<uint8> > want;
  bool ok_host = true;
  for (uint32 i = 0; i < n_images; ++i) {
    const std::vector<uint8> img = MakeImage(h, w, comps, 0, 7000u + i);
    std::memcpy(&all[i * in_stride], img.data(), per_in);
    CompressedImage host;
    ok_host = c->Compress(format, h, w, 0, img.data(), &host) && ok_host;
    want.push_back(ok_host ? std::vector<uint8>(host.GetData(), host.GetData() + host.GetDataSize()) : std::vector<uint8>());
  }
  DeviceBuf d_in(all.size()), d_out(out_stride * n_images);
  (void)hipMemcpy(d_in.p, all.data(), all.size(), hipMemcpyHostToDevice);
  const bool ok_dev = c->CompressBatchDevice(format, h, w, 0, n_images, d_in.p, in_stride, d_out.p, out_stride, per_out, stream);
  (void)hipStreamSynchronize(stream);
  std::vector<uint8> got(out_stride * n_images);
  (void)hipMemcpy(got.data(), d_out.p, got.size(), hipMemcpyDeviceToHost);
  ++g_cases;
  bool same = ok_host == ok_dev;
  for (uint32 i = 0; same && ok_host && i < n_images; ++i) same = std::memcmp(&got[i * out_stride], want[i].data(), per_out) == 0;
  if (!same) ++g_bad;
  std::printf("%s CompressBatchDevice fmt=%d %u x %ux%u: host %s, device %s -> %s\n", name, (int)format, n_images, h, w,
              ok_host ? "true" : "false", ok_dev ? "true" : "false", same ? "equal" : "DIFFERENT");
}

int main() {
  hipStream_t stream;
  if (hipStreamCreate(&stream) != hipSuccess) { std::printf("no HIP device\n"); return 2; }
  DxtcCompressor dxtc;
  EtcCompressor etc;
  PvrtcCompressor pvrtc;
  const CompressedImage::Format f3[2] = { CompressedImage::kRGB, CompressedImage::kBGR };
  const CompressedImage::Format f4[2] = { CompressedImage::kRGBA, CompressedImage::kBGRA };
  const uint32 shapes[4][5] = { { 64, 64, 0, 0, 0 }, { 61, 59, 3, 64, 72 }, { 5, 3, 0, 8, 8 }, { 256, 512, 0, 0, 0 } };
  for (int s = 0; s < 4; ++s) {
    for (int f = 0; f < 2; ++f) {
      Run(&dxtc, "dxtc", f3[f], 3, shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], shapes[s][4], stream);
      Run(&dxtc, "dxtc", f4[f], 4, shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], shapes[s][4], stream);
    }
    for (int st = 0; st < 4; ++st) {
      etc.SetCompressionStrategy((EtcCompressor::CompressionStrategy)st);
      char nm[16];
      std::snprintf(nm, sizeof nm, "etc[s%d]", st);
      Run(&etc, nm, CompressedImage::kRGB, 3, shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], shapes[s][4], stream);
    }
  }
  etc.SetCompressionStrategy(EtcCompressor::kSmallerError);
  Run(&etc, "etc", CompressedImage::kRGBA, 4, 16, 16, 0, 0, 0, stream);  // refused by both (etc_compressor.cc:751-754)
  const uint32 psizes[3] = { 8, 64, 512 };
  for (int s = 0; s < 3; ++s) Run(&pvrtc, "pvrtc", CompressedImage::kRGBA, 4, psizes[s], psizes[s], 0, 0, 0, stream);
  Run(&pvrtc, "pvrtc", CompressedImage::kRGBA, 4, 16, 32, 0, 0, 0, stream);  // not square: refused by both
  Run(&pvrtc, "pvrtc", CompressedImage::kRGBA, 4, 16, 16, 0, 32, 32, stream);  // CompressAndPad: refused by both
  RunBatch(&dxtc, "dxtc", CompressedImage::kRGB, 3, 128, 96, 5, stream);
  RunBatch(&dxtc, "dxtc", CompressedImage::kBGRA, 4, 64, 64, 3, stream);
  RunBatch(&etc, "etc", CompressedImage::kRGB, 3, 128, 128, 4, stream);
  RunBatch(&pvrtc, "pvrtc", CompressedImage::kRGBA, 4, 512, 512, 9, stream);
  RunBatch(&etc, "etc", CompressedImage::kBGR, 3, 16, 16, 2, stream);  // refused by both
  (void)hipStreamDestroy(stream);
  std::printf("device extension: %d cases, %s\n", g_cases, g_bad ? "DIFFERENCES FOUND" : "all equal");
  return g_bad ? 1 : 0;
}
