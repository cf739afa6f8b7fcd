// TEST INFRASTRUCTURE ONLY -- not part of the product.
//
// C-ABI shim around the *unmodified* reference library (google/image-compression,
// compiled from /root/reference where it lies; see oracle/Makefile target `_ref`).
// It lets pytest (ctypes) call the reference's own public Compressor API so that
//   (1) oracle/ic_oracle.c (our CPU restatement) is pinned byte-for-byte, and
//   (2) tests/golden/make_golden.py can emit golden vectors.
// Nothing here is algorithm code: every function forwards to a reference
// entry point declared in image_compression/public/*.h.
#include <stddef.h>
#include <string.h>
#include <vector>

#include "base/integral_types.h"
#include "image_compression/public/compressed_image.h"
#include "image_compression/public/compressor.h"
#include "image_compression/public/dxtc_compressor.h"
#include "image_compression/public/dxtc_to_etc_transcoder.h"
#include "image_compression/public/etc_compressor.h"
#include "image_compression/public/pvrtc_compressor.h"

using namespace image_codec_compression;

namespace {

// codec ids shared with tests/ref_lib.py: 0 = dxtc, 1 = etc, 2 = pvrtc.
struct Holder {
  DxtcCompressor dxtc;
  EtcCompressor etc;
  PvrtcCompressor pvrtc;
  Compressor* get(int codec, int etc_strategy) {
    if (codec == 0) return &dxtc;
    if (codec == 1) {
      etc.SetCompressionStrategy(
          static_cast<EtcCompressor::CompressionStrategy>(etc_strategy));
      return &etc;
    }
    if (codec == 2) return &pvrtc;
    return NULL;
  }
};

const char* CodecName(int codec) {
  return codec == 0 ? "dxtc" : codec == 1 ? "etc" : "pvrtc";
}

}  // namespace

extern "C" {

size_t ref_compressed_size(int codec, int format, uint32 h, uint32 w) {
  Holder hd;
  Compressor* c = hd.get(codec, 2);
  return c ? c->ComputeCompressedDataSize(
                 static_cast<CompressedImage::Format>(format), h, w)
           : 0;
}

int ref_supports_format(int codec, int format) {
  Holder hd;
  Compressor* c = hd.get(codec, 2);
  return c && c->SupportsFormat(static_cast<CompressedImage::Format>(format));
}

// Compress into caller storage of exactly out_size bytes (external-storage
// CompressedImage, compressed_image.h:94-100).  Returns the reference's bool.
// meta[5] (optional) receives uncompressed h,w, compressed h,w, padding.
int ref_compress(int codec, int etc_strategy, int format, uint32 h, uint32 w,
                 uint32 pad, const uint8* buf, uint8* out, size_t out_size,
                 uint32* meta) {
  Holder hd;
  Compressor* c = hd.get(codec, etc_strategy);
  if (!c) return 0;
  CompressedImage img(out_size, out);
  bool ok = c->Compress(static_cast<CompressedImage::Format>(format), h, w,
                        pad, buf, &img);
  if (ok && meta) {
    const CompressedImage::Metadata& m = img.GetMetadata();
    meta[0] = m.uncompressed_height; meta[1] = m.uncompressed_width;
    meta[2] = m.compressed_height;   meta[3] = m.compressed_width;
    meta[4] = m.padding_bytes_per_row;
  }
  return ok;
}

// Same, but lets the reference allocate (owned data) and reports the size it
// chose; copies up to out_cap bytes back.  Returns -1 on false.
long ref_compress_owned(int codec, int etc_strategy, int format, uint32 h,
                        uint32 w, uint32 pad, const uint8* buf, uint8* out,
                        size_t out_cap) {
  Holder hd;
  Compressor* c = hd.get(codec, etc_strategy);
  if (!c) return -1;
  CompressedImage img;
  if (!c->Compress(static_cast<CompressedImage::Format>(format), h, w, pad,
                   buf, &img))
    return -1;
  size_t n = img.GetDataSize();
  if (out && n <= out_cap) memcpy(out, img.GetData(), n);
  return static_cast<long>(n);
}

int ref_compress_and_pad(int codec, int etc_strategy, int format, uint32 h,
                         uint32 w, uint32 padded_h, uint32 padded_w,
                         uint32 pad, const uint8* buf, uint8* out,
                         size_t out_size, uint32* meta) {
  Holder hd;
  Compressor* c = hd.get(codec, etc_strategy);
  if (!c) return 0;
  CompressedImage img(out_size, out);
  bool ok = c->CompressAndPad(static_cast<CompressedImage::Format>(format), h,
                              w, padded_h, padded_w, pad, buf, &img);
  if (ok && meta) {
    const CompressedImage::Metadata& m = img.GetMetadata();
    meta[0] = m.uncompressed_height; meta[1] = m.uncompressed_width;
    meta[2] = m.compressed_height;   meta[3] = m.compressed_width;
    meta[4] = m.padding_bytes_per_row;
  }
  return ok;
}

// Decompress blocks that were produced for an image with the given metadata.
// Returns number of bytes written (the reference resizes its vector), -1 on false.
long ref_decompress(int codec, int format, uint32 uh, uint32 uw, uint32 ch,
                    uint32 cw, uint32 pad, const uint8* blocks,
                    size_t blocks_size, uint8* out, size_t out_cap) {
  Holder hd;
  Compressor* c = hd.get(codec, 2);
  if (!c) return -1;
  CompressedImage img(blocks_size, const_cast<uint8*>(blocks));
  img.SetMetadata(CompressedImage::Metadata(
      static_cast<CompressedImage::Format>(format), CodecName(codec), uh, uw,
      ch, cw, pad));
  std::vector<uint8> v;
  if (!c->Decompress(img, &v)) return -1;
  if (out && v.size() <= out_cap && !v.empty()) memcpy(out, &v[0], v.size());
  return static_cast<long>(v.size());
}

int ref_create_solid(int codec, int format, uint32 h, uint32 w,
                     const uint8* color, uint8* out, size_t out_size) {
  Holder hd;
  Compressor* c = hd.get(codec, 2);
  if (!c) return 0;
  CompressedImage img(out_size, out);
  return c->CreateSolidImage(static_cast<CompressedImage::Format>(format), h,
                             w, color, &img);
}

// Pad / Downsample / CopySubimage on a block buffer; result size via *out_n.
int ref_pad(int codec, int etc_strategy, int format, uint32 uh, uint32 uw,
            uint32 ch, uint32 cw, const uint8* blocks, size_t blocks_size,
            uint32 padded_h, uint32 padded_w, uint8* out, size_t out_cap,
            size_t* out_n, uint32* meta) {
  Holder hd;
  Compressor* c = hd.get(codec, etc_strategy);
  if (!c) return 0;
  CompressedImage img(blocks_size, const_cast<uint8*>(blocks));
  img.SetMetadata(CompressedImage::Metadata(
      static_cast<CompressedImage::Format>(format), CodecName(codec), uh, uw,
      ch, cw, 0));
  CompressedImage res;
  if (!c->Pad(img, padded_h, padded_w, &res)) return 0;
  *out_n = res.GetDataSize();
  if (out && *out_n <= out_cap) memcpy(out, res.GetData(), *out_n);
  if (meta) {
    const CompressedImage::Metadata& m = res.GetMetadata();
    meta[0] = m.uncompressed_height; meta[1] = m.uncompressed_width;
    meta[2] = m.compressed_height;   meta[3] = m.compressed_width;
    meta[4] = m.padding_bytes_per_row;
  }
  return 1;
}

int ref_downsample(int codec, int etc_strategy, int format, uint32 uh,
                   uint32 uw, uint32 ch, uint32 cw, const uint8* blocks,
                   size_t blocks_size, uint8* out, size_t out_cap,
                   size_t* out_n, uint32* meta) {
  Holder hd;
  Compressor* c = hd.get(codec, etc_strategy);
  if (!c) return 0;
  CompressedImage img(blocks_size, const_cast<uint8*>(blocks));
  img.SetMetadata(CompressedImage::Metadata(
      static_cast<CompressedImage::Format>(format), CodecName(codec), uh, uw,
      ch, cw, 0));
  CompressedImage res;
  if (!c->Downsample(img, &res)) return 0;
  *out_n = res.GetDataSize();
  if (out && *out_n <= out_cap) memcpy(out, res.GetData(), *out_n);
  if (meta) {
    const CompressedImage::Metadata& m = res.GetMetadata();
    meta[0] = m.uncompressed_height; meta[1] = m.uncompressed_width;
    meta[2] = m.compressed_height;   meta[3] = m.compressed_width;
    meta[4] = m.padding_bytes_per_row;
  }
  return 1;
}

int ref_copy_subimage(int codec, int format, uint32 uh, uint32 uw, uint32 ch,
                      uint32 cw, const uint8* blocks, size_t blocks_size,
                      uint32 row, uint32 col, uint32 h, uint32 w, uint8* out,
                      size_t out_cap, size_t* out_n, uint32* meta) {
  Holder hd;
  Compressor* c = hd.get(codec, 2);
  if (!c) return 0;
  CompressedImage img(blocks_size, const_cast<uint8*>(blocks));
  img.SetMetadata(CompressedImage::Metadata(
      static_cast<CompressedImage::Format>(format), CodecName(codec), uh, uw,
      ch, cw, 0));
  CompressedImage res;
  if (!c->CopySubimage(img, row, col, h, w, &res)) return 0;
  *out_n = res.GetDataSize();
  if (out && *out_n <= out_cap) memcpy(out, res.GetData(), *out_n);
  if (meta) {
    const CompressedImage::Metadata& m = res.GetMetadata();
    meta[0] = m.uncompressed_height; meta[1] = m.uncompressed_width;
    meta[2] = m.compressed_height;   meta[3] = m.compressed_width;
    meta[4] = m.padding_bytes_per_row;
  }
  return 1;
}

// In-place DXT1 -> ETC1 (dxtc_to_etc_transcoder.cc:29-40).
void ref_transcode_dxt1_to_etc1(uint32 uh, uint32 uw, uint32 ch, uint32 cw,
                                uint8* blocks, size_t blocks_size) {
  CompressedImage img(blocks_size, blocks);
  img.SetMetadata(CompressedImage::Metadata(CompressedImage::kRGB, "dxtc", uh,
                                            uw, ch, cw, 0));
  TranscodeDxt1ToEtc1(&img);
}

}  // extern "C"
