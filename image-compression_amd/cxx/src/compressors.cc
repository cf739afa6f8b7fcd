// compressors.cc -- the reference's concrete Compressor classes, re-implemented as thin C++ hosts
// over the MI355X C ABI (include/ic_amd.h).  Argument handling and CompressedImage set-up follow the
// reference (internal/compressor4x4_helper.cc:22-43, dxtc_compressor.cc:700-854,
// etc_compressor.cc:700-826, pvrtc_compressor.cc:599-705); all encoding happens in HIP kernels.
// A device failure (no GPU, HIP error) is reported on stderr and returned as `false`: the API has no
// other error channel, and there is deliberately no CPU fallback.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ic_amd.h"
#include "image_compression/public/compressed_image.h"
#include "image_compression/public/compressor.h"
#include "image_compression/public/dxtc_compressor.h"
#include "image_compression/public/dxtc_to_etc_transcoder.h"
#include "image_compression/public/etc_compressor.h"
#include "image_compression/public/pvrtc_compressor.h"

namespace image_codec_compression {

namespace {

uint32 BlocksFor(uint32 pixels) { return (pixels + 3) / 4; }

bool ReportStatus(int status, const char *where) {
  if (status == ICAMD_OK) return true;
  if (status < 0) std::fprintf(stderr, "image-compression_amd: %s failed (%d): %s\n", where, status, icamd_last_error());
  return false;
}

// SetUpCompressedImage (compressor4x4_helper.cc:22-43) for the 4x4 codecs, generalised to PVRTC's
// metadata (pvrtc_compressor.cc:653-662): allocate owned storage, or validate external storage.
bool PrepareImage(const CompressedImage::Metadata &metadata, size_t data_size, CompressedImage *image) {
  if (image->OwnsData()) {
    image->CreateOwnedData(metadata, data_size);
    return true;
  }
  if (image->GetDataSize() != data_size) return false;
  image->SetMetadata(metadata);
  return true;
}

// Shared body of Compress / CompressAndPad for DXTC and ETC.
bool Compress4x4(int compressor, int etc_strategy, const char *name, CompressedImage::Format format, uint32 height,
                 uint32 width, uint32 padded_height, uint32 padded_width, uint32 padding_bytes_per_row,
                 const uint8 *buffer, CompressedImage *image) {
  const uint32 final_height = std::max(height, padded_height), final_width = std::max(width, padded_width);
  const size_t data_size = icamd_compute_compressed_data_size(compressor, format, final_height, final_width);
  const CompressedImage::Metadata metadata(format, name, final_height, final_width, 4 * BlocksFor(final_height),
                                           4 * BlocksFor(final_width), padding_bytes_per_row);
  if (!PrepareImage(metadata, data_size, image)) return false;
  return ReportStatus(icamd_compress_and_pad(compressor, etc_strategy, format, height, width, padded_height,
                                             padded_width, padding_bytes_per_row, buffer, image->GetMutableData(),
                                             data_size),
                      "icamd_compress_and_pad");
}

bool Valid4x4(const CompressedImage &image, int compressor, const char *name, bool rgb_only) {
  const CompressedImage::Metadata &m = image.GetMetadata();
  if (rgb_only && m.format != CompressedImage::kRGB) return false;
  return m.compressor_name == name && m.uncompressed_height > 0 && m.uncompressed_width > 0 &&
         m.compressed_height >= m.uncompressed_height && m.compressed_width >= m.uncompressed_width &&
         image.GetDataSize() == icamd_compute_compressed_data_size(compressor, m.format, m.compressed_height,
                                                                   m.compressed_width);
}

bool Decompress4x4(const CompressedImage &image, int compressor, std::vector<uint8> *out) {
  const CompressedImage::Metadata &m = image.GetMetadata();
  const size_t comps = GetNumFormatComponents(m.format);
  // The reference sizes the vector without the row padding but addresses rows with it
  // (compressor4x4_helper.h:225-238); we allocate what is actually addressed.
  out->assign((size_t)m.uncompressed_height * ((size_t)m.uncompressed_width * comps + m.padding_bytes_per_row), 0);
  const bool ok = ReportStatus(
      icamd_decompress(compressor, m.format, m.uncompressed_height, m.uncompressed_width, m.padding_bytes_per_row,
                       image.GetData(), image.GetDataSize(), out->data(), out->size()),
      "icamd_decompress");
  if (ok && m.padding_bytes_per_row != 0) out->resize((size_t)m.uncompressed_height * m.uncompressed_width * comps);
  return ok;
}

// Compressor4x4Helper::Pad (compressor4x4_helper.h:393-477).
bool Pad4x4(const CompressedImage &image, int compressor, int etc_strategy, uint32 padded_height, uint32 padded_width,
            CompressedImage *padded_image) {
  const CompressedImage::Metadata &m = image.GetMetadata();
  if (m.compressed_height >= padded_height && m.compressed_width >= padded_width) {
    padded_image->Duplicate(image);  // nothing to pad: plain copy (helper.h:404-408)
    return true;
  }
  // The reference overruns its buffers when exactly one padded dimension has fewer blocks than the image; refuse.
  if (BlocksFor(padded_height) < BlocksFor(m.compressed_height) || BlocksFor(padded_width) < BlocksFor(m.compressed_width))
    return false;
  const size_t data_size = icamd_compute_compressed_data_size(compressor, m.format, padded_height, padded_width);
  const CompressedImage::Metadata metadata(m.format, m.compressor_name, padded_height, padded_width,
                                           4 * BlocksFor(padded_height), 4 * BlocksFor(padded_width), 0);
  if (!PrepareImage(metadata, data_size, padded_image)) return false;
  return ReportStatus(icamd_pad(compressor, etc_strategy, m.format, m.compressed_height, m.compressed_width,
                                image.GetData(), padded_height, padded_width, padded_image->GetMutableData(), data_size),
                      "icamd_pad");
}

// Compressor4x4Helper::Downsample (compressor4x4_helper.h:264-391).
bool Downsample4x4(const CompressedImage &image, int compressor, int etc_strategy, CompressedImage *downsampled) {
  const CompressedImage::Metadata &m = image.GetMetadata();
  const uint32 rows = BlocksFor(m.uncompressed_height), cols = BlocksFor(m.uncompressed_width);
  if ((rows > 1 && rows % 2 != 0) || (cols > 1 && cols % 2 != 0)) return false;
  const uint32 dh = (m.uncompressed_height + 1) / 2, dw = (m.uncompressed_width + 1) / 2;
  const size_t data_size = icamd_compute_compressed_data_size(compressor, m.format, dh, dw);
  const CompressedImage::Metadata metadata(m.format, m.compressor_name, dh, dw, 4 * BlocksFor(dh), 4 * BlocksFor(dw), 0);
  if (!PrepareImage(metadata, data_size, downsampled)) return false;
  // (like the reference, the single-block case can still fail after the image was set up: 3-pixel sides)
  return ReportStatus(icamd_downsample(compressor, etc_strategy, m.format, m.uncompressed_height, m.uncompressed_width,
                                       image.GetData(), downsampled->GetMutableData(), data_size),
                      "icamd_downsample");
}

// Compressor4x4Helper::CreateSolidImage (compressor4x4_helper.h:522-543): image set-up here, the block and its
// replication behind the C ABI (icamd_create_solid; device-resident grids: icamd_create_solid_device).
bool Solid4x4(int compressor, const char *name, CompressedImage::Format format, uint32 height, uint32 width,
              const uint8 *color, CompressedImage *image) {
  const size_t block_size = (compressor == ICAMD_COMPRESSOR_ETC || GetNumFormatComponents(format) == 3) ? 8 : 16;
  const size_t data_size = (size_t)BlocksFor(height) * BlocksFor(width) * block_size;
  const CompressedImage::Metadata metadata(format, name, height, width, 4 * BlocksFor(height), 4 * BlocksFor(width), 0);
  if (!PrepareImage(metadata, data_size, image)) return false;
  return ReportStatus(icamd_create_solid(compressor, format, height, width, color, image->GetMutableData(), data_size),
                      "icamd_create_solid");
}

// Compressor4x4Helper::CopySubimage (compressor4x4_helper.h:545-592): validation + set-up here (the reference refuses
// BEFORE touching the output image), block-row copies behind the C ABI (icamd_copy_subimage / _device).
bool Subimage4x4(const CompressedImage &image, int compressor, size_t block_size, uint32 start_row, uint32 start_column,
                 uint32 height, uint32 width, CompressedImage *subimage) {
  const CompressedImage::Metadata &m = image.GetMetadata();
  if (start_row % 4 != 0 || start_column % 4 != 0 || height % 4 != 0 || width % 4 != 0 ||
      start_row > m.compressed_height || start_column > m.compressed_width ||
      start_row + height > m.compressed_height || start_column + width > m.compressed_width)
    return false;
  const uint32 sub_rows = BlocksFor(height), sub_cols = BlocksFor(width);
  const size_t data_size = (size_t)sub_rows * sub_cols * block_size;
  const CompressedImage::Metadata metadata(m.format, m.compressor_name, height, width, 4 * sub_rows, 4 * sub_cols, 0);
  if (!PrepareImage(metadata, data_size, subimage)) return false;
  return ReportStatus(icamd_copy_subimage(compressor, m.format, m.compressed_height, m.compressed_width, image.GetData(),
                                          start_row, start_column, height, width, subimage->GetMutableData(), data_size),
                      "icamd_copy_subimage");
}

}  // namespace

// ------------------------------------------------------------------ DXTC

DxtcCompressor::DxtcCompressor() {}
DxtcCompressor::~DxtcCompressor() {}

bool DxtcCompressor::SupportsFormat(CompressedImage::Format format) const {
  return icamd_supports_format(ICAMD_COMPRESSOR_DXTC, format) != 0;
}

bool DxtcCompressor::IsValidCompressedImage(const CompressedImage &image) {
  return Valid4x4(image, ICAMD_COMPRESSOR_DXTC, "dxtc", false);
}

size_t DxtcCompressor::ComputeCompressedDataSize(CompressedImage::Format format, uint32 height, uint32 width) {
  return icamd_compute_compressed_data_size(ICAMD_COMPRESSOR_DXTC, format, height, width);
}

bool DxtcCompressor::Compress(CompressedImage::Format format, uint32 height, uint32 width,
                              uint32 padding_bytes_per_row, const uint8 *buffer, CompressedImage *image) {
  if (!buffer || !image || height == 0 || width == 0) return false;
  return Compress4x4(ICAMD_COMPRESSOR_DXTC, 0, "dxtc", format, height, width, height, width, padding_bytes_per_row,
                     buffer, image);
}

bool DxtcCompressor::CompressAndPad(CompressedImage::Format format, uint32 height, uint32 width, uint32 padded_height,
                                    uint32 padded_width, uint32 padding_bytes_per_row, const uint8 *buffer,
                                    CompressedImage *padded_image) {
  if (!buffer || !padded_image || height == 0 || width == 0) return false;
  return Compress4x4(ICAMD_COMPRESSOR_DXTC, 0, "dxtc", format, height, width, padded_height, padded_width,
                     padding_bytes_per_row, buffer, padded_image);
}

bool DxtcCompressor::Decompress(const CompressedImage &image, std::vector<uint8> *decompressed_buffer) {
  if (!IsValidCompressedImage(image) || !decompressed_buffer) return false;
  return Decompress4x4(image, ICAMD_COMPRESSOR_DXTC, decompressed_buffer);
}

bool DxtcCompressor::Downsample(const CompressedImage &image, CompressedImage *downsampled_image) {
  if (!IsValidCompressedImage(image) || !downsampled_image) return false;
  return Downsample4x4(image, ICAMD_COMPRESSOR_DXTC, 0, downsampled_image);
}

bool DxtcCompressor::Pad(const CompressedImage &image, uint32 padded_height, uint32 padded_width,
                         CompressedImage *padded_image) {
  if (!IsValidCompressedImage(image) || !padded_image) return false;
  return Pad4x4(image, ICAMD_COMPRESSOR_DXTC, 0, padded_height, padded_width, padded_image);
}

bool DxtcCompressor::CreateSolidImage(CompressedImage::Format format, uint32 height, uint32 width, const uint8 *color,
                                      CompressedImage *image) {
  if (!image) return false;
  // dxtc_compressor.cc:820-839; the solid block itself: icamd_create_solid (ic_capi.hip, solid_block)
  return Solid4x4(ICAMD_COMPRESSOR_DXTC, "dxtc", format, height, width, color, image);
}

bool DxtcCompressor::CopySubimage(const CompressedImage &image, uint32 start_row, uint32 start_column, uint32 height,
                                  uint32 width, CompressedImage *subimage) {
  if (!IsValidCompressedImage(image) || !subimage) return false;
  return Subimage4x4(image, ICAMD_COMPRESSOR_DXTC, GetNumFormatComponents(image.GetMetadata().format) == 3 ? 8 : 16,
                     start_row, start_column, height, width, subimage);
}

// ------------------------------------------------------------------- ETC

EtcCompressor::EtcCompressor() : compression_strategy_(kSmallerError) {}
EtcCompressor::~EtcCompressor() {}

bool EtcCompressor::SupportsFormat(CompressedImage::Format format) const {
  return icamd_supports_format(ICAMD_COMPRESSOR_ETC, format) != 0;
}

bool EtcCompressor::IsValidCompressedImage(const CompressedImage &image) {
  return Valid4x4(image, ICAMD_COMPRESSOR_ETC, "etc", true);
}

size_t EtcCompressor::ComputeCompressedDataSize(CompressedImage::Format format, uint32 height, uint32 width) {
  return icamd_compute_compressed_data_size(ICAMD_COMPRESSOR_ETC, format, height, width);
}

bool EtcCompressor::Compress(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,
                             const uint8 *buffer, CompressedImage *image) {
  if (!buffer || !image || height == 0 || width == 0 || format != CompressedImage::kRGB) return false;
  return Compress4x4(ICAMD_COMPRESSOR_ETC, compression_strategy_, "etc", format, height, width, height, width,
                     padding_bytes_per_row, buffer, image);
}

bool EtcCompressor::CompressAndPad(CompressedImage::Format format, uint32 height, uint32 width, uint32 padded_height,
                                   uint32 padded_width, uint32 padding_bytes_per_row, const uint8 *buffer,
                                   CompressedImage *padded_image) {
  if (!buffer || !padded_image || height == 0 || width == 0 || format != CompressedImage::kRGB) return false;
  return Compress4x4(ICAMD_COMPRESSOR_ETC, compression_strategy_, "etc", format, height, width, padded_height,
                     padded_width, padding_bytes_per_row, buffer, padded_image);
}

bool EtcCompressor::Decompress(const CompressedImage &image, std::vector<uint8> *decompressed_buffer) {
  if (!IsValidCompressedImage(image) || !decompressed_buffer) return false;
  return Decompress4x4(image, ICAMD_COMPRESSOR_ETC, decompressed_buffer);
}

bool EtcCompressor::Downsample(const CompressedImage &image, CompressedImage *downsampled_image) {
  if (!IsValidCompressedImage(image) || !downsampled_image) return false;
  return Downsample4x4(image, ICAMD_COMPRESSOR_ETC, compression_strategy_, downsampled_image);
}

bool EtcCompressor::Pad(const CompressedImage &image, uint32 padded_height, uint32 padded_width,
                        CompressedImage *padded_image) {
  if (!IsValidCompressedImage(image) || !padded_image) return false;
  return Pad4x4(image, ICAMD_COMPRESSOR_ETC, compression_strategy_, padded_height, padded_width, padded_image);
}

bool EtcCompressor::CreateSolidImage(CompressedImage::Format format, uint32 height, uint32 width, const uint8 *color,
                                     CompressedImage *image) {
  if (!image || format != CompressedImage::kRGB) return false;  // etc_compressor.cc:802-812
  return Solid4x4(ICAMD_COMPRESSOR_ETC, "etc", format, height, width, color, image);
}

bool EtcCompressor::CopySubimage(const CompressedImage &image, uint32 start_row, uint32 start_column, uint32 height,
                                 uint32 width, CompressedImage *subimage) {
  if (!IsValidCompressedImage(image) || !subimage) return false;
  return Subimage4x4(image, ICAMD_COMPRESSOR_ETC, 8, start_row, start_column, height, width, subimage);
}

// ----------------------------------------------------------------- PVRTC

PvrtcCompressor::PvrtcCompressor() {}
PvrtcCompressor::~PvrtcCompressor() {}

bool PvrtcCompressor::SupportsFormat(CompressedImage::Format format) const {
  return icamd_supports_format(ICAMD_COMPRESSOR_PVRTC, format) != 0;
}

bool PvrtcCompressor::IsValidCompressedImage(const CompressedImage &image) {  // pvrtc_compressor.cc:611-629
  const CompressedImage::Metadata &m = image.GetMetadata();
  const uint32 h = m.uncompressed_height, w = m.uncompressed_width;
  return m.format == CompressedImage::kRGBA && m.compressor_name == "pvrtc" && h >= 4 && w >= 8 &&
         m.compressed_width == m.compressed_height && h != 0 && !(h & (h - 1)) && w != 0 && !(w & (w - 1)) &&
         m.compressed_height == h && m.compressed_width == w &&
         image.GetDataSize() == ComputeCompressedDataSize(m.format, h, w);
}

size_t PvrtcCompressor::ComputeCompressedDataSize(CompressedImage::Format format, uint32 height, uint32 width) {
  return icamd_compute_compressed_data_size(ICAMD_COMPRESSOR_PVRTC, format, height, width);
}

bool PvrtcCompressor::Compress(CompressedImage::Format format, uint32 height, uint32 width,
                               uint32 padding_bytes_per_row, const uint8 *buffer, CompressedImage *image) {
  // pvrtc_compressor.cc:636-650 (the format argument is not validated there either)
  if (!buffer || !image || height == 0 || width == 0) return false;
  if ((width & (width - 1)) || (height & (height - 1)) || width != height) return false;
  if (padding_bytes_per_row != 0 || width % 8 != 0 || height % 4 != 0) return false;
  const size_t data_size = ComputeCompressedDataSize(format, height, width);
  const CompressedImage::Metadata metadata(format, "pvrtc", height, width, height, width, 0);
  if (!PrepareImage(metadata, data_size, image)) return false;
  return ReportStatus(icamd_compress(ICAMD_COMPRESSOR_PVRTC, 0, format, height, width, 0, buffer,
                                     image->GetMutableData(), data_size),
                      "icamd_compress");
}

// pvrtc_compressor.cc:669-705: the reference implements none of these for PVRTC.
// pvrtc_compressor.cc:669-672: `return false`.  Opt-in extension (ICAMD_PVRTC_DECOMPRESS_EXTENSION=1): the decoder
// written from the encoder's own rules (include/ic_amd.h, icamd_pvrtc2_decompress; parity unpinned).
bool PvrtcCompressor::Decompress(const CompressedImage &image, std::vector<uint8> *decompressed_buffer) {
  const char *opt = std::getenv("ICAMD_PVRTC_DECOMPRESS_EXTENSION");
  if (!opt || opt[0] != '1' || !decompressed_buffer || !IsValidCompressedImage(image)) return false;
  const uint32 n = image.GetMetadata().uncompressed_width;
  decompressed_buffer->assign((size_t)n * n * 4, 0);
  return ReportStatus(icamd_pvrtc2_decompress(n, image.GetData(), image.GetDataSize(), decompressed_buffer->data(),
                                              decompressed_buffer->size()),
                      "icamd_pvrtc2_decompress");
}
bool PvrtcCompressor::Downsample(const CompressedImage &, CompressedImage *) { return false; }
bool PvrtcCompressor::Pad(const CompressedImage &, uint32, uint32, CompressedImage *) { return false; }
bool PvrtcCompressor::CompressAndPad(CompressedImage::Format, uint32, uint32, uint32, uint32, uint32, const uint8 *,
                                     CompressedImage *) {
  return false;
}
bool PvrtcCompressor::CreateSolidImage(CompressedImage::Format, uint32, uint32, const uint8 *, CompressedImage *) {
  return false;
}
bool PvrtcCompressor::CopySubimage(const CompressedImage &, uint32, uint32, uint32, uint32, CompressedImage *) {
  return false;
}

// ------------------------------------------- extension: device-resident hot path (compressor.h, not in the reference)

namespace {
// Compress / CompressAndPad of one image whose pixels and output live in HBM.
bool DeviceOne(int compressor, int etc_strategy, CompressedImage::Format format, uint32 height, uint32 width,
               uint32 padded_height, uint32 padded_width, uint32 padding_bytes_per_row, const void *d_buffer, void *d_out,
               size_t out_size, void *hip_stream, bool and_pad) {
  if (!d_buffer || !d_out || height == 0 || width == 0) return false;
  return and_pad ? ReportStatus(icamd_compress_and_pad_device(compressor, etc_strategy, format, height, width, padded_height,
                                                              padded_width, padding_bytes_per_row, d_buffer, d_out, out_size,
                                                              hip_stream),
                                "icamd_compress_and_pad_device")
                 : ReportStatus(icamd_compress_device(compressor, etc_strategy, format, height, width, padding_bytes_per_row,
                                                      d_buffer, d_out, out_size, hip_stream),
                                "icamd_compress_device");
}
// n equally shaped images, one launch.  codec / components as DxtcCompressor / EtcCompressor / PvrtcCompressor::Compress map
// the format (dxtc_compressor.cc:741-749, etc_compressor.cc:751-754, pvrtc_compressor.cc:664).
bool DeviceBatch(int compressor, int etc_strategy, CompressedImage::Format format, uint32 height, uint32 width,
                 uint32 padding_bytes_per_row, uint32 n_images, const void *d_buffer, size_t src_image_stride_bytes, void *d_out,
                 size_t dst_image_stride_bytes, size_t out_size_per_image, void *hip_stream) {
  if (!d_buffer || !d_out || height == 0 || width == 0) return false;
  if (!icamd_supports_format(compressor, format) && compressor != ICAMD_COMPRESSOR_PVRTC) return false;
  if (out_size_per_image != icamd_compute_compressed_data_size(compressor, format, height, width)) return false;
  if (n_images > 1 && dst_image_stride_bytes < out_size_per_image) return false;
  const int comps = compressor == ICAMD_COMPRESSOR_PVRTC ? 4 : (int)GetNumFormatComponents(format);
  const int codec = compressor == ICAMD_COMPRESSOR_PVRTC ? ICAMD_PVRTC2
                    : compressor == ICAMD_COMPRESSOR_ETC ? ICAMD_ETC1 : (comps == 3 ? ICAMD_DXT1 : ICAMD_DXT5);
  const int swap = (format == CompressedImage::kBGR || format == CompressedImage::kBGRA) ? 1 : 0;
  const uint32 stride = width * (uint32)comps + padding_bytes_per_row;
  return ReportStatus(icamd_encode_device(codec, etc_strategy, comps, swap, height, width, height, width, stride, n_images,
                                          src_image_stride_bytes, dst_image_stride_bytes, d_buffer, d_out, hip_stream),
                      "icamd_encode_device");
}
}  // namespace

#define ICAMD_DEFINE_DEVICE_EXTENSION(CLASS, COMPRESSOR, STRATEGY)                                                             \
  bool CLASS::CompressDevice(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,        \
                             const void *d_buffer, void *d_out, size_t out_size, void *hip_stream) {                           \
    return DeviceOne(COMPRESSOR, STRATEGY, format, height, width, height, width, padding_bytes_per_row, d_buffer, d_out,       \
                     out_size, hip_stream, false);                                                                             \
  }                                                                                                                            \
  bool CLASS::CompressAndPadDevice(CompressedImage::Format format, uint32 height, uint32 width, uint32 padded_height,          \
                                   uint32 padded_width, uint32 padding_bytes_per_row, const void *d_buffer, void *d_out,       \
                                   size_t out_size, void *hip_stream) {                                                        \
    return DeviceOne(COMPRESSOR, STRATEGY, format, height, width, padded_height, padded_width, padding_bytes_per_row, d_buffer,\
                     d_out, out_size, hip_stream, true);                                                                       \
  }                                                                                                                            \
  bool CLASS::CompressBatchDevice(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,   \
                                  uint32 n_images, const void *d_buffer, size_t src_image_stride_bytes, void *d_out,           \
                                  size_t dst_image_stride_bytes, size_t out_size_per_image, void *hip_stream) {                \
    return DeviceBatch(COMPRESSOR, STRATEGY, format, height, width, padding_bytes_per_row, n_images, d_buffer,                 \
                       src_image_stride_bytes, d_out, dst_image_stride_bytes, out_size_per_image, hip_stream);                 \
  }
ICAMD_DEFINE_DEVICE_EXTENSION(DxtcCompressor, ICAMD_COMPRESSOR_DXTC, 0)
ICAMD_DEFINE_DEVICE_EXTENSION(EtcCompressor, ICAMD_COMPRESSOR_ETC, compression_strategy_)
ICAMD_DEFINE_DEVICE_EXTENSION(PvrtcCompressor, ICAMD_COMPRESSOR_PVRTC, 0)
#undef ICAMD_DEFINE_DEVICE_EXTENSION

// ------------------------------------------------------------ transcoder

void TranscodeDxt1ToEtc1(CompressedImage *image) {  // dxtc_to_etc_transcoder.cc:29-40: in place, data only
  ReportStatus(icamd_transcode_dxt1_to_etc1(image->GetMutableData(), image->GetDataSize()),
               "icamd_transcode_dxt1_to_etc1");
}

}  // namespace image_codec_compression
