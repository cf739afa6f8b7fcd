// Compressor: abstract interface of the block texture codecs.  API-compatible with the reference's
// image_compression/public/compressor.h (:48-138); here every implementation runs on an MI355X.
//
// Uncompressed images are 8 bits per channel, 3 (RGB/BGR) or 4 (RGBA/BGRA) interleaved channels,
// row-major, optionally with padding bytes after each row.  Functions that produce a CompressedImage
// accept either a default-constructed instance (the function allocates) or one built over caller
// storage of exactly ComputeCompressedDataSize() bytes.  All failures are reported as `false`.
#ifndef IMAGE_COMPRESSION_PUBLIC_COMPRESSOR_H_
#define IMAGE_COMPRESSION_PUBLIC_COMPRESSOR_H_

#include <stddef.h>

#include <vector>

#include "base/integral_types.h"
#include "image_compression/public/compressed_image.h"

namespace image_codec_compression {

class Compressor {
 public:
  virtual ~Compressor() {}

  virtual bool SupportsFormat(CompressedImage::Format format) const = 0;

  // True iff `image` carries this compressor's name and self-consistent sizes.
  virtual bool IsValidCompressedImage(const CompressedImage &image) = 0;

  // Bytes Compress() produces for an image of that format and size (0 if unsupported).
  virtual size_t ComputeCompressedDataSize(CompressedImage::Format format, uint32 height, uint32 width) = 0;

  // THE HOT PATH.  Encodes `height` rows of `width` pixels (+ padding_bytes_per_row) from `buffer`.
  virtual bool Compress(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,
                        const uint8 *buffer, CompressedImage *image) = 0;

  virtual bool Decompress(const CompressedImage &image, std::vector<uint8> *decompressed_buffer) = 0;

  // Halves both dimensions ((n + 1) / 2) in the compressed domain.
  virtual bool Downsample(const CompressedImage &image, CompressedImage *downsampled_image) = 0;

  // Grows the image by replicating its last row / column.
  virtual bool Pad(const CompressedImage &image, uint32 padded_height, uint32 padded_width,
                   CompressedImage *padded_image) = 0;

  // Compress() over a block grid that may extend past the image (edge pixels are replicated).
  virtual bool CompressAndPad(CompressedImage::Format format, uint32 height, uint32 width, uint32 padded_height,
                              uint32 padded_width, uint32 padding_bytes_per_row, const uint8 *buffer,
                              CompressedImage *padded_image) = 0;

  virtual bool CreateSolidImage(CompressedImage::Format format, uint32 height, uint32 width, const uint8 *color,
                                CompressedImage *image) = 0;

  virtual bool CopySubimage(const CompressedImage &image, uint32 start_row, uint32 start_column, uint32 height,
                            uint32 width, CompressedImage *subimage) = 0;
};

}  // namespace image_codec_compression

// Every concrete compressor overrides the same ten methods; the class headers expand this list instead of
// repeating it.
#define ICAMD_DECLARE_COMPRESSOR_OVERRIDES()                                                                          \
  virtual bool SupportsFormat(CompressedImage::Format format) const;                                                  \
  virtual bool IsValidCompressedImage(const CompressedImage &image);                                                  \
  virtual size_t ComputeCompressedDataSize(CompressedImage::Format format, uint32 height, uint32 width);              \
  virtual bool Compress(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,    \
                        const uint8 *buffer, CompressedImage *image);                                                 \
  virtual bool Decompress(const CompressedImage &image, std::vector<uint8> *decompressed_buffer);                     \
  virtual bool Downsample(const CompressedImage &image, CompressedImage *downsampled_image);                          \
  virtual bool Pad(const CompressedImage &image, uint32 padded_height, uint32 padded_width,                           \
                   CompressedImage *padded_image);                                                                    \
  virtual bool CompressAndPad(CompressedImage::Format format, uint32 height, uint32 width, uint32 padded_height,      \
                              uint32 padded_width, uint32 padding_bytes_per_row, const uint8 *buffer,                 \
                              CompressedImage *padded_image);                                                         \
  virtual bool CreateSolidImage(CompressedImage::Format format, uint32 height, uint32 width, const uint8 *color,      \
                                CompressedImage *image);                                                              \
  virtual bool CopySubimage(const CompressedImage &image, uint32 start_row, uint32 start_column, uint32 height,       \
                            uint32 width, CompressedImage *subimage)

// EXTENSION of this backend (NOT part of the reference's interface, so not virtual and not on the base class: a caller that
// re-links unchanged never sees it).  The device-resident forms of the hot path, for callers whose pixels already live in
// HBM -- the path DESIGN.md section 5 measures (1.4 Tpixel/s DXT1) instead of the PCIe-bound host-buffer Compress
// (15 Gpixel/s).  d_buffer / d_out are device pointers on the current HIP device; the work is enqueued on `hip_stream`
// (a hipStream_t passed as void *, NULL = the default stream) and NOT synchronised.  out_size is the caller's storage per
// image and must equal ComputeCompressedDataSize() of the (padded) image, like external CompressedImage storage
// (compressor4x4_helper.cc:34-41).  Argument meaning and the `false` cases are those of Compress / CompressAndPad
// (public/compressor.h:77-80, 114-119); a device failure is reported on stderr and returned as false.
//   CompressDevice         one image                                      -> icamd_compress_device
//   CompressAndPadDevice   one image over a larger block grid             -> icamd_compress_and_pad_device
//   CompressBatchDevice    n_images equally shaped images in ONE launch, image i at d_buffer + i * src_image_stride_bytes ->
//                          d_out + i * dst_image_stride_bytes              -> icamd_encode_device
#define ICAMD_DECLARE_DEVICE_EXTENSION()                                                                              \
  bool CompressDevice(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,     \
                      const void *d_buffer, void *d_out, size_t out_size, void *hip_stream);                         \
  bool CompressAndPadDevice(CompressedImage::Format format, uint32 height, uint32 width, uint32 padded_height,       \
                            uint32 padded_width, uint32 padding_bytes_per_row, const void *d_buffer, void *d_out,    \
                            size_t out_size, void *hip_stream);                                                      \
  bool CompressBatchDevice(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,\
                           uint32 n_images, const void *d_buffer, size_t src_image_stride_bytes, void *d_out,        \
                           size_t dst_image_stride_bytes, size_t out_size_per_image, void *hip_stream)

#endif  // IMAGE_COMPRESSION_PUBLIC_COMPRESSOR_H_
