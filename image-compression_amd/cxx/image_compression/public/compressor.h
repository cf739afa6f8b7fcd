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

#endif  // IMAGE_COMPRESSION_PUBLIC_COMPRESSOR_H_
