// DxtcCompressor: DXT1 (3-channel formats) / DXT5 (4-channel formats); reference public/dxtc_compressor.h:52-83.
// MI355X backend: Compress / CompressAndPad / Decompress run as HIP kernels through include/ic_amd.h.
#ifndef IMAGE_COMPRESSION_PUBLIC_DXTC_COMPRESSOR_H_
#define IMAGE_COMPRESSION_PUBLIC_DXTC_COMPRESSOR_H_

#include <stddef.h>

#include <vector>

#include "base/integral_types.h"
#include "image_compression/public/compressed_image.h"
#include "image_compression/public/compressor.h"

namespace image_codec_compression {

class DxtcCompressor : public Compressor {
 public:
  DxtcCompressor();
  virtual ~DxtcCompressor();

  virtual bool SupportsFormat(CompressedImage::Format format) const;
  virtual bool IsValidCompressedImage(const CompressedImage &image);
  virtual size_t ComputeCompressedDataSize(CompressedImage::Format format, uint32 height, uint32 width);
  virtual bool Compress(CompressedImage::Format format, uint32 height, uint32 width, uint32 padding_bytes_per_row,
                        const uint8 *buffer, CompressedImage *image);
  virtual bool Decompress(const CompressedImage &image, std::vector<uint8> *decompressed_buffer);
  virtual bool Downsample(const CompressedImage &image, CompressedImage *downsampled_image);
  virtual bool Pad(const CompressedImage &image, uint32 padded_height, uint32 padded_width,
                   CompressedImage *padded_image);
  virtual bool CompressAndPad(CompressedImage::Format format, uint32 height, uint32 width, uint32 padded_height,
                              uint32 padded_width, uint32 padding_bytes_per_row, const uint8 *buffer,
                              CompressedImage *padded_image);
  virtual bool CreateSolidImage(CompressedImage::Format format, uint32 height, uint32 width, const uint8 *color,
                                CompressedImage *image);
  virtual bool CopySubimage(const CompressedImage &image, uint32 start_row, uint32 start_column, uint32 height,
                            uint32 width, CompressedImage *subimage);
};

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_DXTC_COMPRESSOR_H_
