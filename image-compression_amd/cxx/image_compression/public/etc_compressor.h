// EtcCompressor: ETC1, kRGB input only; reference public/etc_compressor.h:53-109.
// MI355X backend: Compress / CompressAndPad / Decompress run as HIP kernels through include/ic_amd.h.
#ifndef IMAGE_COMPRESSION_PUBLIC_ETC_COMPRESSOR_H_
#define IMAGE_COMPRESSION_PUBLIC_ETC_COMPRESSOR_H_

#include <stddef.h>

#include <vector>

#include "base/integral_types.h"
#include "image_compression/public/compressed_image.h"
#include "image_compression/public/compressor.h"

namespace image_codec_compression {

class EtcCompressor : public Compressor {
 public:
  // How each 4x4 block is split into its two 2x4 / 4x2 sub-blocks.
  enum CompressionStrategy {
    kSplitHorizontally,  // top | bottom halves (flip bit set)
    kSplitVertically,    // left | right halves
    kSmallerError,       // try both, keep the smaller squared error (default)
    kHeuristic,          // pick the split and the codewords from cheap statistics
  };

  EtcCompressor();
  virtual ~EtcCompressor();

  void SetCompressionStrategy(CompressionStrategy strategy) { compression_strategy_ = strategy; }
  CompressionStrategy GetCompressionStrategy() const { return compression_strategy_; }

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

 private:
  CompressionStrategy compression_strategy_;
};

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_ETC_COMPRESSOR_H_
