// EtcCompressor: ETC1, kRGB input only (reference public/etc_compressor.h:53-109).
// MI355X backend: Compress / CompressAndPad / Decompress / Pad / Downsample run as HIP kernels through the C ABI of
// include/ic_amd.h; the class itself only validates arguments and sets up the CompressedImage.
#ifndef IMAGE_COMPRESSION_PUBLIC_ETC_COMPRESSOR_H_
#define IMAGE_COMPRESSION_PUBLIC_ETC_COMPRESSOR_H_

#include "image_compression/public/compressor.h"

namespace image_codec_compression {

class EtcCompressor : public Compressor {
 public:
  EtcCompressor();
  virtual ~EtcCompressor();

  // How a 4x4 block is cut into its two sub-blocks (values are part of the API: public/etc_compressor.h:57-62).
  enum CompressionStrategy {
    kSplitHorizontally,  // top | bottom (flip bit set)
    kSplitVertically,    // left | right
    kSmallerError,       // both; keep the smaller squared error.  Default.
    kHeuristic,          // split and codewords from cheap statistics
  };
  void SetCompressionStrategy(CompressionStrategy strategy) { compression_strategy_ = strategy; }
  CompressionStrategy GetCompressionStrategy() const { return compression_strategy_; }

  ICAMD_DECLARE_COMPRESSOR_OVERRIDES();
  ICAMD_DECLARE_DEVICE_EXTENSION();  // extension: device-resident hot path (compressor.h)

 private:
  CompressionStrategy compression_strategy_;
};

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_ETC_COMPRESSOR_H_
