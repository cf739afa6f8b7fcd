// DxtcCompressor: DXT1 for the 3-channel formats, DXT5 for the 4-channel ones (reference public/dxtc_compressor.h:52-83).
// MI355X backend: Compress / CompressAndPad / Decompress / Pad / Downsample run as HIP kernels through the C ABI of
// include/ic_amd.h; the class itself only validates arguments and sets up the CompressedImage.
#ifndef IMAGE_COMPRESSION_PUBLIC_DXTC_COMPRESSOR_H_
#define IMAGE_COMPRESSION_PUBLIC_DXTC_COMPRESSOR_H_

#include "image_compression/public/compressor.h"

namespace image_codec_compression {

class DxtcCompressor : public Compressor {
 public:
  DxtcCompressor();
  virtual ~DxtcCompressor();

  ICAMD_DECLARE_COMPRESSOR_OVERRIDES();
  ICAMD_DECLARE_DEVICE_EXTENSION();  // extension: device-resident hot path (compressor.h)
};

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_DXTC_COMPRESSOR_H_
