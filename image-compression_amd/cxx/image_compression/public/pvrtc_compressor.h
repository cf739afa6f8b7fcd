// PvrtcCompressor: PVRTC1 at 2 bits per pixel, RGBA, square power-of-two images only (reference public/pvrtc_compressor.h:71-104).
// Only Compress and ComputeCompressedDataSize do anything; the reference implements nothing else for PVRTC either.
// MI355X backend: Compress / CompressAndPad / Decompress / Pad / Downsample run as HIP kernels through the C ABI of
// include/ic_amd.h; the class itself only validates arguments and sets up the CompressedImage.
#ifndef IMAGE_COMPRESSION_PUBLIC_PVRTC_COMPRESSOR_H_
#define IMAGE_COMPRESSION_PUBLIC_PVRTC_COMPRESSOR_H_

#include "image_compression/public/compressor.h"

namespace image_codec_compression {

class PvrtcCompressor : public Compressor {
 public:
  PvrtcCompressor();
  virtual ~PvrtcCompressor();

  ICAMD_DECLARE_COMPRESSOR_OVERRIDES();
  ICAMD_DECLARE_DEVICE_EXTENSION();  // extension: device-resident hot path (compressor.h)
};

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_PVRTC_COMPRESSOR_H_
