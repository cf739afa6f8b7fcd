// In-place DXT1 -> ETC1 transcoding (reference public/dxtc_to_etc_transcoder.h:24): every 8-byte DXT1 block of the
// image data is decoded and re-encoded as ETC1 with the kHeuristic strategy.  Metadata is left untouched.
// MI355X backend: one HIP kernel over the block array.
#ifndef IMAGE_COMPRESSION_PUBLIC_DXTC_TO_ETC_TRANSCODER
#define IMAGE_COMPRESSION_PUBLIC_DXTC_TO_ETC_TRANSCODER

#include "image_compression/public/compressed_image.h"

namespace image_codec_compression {

void TranscodeDxt1ToEtc1(CompressedImage *image);

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_DXTC_TO_ETC_TRANSCODER
