// CompressedImage -- what every Compressor call produces: a byte buffer of encoded blocks plus the metadata needed
// to interpret it.  The buffer is either owned by the instance (allocated by the compressor) or borrowed from the
// caller, who then guarantees size and lifetime.  API-compatible with the reference's
// image_compression/public/compressed_image.h (:32-204): same names, same semantics, same non-copyability.
#ifndef IMAGE_COMPRESSION_PUBLIC_COMPRESSED_IMAGE_H_
#define IMAGE_COMPRESSION_PUBLIC_COMPRESSED_IMAGE_H_

#include <stddef.h>

#include <cstring>
#include <string>

#include "base/integral_types.h"
#include "base/logging.h"

namespace image_codec_compression {

class CompressedImage {
 public:
  // Channel order of the uncompressed pixels.  The numeric values cross the C ABI (include/ic_amd.h).
  enum Format { kRGB, kBGR, kRGBA, kBGRA };

  // Everything known about the image except its bytes.
  struct Metadata {
    Format format;
    std::string compressor_name;   // "dxtc", "etc" or "pvrtc"
    uint32 uncompressed_height;    // source image, pixels
    uint32 uncompressed_width;
    uint32 compressed_height;      // pixels covered by the block grid
    uint32 compressed_width;
    uint32 padding_bytes_per_row;  // of the source rows; reused by Decompress

    Metadata(Format f, const std::string &name, uint32 uh, uint32 uw, uint32 ch, uint32 cw, uint32 row_padding)
        : format(f), compressor_name(name), uncompressed_height(uh), uncompressed_width(uw), compressed_height(ch),
          compressed_width(cw), padding_bytes_per_row(row_padding) {}
  };

  CompressedImage();                                        // empty; will own what a Compressor allocates
  CompressedImage(size_t data_size, uint8 *external_data);  // over caller storage; never freed here
  ~CompressedImage();

  void Duplicate(const CompressedImage &from);                             // deep copy; result is owned
  void CreateOwnedData(const Metadata &metadata, size_t data_size);        // fresh owned buffer
  void SetMetadata(const Metadata &metadata) { metadata_ = metadata; }     // for external storage

  const Metadata &GetMetadata() const { return metadata_; }
  bool OwnsData() const { return owned_; }
  size_t GetDataSize() const { return size_; }
  const uint8 *GetData() const { return bytes_; }
  uint8 *GetMutableData() { return bytes_; }

 private:
  CompressedImage(const CompressedImage &);  // use Duplicate()
  void operator=(const CompressedImage &);

  void Release() {
    if (owned_) delete[] bytes_;
  }

  Metadata metadata_;
  uint8 *bytes_;
  size_t size_;
  bool owned_;
};

inline CompressedImage::CompressedImage() : metadata_(kRGB, "", 0, 0, 0, 0, 0), bytes_(NULL), size_(0), owned_(true) {}

inline CompressedImage::CompressedImage(size_t data_size, uint8 *external_data)
    : metadata_(kRGB, "", 0, 0, 0, 0, 0), bytes_(external_data), size_(data_size), owned_(false) {}

inline CompressedImage::~CompressedImage() { Release(); }

inline void CompressedImage::CreateOwnedData(const Metadata &metadata, size_t data_size) {
  Release();
  metadata_ = metadata;
  bytes_ = new uint8[data_size];
  size_ = data_size;
  owned_ = true;
}

inline void CompressedImage::Duplicate(const CompressedImage &from) {
  if (&from == this && owned_) return;  // self-copy of owned data: nothing to do
  const uint8 *source = from.bytes_;    // read before CreateOwnedData may replace it (self-copy of borrowed data)
  CreateOwnedData(from.metadata_, from.size_);
  std::memcpy(bytes_, source, size_);
}

inline int GetNumFormatComponents(CompressedImage::Format format) {
  switch (format) {
    case CompressedImage::kRGB: case CompressedImage::kBGR: return 3;
    case CompressedImage::kRGBA: case CompressedImage::kBGRA: return 4;
  }
  return 0;
}

inline bool NeedsRedAndBlueSwapped(CompressedImage::Format format) {
  return format == CompressedImage::kBGR || format == CompressedImage::kBGRA;
}

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_COMPRESSED_IMAGE_H_
