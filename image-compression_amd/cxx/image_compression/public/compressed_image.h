// CompressedImage: the result container of every Compressor call -- a byte buffer (owned by the
// instance, or borrowed from the caller) plus the metadata needed to interpret it.
// API-compatible with the reference's image_compression/public/compressed_image.h (:32-204).
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
  // Channel layout of the *uncompressed* pixels.
  enum Format {
    kRGB,   // 3 bytes per pixel: R, G, B
    kBGR,   // 3 bytes per pixel: B, G, R
    kRGBA,  // 4 bytes per pixel: R, G, B, A
    kBGRA,  // 4 bytes per pixel: B, G, R, A
  };

  struct Metadata {
    Metadata(Format format_in, const std::string &compressor_name_in, uint32 uncompressed_height_in,
             uint32 uncompressed_width_in, uint32 compressed_height_in, uint32 compressed_width_in,
             uint32 padding_bytes_per_row_in)
        : format(format_in),
          compressor_name(compressor_name_in),
          uncompressed_height(uncompressed_height_in),
          uncompressed_width(uncompressed_width_in),
          compressed_height(compressed_height_in),
          compressed_width(compressed_width_in),
          padding_bytes_per_row(padding_bytes_per_row_in) {}

    Format format;
    std::string compressor_name;   // "dxtc", "etc" or "pvrtc"
    uint32 uncompressed_height;    // pixels of the source image
    uint32 uncompressed_width;
    uint32 compressed_height;      // pixels covered by the block grid (multiples of the block size)
    uint32 compressed_width;
    uint32 padding_bytes_per_row;  // extra bytes per source row; reused when decompressing
  };

  // Empty image that will own whatever a Compressor allocates for it.
  CompressedImage() : metadata_(kRGB, "", 0, 0, 0, 0, 0), size_(0), bytes_(NULL), owned_(true) {}

  // Image over caller-managed storage: Compressors write into it and never free it.
  CompressedImage(size_t data_size, uint8 *external_data)
      : metadata_(kRGB, "", 0, 0, 0, 0, 0), size_(data_size), bytes_(external_data), owned_(false) {}

  ~CompressedImage() {
    if (owned_) delete[] bytes_;
  }

  // Deep copy (metadata and bytes); this instance owns the copy afterwards.
  void Duplicate(const CompressedImage &from) {
    if (&from == this && owned_) return;
    const uint8 *src = from.bytes_;
    CreateOwnedData(from.metadata_, from.size_);
    std::memcpy(bytes_, src, size_);
  }

  // Replaces the contents by a freshly allocated, owned buffer of data_size bytes.
  void CreateOwnedData(const Metadata &metadata, size_t data_size) {
    if (owned_) delete[] bytes_;
    metadata_ = metadata;
    size_ = data_size;
    bytes_ = new uint8[data_size];
    owned_ = true;
  }

  // For instances over external storage.
  void SetMetadata(const Metadata &metadata) { metadata_ = metadata; }

  const Metadata &GetMetadata() const { return metadata_; }
  bool OwnsData() const { return owned_; }
  size_t GetDataSize() const { return size_; }
  const uint8 *GetData() const { return bytes_; }
  uint8 *GetMutableData() { return bytes_; }

 private:
  Metadata metadata_;
  size_t size_;
  uint8 *bytes_;
  bool owned_;

  CompressedImage(const CompressedImage &);  // non-copyable: use Duplicate()
  void operator=(const CompressedImage &);
};

inline int GetNumFormatComponents(CompressedImage::Format format) {
  return (format == CompressedImage::kRGB || format == CompressedImage::kBGR)     ? 3
         : (format == CompressedImage::kRGBA || format == CompressedImage::kBGRA) ? 4
                                                                                   : 0;
}

inline bool NeedsRedAndBlueSwapped(CompressedImage::Format format) {
  return format == CompressedImage::kBGR || format == CompressedImage::kBGRA;
}

}  // namespace image_codec_compression

#endif  // IMAGE_COMPRESSION_PUBLIC_COMPRESSED_IMAGE_H_
