// containers.h -- file-container framing for the encoders' block streams (host code, no device work).
//
// SURVEY.md §8(f) row 4, tail: "container formats (DDS/KTX/PVR headers) which the reference does not have at all".
// google/image-compression stops at the raw block stream (CompressedImage, compressed_image.h:52-66 carries format +
// dimensions only), so there is NOTHING in the reference to pin these against: parity unpinned by construction.  The
// layouts below are written from the public file-format descriptions:
//   * DDS  -- "DDS " magic + 124-byte DDS_HEADER with a FOURCC pixel format (DXT1 / DXT5 only; the legacy header has
//             no code for ETC1 or PVRTC);
//   * KTX  -- KTX 1.1: 12-byte identifier, 13 little-endian uint32 fields, then per mip level uint32 imageSize + data
//             (glInternalFormat: COMPRESSED_RGB_S3TC_DXT1_EXT 0x83F0, COMPRESSED_RGBA_S3TC_DXT5_EXT 0x83F3,
//             ETC1_RGB8_OES 0x8D64, COMPRESSED_RGBA_PVRTC_2BPPV1_IMG 0x8C03);
//   * PKM  -- "PKM 10": 16-byte big-endian header, ETC1 only, exactly one level;
//   * PVR  -- PVR v3: 52-byte little-endian header (pixel format 1 = PVRTC 2bpp RGBA, 6 = ETC1, 7 = DXT1, 11 = DXT5),
//             no metadata, levels follow largest first.  PVRTC data is stored in the Z-order the encoder already
//             produces (pvrtc_compressor.cc:551-580).
// Mip level l of an h x w texture is max(1, h >> l) x max(1, w >> l) pixels; its block stream is what
// icamd_compress / icamd_downsample produce for that size (compressor4x4_helper.h:594-636 halves the same way).
#ifndef ICAMD_CONTAINERS_H_
#define ICAMD_CONTAINERS_H_

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "ic_amd.h"

namespace icamd {

inline void put_le32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
inline void put_le64(uint8_t *p, uint64_t v) {
  put_le32(p, (uint32_t)v);
  put_le32(p + 4, (uint32_t)(v >> 32));
}
inline void put_be16(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v;
}

inline size_t container_header_size(int container) {
  switch (container) {
    case ICAMD_CONTAINER_DDS: return 128;
    case ICAMD_CONTAINER_KTX: return 64;
    case ICAMD_CONTAINER_PKM: return 16;
    case ICAMD_CONTAINER_PVR: return 52;
    default: return 0;
  }
}

inline bool container_supports(int container, int codec) {
  switch (container) {
    case ICAMD_CONTAINER_DDS: return codec == ICAMD_DXT1 || codec == ICAMD_DXT5;
    case ICAMD_CONTAINER_KTX:
    case ICAMD_CONTAINER_PVR: return codec == ICAMD_DXT1 || codec == ICAMD_DXT5 || codec == ICAMD_ETC1 || codec == ICAMD_PVRTC2;
    case ICAMD_CONTAINER_PKM: return codec == ICAMD_ETC1;
    default: return false;
  }
}

// bytes of one mip level's block stream; 0 if the level cannot exist for this codec
inline size_t container_level_bytes(int codec, uint32_t height, uint32_t width, uint32_t level) {
  if (level >= 32) return 0;
  const uint32_t h = (height >> level) ? (height >> level) : 1u, w = (width >> level) ? (width >> level) : 1u;
  if (codec == ICAMD_PVRTC2) {  // the encoder's own domain: square powers of two, 8 x 8 and up (pvrtc.cc:636-652)
    if ((height >> level) < 8 || (width >> level) < 8) return 0;
    return (size_t)w * h / 4;
  }
  return (size_t)((h + 3) / 4) * ((w + 3) / 4) * (codec == ICAMD_DXT5 ? 16u : 8u);
}

// framing bytes in front of every level (KTX: the uint32 imageSize)
inline size_t container_level_prefix(int container) { return container == ICAMD_CONTAINER_KTX ? 4 : 0; }

// Writes the header of `container` into out (container_header_size bytes).  level0_bytes = size of the first level.
inline void container_write_header(int container, int codec, uint32_t height, uint32_t width, uint32_t levels,
                                   size_t level0_bytes, uint8_t *out) {
  memset(out, 0, container_header_size(container));
  switch (container) {
    case ICAMD_CONTAINER_DDS: {
      memcpy(out, "DDS ", 4);
      put_le32(out + 4, 124);                                                            // dwSize
      put_le32(out + 8, 0x1u | 0x2u | 0x4u | 0x1000u | 0x80000u | (levels > 1 ? 0x20000u : 0u));  // CAPS HEIGHT WIDTH PIXELFORMAT LINEARSIZE [MIPMAPCOUNT]
      put_le32(out + 12, height);
      put_le32(out + 16, width);
      put_le32(out + 20, (uint32_t)level0_bytes);                                        // dwPitchOrLinearSize
      put_le32(out + 28, levels);                                                        // dwMipMapCount
      put_le32(out + 76, 32);                                                            // ddspf.dwSize
      put_le32(out + 80, 0x4u);                                                          // DDPF_FOURCC
      memcpy(out + 84, codec == ICAMD_DXT1 ? "DXT1" : "DXT5", 4);
      put_le32(out + 108, 0x1000u | (levels > 1 ? 0x8u | 0x400000u : 0u));               // DDSCAPS_TEXTURE [COMPLEX MIPMAP]
      break;
    }
    case ICAMD_CONTAINER_KTX: {
      static const uint8_t id[12] = { 0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A };
      memcpy(out, id, 12);
      put_le32(out + 12, 0x04030201u);  // endianness
      put_le32(out + 16, 0);            // glType (compressed)
      put_le32(out + 20, 1);            // glTypeSize
      put_le32(out + 24, 0);            // glFormat (compressed)
      const uint32_t internal = codec == ICAMD_DXT1 ? 0x83F0u : codec == ICAMD_DXT5 ? 0x83F3u : codec == ICAMD_ETC1 ? 0x8D64u : 0x8C03u;
      put_le32(out + 28, internal);
      put_le32(out + 32, (codec == ICAMD_DXT5 || codec == ICAMD_PVRTC2) ? 0x1908u : 0x1907u);  // GL_RGBA / GL_RGB
      put_le32(out + 36, width);
      put_le32(out + 40, height);
      put_le32(out + 44, 0);            // pixelDepth
      put_le32(out + 48, 0);            // numberOfArrayElements
      put_le32(out + 52, 1);            // numberOfFaces
      put_le32(out + 56, levels);
      put_le32(out + 60, 0);            // bytesOfKeyValueData
      break;
    }
    case ICAMD_CONTAINER_PKM: {
      memcpy(out, "PKM 10", 6);
      put_be16(out + 6, 0);                        // ETC1_RGB_NO_MIPMAPS
      put_be16(out + 8, (width + 3u) & ~3u);       // encoded (block-aligned) size
      put_be16(out + 10, (height + 3u) & ~3u);
      put_be16(out + 12, width);                   // original size
      put_be16(out + 14, height);
      break;
    }
    case ICAMD_CONTAINER_PVR: {
      put_le32(out, 0x03525650u);  // "PVR\3"
      put_le32(out + 4, 0);        // flags
      put_le64(out + 8, codec == ICAMD_PVRTC2 ? 1u : codec == ICAMD_ETC1 ? 6u : codec == ICAMD_DXT1 ? 7u : 11u);
      put_le32(out + 16, 0);       // colour space: linear RGB
      put_le32(out + 20, 0);       // channel type: unsigned byte, normalised
      put_le32(out + 24, height);
      put_le32(out + 28, width);
      put_le32(out + 32, 1);       // depth
      put_le32(out + 36, 1);       // surfaces
      put_le32(out + 40, 1);       // faces
      put_le32(out + 44, levels);
      put_le32(out + 48, 0);       // metadata bytes
      break;
    }
    default: break;
  }
}

}  // namespace icamd

#endif  // ICAMD_CONTAINERS_H_
