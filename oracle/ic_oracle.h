/* TEST INFRASTRUCTURE ONLY -- not part of the product, never linked into it.
 *
 * Plain-C CPU restatement of the per-block encode path of google/image-compression
 * (DXT1/DXT5, ETC1, PVRTC1 2bpp).  It exists to (a) check the HIP kernels
 * bit-for-bit on the GPU box, where /root/reference is absent, and (b) serve as
 * the timed CPU baseline in bench.py (`cpu_baseline.kind = "port"`).
 *
 * PARITY PINNING: the reference ships no tests or golden vectors.
 * This restatement is pinned against the *compiled reference itself*
 * (oracle/_ref/libic_ref.so, built by `make -C oracle _ref` from the sources
 * where they lie under /root/reference) by tests/test_oracle_vs_ref.py, and
 * against the committed fixtures under tests/golden/ that were generated from
 * that reference build by tests/golden/make_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef IC_ORACLE_H_
#define IC_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Block codecs (what is written), independent of the reference's class split. */
enum { ICO_DXT1 = 0, ICO_DXT5 = 1, ICO_ETC1 = 2, ICO_PVRTC2 = 3,
       ICO_PVRTC4 = 4 /* EXTENSION, parity unpinned: PVRTC1 4 bpp written from the 2 bpp rules (ic_oracle.c) */ };

/* Reference compressor classes and CompressedImage::Format (compressed_image.h:35-40). */
enum { ICO_COMPRESSOR_DXTC = 0, ICO_COMPRESSOR_ETC = 1, ICO_COMPRESSOR_PVRTC = 2 };
enum { ICO_RGB = 0, ICO_BGR = 1, ICO_RGBA = 2, ICO_BGRA = 3 };

/* EtcCompressor::CompressionStrategy (public/etc_compressor.h:57-62). */
enum { ICO_ETC_SPLIT_HORIZONTALLY = 0, ICO_ETC_SPLIT_VERTICALLY = 1,
       ICO_ETC_SMALLER_ERROR = 2, ICO_ETC_HEURISTIC = 3 };

/* ---- reference-shaped entry points (same argument meaning as Compressor::*) ---- */

/* Compressor::ComputeCompressedDataSize (dxtc.cc:725-733, etc.cc:734-745, pvrtc.cc:631-634). */
size_t ico_compute_compressed_data_size(int compressor, int format, uint32_t height, uint32_t width);

/* Compressor::Compress into caller storage of exactly out_size bytes.
 * Returns 1/0 like the reference's bool (0 also when out_size mismatches, the
 * external-storage rule of compressor4x4_helper.cc:34-41). */
int ico_compress(int compressor, int etc_strategy, int format, uint32_t height, uint32_t width,
                 uint32_t padding_bytes_per_row, const uint8_t *buffer, uint8_t *out, size_t out_size);

/* Compressor::CompressAndPad (helper.h:479-520). */
int ico_compress_and_pad(int compressor, int etc_strategy, int format, uint32_t height, uint32_t width,
                         uint32_t padded_height, uint32_t padded_width, uint32_t padding_bytes_per_row,
                         const uint8_t *buffer, uint8_t *out, size_t out_size);

/* ---- generic block-grid encoder (covers the RGBA8->DXT1/ETC1 extension too) ----
 * codec: ICO_*; src_components: 3 or 4 bytes per source pixel (alpha is ignored by
 * DXT1/ETC1); swap_rb: source is B,G,R(,A); grid_* >= image dims select the
 * CompressAndPad grid (pass the image dims for plain Compress).
 * threads > 1 splits block rows into slabs (used only for the CPU baseline). */
int ico_encode(int codec, int etc_strategy, int src_components, int swap_rb,
               uint32_t height, uint32_t width, uint32_t grid_height, uint32_t grid_width,
               uint32_t row_stride_bytes, const uint8_t *src, uint8_t *out, int threads);

/* Bytes ico_encode writes for that grid. */
size_t ico_encoded_size(int codec, uint32_t grid_height, uint32_t grid_width);

/* ---- decoders ("next" rows): DXT1/DXT5/ETC1 follow the reference
 * (dxtc.cc:167-267, etc.cc:198-289, helper.h:218-262).  out has
 * height * (width*comps + padding_bytes_per_row) bytes addressing, like the reference.
 * ICO_PVRTC2: an EXTENSION with PARITY UNPINNED -- the reference has no PVRTC decoder (pvrtc.cc:669-672); written
 * from the encoder's own interpolation / modulation rules (see ic_oracle.c); square power-of-two RGBA8 output. */
int ico_decode(int codec, int swap_rb, uint32_t height, uint32_t width, uint32_t padding_bytes_per_row,
               const uint8_t *blocks, uint8_t *out);

/* ---- compressed-domain operations ----
 * All take the block grid of an image whose metadata says compressed dims (ch, cw) / uncompressed dims (uh, uw).
 * Return 1/0 like the reference's bool; *out_h / *out_w receive the result's uncompressed dims. */

/* Compressor::CreateSolidImage (dxtc.cc:820-839, etc.cc:801-811, helper.h:522-543): `out` gets
 * ico_compute_compressed_data_size(compressor, format, h, w) bytes. */
int ico_create_solid(int compressor, int format, uint32_t height, uint32_t width, const uint8_t *color, uint8_t *out);

/* Compressor::CopySubimage (helper.h:545-592). */
int ico_copy_subimage(int compressor, int format, uint32_t ch, uint32_t cw, const uint8_t *blocks,
                      uint32_t start_row, uint32_t start_col, uint32_t height, uint32_t width, uint8_t *out);

/* Compressor::Pad (helper.h:393-477; pad functors dxtc.cc:594-696, etc.cc:645-698).  Returns 2 when the
 * reference only duplicates the image (both padded dims <= compressed dims).  Refuses (0) the case where exactly
 * one padded dimension is smaller than the compressed one: the reference writes out of bounds there. */
int ico_pad(int compressor, int etc_strategy, int format, uint32_t ch, uint32_t cw, const uint8_t *blocks,
            uint32_t padded_height, uint32_t padded_width, uint8_t *out);

/* Compressor::Downsample (helper.h:264-391,594-636): decodes 2x2 / 2x1 / 1x2 / 1 block(s), averages 2x2 pixels
 * (cutil.h:335-380), re-encodes.  out holds blocks for ((uh+1)/2, (uw+1)/2). */
int ico_downsample(int compressor, int etc_strategy, int format, uint32_t uh, uint32_t uw, const uint8_t *blocks,
                   uint8_t *out);

/* TranscodeDxt1ToEtc1 (dxtc_to_etc_transcoder.cc:29-40): in place, n_bytes of DXT1 blocks -> ETC1 (kHeuristic). */
void ico_transcode_dxt1_to_etc1(uint8_t *blocks, size_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif  /* IC_ORACLE_H_ */
