/* ic_amd.h -- C ABI of the MI355X (gfx950) texture block-encode backend.
 *
 * Drop-in boundary for the per-4x4-block encode path of google/image-compression:
 * these entry points are what a C FFI for `image_codec_compression::Compressor`
 * (reference image_compression/public/compressor.h:48-138) binds for the hot path,
 * and what our own C++ `DxtcCompressor / EtcCompressor / PvrtcCompressor` classes
 * (image-compression_amd/cxx/) call.  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.  INTEGRATION.md shows the reference-side binding.
 *
 * There is NO CPU fallback behind this ABI: every encode entry point runs the
 * hand-written HIP kernels on the current HIP device, and returns a negative
 * ICAMD_ERR_* (never silently a host result) when no device / kernel is available.
 */
#ifndef IC_AMD_H_
#define IC_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enumerations (values match the reference's enums so they can be passed through) ---- */

/* Which reference Compressor subclass is being replaced. */
enum { ICAMD_COMPRESSOR_DXTC = 0,   /* public/dxtc_compressor.h:52-83  */
       ICAMD_COMPRESSOR_ETC = 1,    /* public/etc_compressor.h:53-109  */
       ICAMD_COMPRESSOR_PVRTC = 2 };/* public/pvrtc_compressor.h:71-104 */

/* CompressedImage::Format, public/compressed_image.h:35-40. */
enum { ICAMD_RGB = 0, ICAMD_BGR = 1, ICAMD_RGBA = 2, ICAMD_BGRA = 3 };

/* EtcCompressor::CompressionStrategy, public/etc_compressor.h:57-62. */
enum { ICAMD_ETC_SPLIT_HORIZONTALLY = 0, ICAMD_ETC_SPLIT_VERTICALLY = 1,
       ICAMD_ETC_SMALLER_ERROR = 2, ICAMD_ETC_HEURISTIC = 3 };

/* Block codec actually written (the reference derives it from compressor + format,
 * internal/dxtc_compressor.cc:741-749). */
enum { ICAMD_DXT1 = 0, ICAMD_DXT5 = 1, ICAMD_ETC1 = 2, ICAMD_PVRTC2 = 3,
       /* EXTENSION, PARITY UNPINNED: PVRTC1 4 bpp (4 x 4-pixel blocks, 64 bits each).  BASELINE.json's config 5 names it, the
        * reference only implements 2 bpp (public/pvrtc_compressor.h:15-18), so it is written from the reference's 2 bpp rules
        * (pvrtc_compressor.cc:111-349) with the block shape changed, and checked against oracle/ic_oracle.c's restatement of the
        * same rules and by decoding.  Accepted by icamd_encode_device (RGBA8, square power of two >= 8, no row padding:
        * size * size / 2 bytes per image, blocks in the 2 bpp Z order) and icamd_encoded_size only; under stream capture it needs a
        * caller workspace of icamd_pvrtc4_workspace_size bytes (icamd_pvrtc2_set_workspace). */
       ICAMD_PVRTC4 = 4 };

/* Status codes.  0 = the reference's `true`; 1 = the reference's `false` (argument
 * validation, unsupported format, external-storage size mismatch); < 0 = the device
 * path could not run -- callers must treat that as a hard error. */
enum { ICAMD_OK = 0, ICAMD_FALSE = 1,
       ICAMD_ERR_NO_DEVICE = -1, ICAMD_ERR_HIP = -2, ICAMD_ERR_ALLOC = -3, ICAMD_ERR_ARG = -4 };

/* ---- size / capability queries (host only, no device needed) ---- */

/* Compressor::ComputeCompressedDataSize -- compressor.h:68-69; semantics of
 * dxtc_compressor.cc:725-733, etc_compressor.cc:734-745, pvrtc_compressor.cc:631-634. */
size_t icamd_compute_compressed_data_size(int compressor, int format, uint32_t height, uint32_t width);

/* Compressor::SupportsFormat -- compressor.h:54; dxtc.cc:700-703 (all four),
 * etc.cc:713-717 (kRGB only), pvrtc.cc:607-609 (kRGBA only). */
int icamd_supports_format(int compressor, int format);

/* Bytes written by icamd_encode_device for one image of that block grid. */
size_t icamd_encoded_size(int codec, uint32_t grid_height, uint32_t grid_width);

/* ---- the hot path, host buffers: exact drop-in for Compressor::Compress ----
 * compressor.h:77-80 (note (height, width) order).  `buffer` is `height` rows of
 * width*components + padding_bytes_per_row bytes of host memory; `out` is caller storage
 * of exactly out_size == icamd_compute_compressed_data_size(...) bytes (the reference's
 * external-storage contract, internal/compressor4x4_helper.cc:34-41).
 * Returns when `out` is filled.  Inside: DXT / ETC images are cut into bands of whole block rows whose H2D copy,
 * kernel and D2H copy are pipelined over two internal per-thread streams (bands are independent images: blocks are
 * row-major, compressor4x4_helper.h:202-214); PVRTC is staged whole.  Pageable caller buffers work as they are;
 * buffers page-locked with icamd_host_register (or hipHostMalloc) are DMA-ed at the PCIe rate. */
int icamd_compress(int compressor, int etc_strategy, int format,
                   uint32_t height, uint32_t width, uint32_t padding_bytes_per_row,
                   const uint8_t *buffer, uint8_t *out, size_t out_size);

/* Optional: page-lock a caller buffer that will be passed to icamd_compress / icamd_compress_and_pad repeatedly
 * (input images, output storage), so that the copies are true asynchronous DMA.  Thin wrappers over hipHostRegister /
 * hipHostUnregister; the caller unregisters before freeing the memory. */
int icamd_host_register(void *host_ptr, size_t bytes);
int icamd_host_unregister(void *host_ptr);

/* Compressor::CompressAndPad -- compressor.h:114-119; helper.h:479-520.
 * PVRTC returns ICAMD_FALSE like pvrtc_compressor.cc:684-691. */
int icamd_compress_and_pad(int compressor, int etc_strategy, int format,
                           uint32_t height, uint32_t width,
                           uint32_t padded_height, uint32_t padded_width,
                           uint32_t padding_bytes_per_row,
                           const uint8_t *buffer, uint8_t *out, size_t out_size);

/* Multi-GPU sharding of ONE PVRTC texture: encodes only the blocks whose Z-order index
 * (pvrtc_compressor.cc:80-86, :551-580) lies in [first_block, first_block + n_blocks) -- n_blocks a power of two,
 * first_block a multiple of it, i.e. a rectangle of the block grid and ONE contiguous 8*n_blocks-byte range of the
 * texture's output.  d_src is the whole size x size RGBA8 image (device); only the region's pixels and a one-block
 * toroidal ring around it are read (the neighbours' colours are recomputed locally: no exchange between ranks).
 * d_dst_region receives 8*n_blocks bytes.  ICAMD_FALSE where PvrtcCompressor::Compress would refuse the size. */
int icamd_pvrtc2_encode_region_device(uint32_t size, uint32_t first_block, uint32_t n_blocks, const void *d_src,
                                      void *d_dst_region, void *hip_stream);

/* PVRTC scratch memory.  The PVRTC encoder keeps 8 bytes per block (the reference's two low-resolution colour images,
 * pvrtc_compressor.cc:586-597) between its two kernels.  By default that lives in a per-thread, grow-only buffer the
 * library allocates, which cannot be used while the stream is being CAPTURED into a HIP graph (the graph would keep a
 * pointer that a later, larger call frees; a PVRTC call under capture then returns ICAMD_ERR_HIP).  To capture -- or to
 * control the memory -- hand the library a buffer of at least icamd_pvrtc2_workspace_size(size, n_images) bytes for the
 * calls that follow on THIS thread; the caller keeps it alive and exclusive for as long as work (or a graph) using it
 * can run, one per graph.  NULL returns to the internal buffer. */
size_t icamd_pvrtc2_workspace_size(uint32_t size, uint32_t n_images);
size_t icamd_pvrtc4_workspace_size(uint32_t size, uint32_t n_images); /* the same for ICAMD_PVRTC4 (extension) */
int icamd_pvrtc2_set_workspace(void *d_workspace, size_t bytes);

/* PVRTC kernel selection (EXTENSION, tuning / test hook; results are identical either way).  Whole textures of 512^2 ...
 * 4096^2 in launches large enough to fill the chip take the one-pass kernel (Morph, Modulate and Encode of
 * pvrtc_compressor.cc:586-597 in ONE read of the pixels, no scratch memory -- such launches can be captured into a HIP graph
 * without a caller-owned workspace).  Textures of 8192^2 and more, and regions (icamd_pvrtc2_encode_region_device) at least 64
 * block columns wide, take the same kernel in its HALO form where the time model prefers it (r06: a block row split over several
 * workgroups, each reducing the block columns beyond its edges itself -- still one launch and no scratch memory).  Everything
 * else takes the morph + encode pair.  mode 0 = automatic (default; also the
 * value of the environment variable ICAMD_PVRTC2_PATH=auto|two|one read at the first launch), 1 = always the pair, 2 = one pass
 * wherever eligible; log2_strip < 0 = automatic strip height (blocks per lane) of the one-pass kernel (2 ... 6), else that height, clamped to
 * 2 ... log2(size / 4) (the tallest strip is the whole texture: one workgroup per texture).
 * Process-wide.  Returns ICAMD_OK, or ICAMD_ERR_ARG for a mode outside 0 ... 2. */
int icamd_pvrtc2_tune(int mode, int log2_strip);

/* ---- the hot path, device-resident (the roofline entry points) ----
 * Same contracts, but `d_buffer` / `d_out` are device pointers on the current HIP
 * device and the work is enqueued on `hip_stream` (a hipStream_t, NULL = default
 * stream) without synchronising. */
int icamd_compress_device(int compressor, int etc_strategy, int format,
                          uint32_t height, uint32_t width, uint32_t padding_bytes_per_row,
                          const void *d_buffer, void *d_out, size_t out_size, void *hip_stream);

int icamd_compress_and_pad_device(int compressor, int etc_strategy, int format,
                                  uint32_t height, uint32_t width,
                                  uint32_t padded_height, uint32_t padded_width,
                                  uint32_t padding_bytes_per_row,
                                  const void *d_buffer, void *d_out, size_t out_size, void *hip_stream);

/* Generic block-grid encoder over a batch of equally-shaped images (one launch).
 *   codec            ICAMD_DXT1 / DXT5 / ETC1 / PVRTC2
 *   src_components   3 or 4 bytes per source pixel.  4 with DXT1/ETC1 is the
 *                    "RGBA8, alpha ignored" extension named by BASELINE.json; its result
 *                    is defined as the reference's output for the alpha-stripped image.
 *   swap_rb          source is B,G,R(,A) (NeedsRedAndBlueSwapped, compressed_image.h:202-204)
 *   grid_height/width  >= height/width: block grid to emit (CompressAndPad); pass the
 *                    image dims for plain Compress.
 *   row_stride_bytes distance between source rows; *_image_stride_bytes between images.
 * Image i is read at d_src + i*src_image_stride_bytes and its blocks written at
 * d_dst + i*dst_image_stride_bytes (row-major blocks; PVRTC: Z-order, pvrtc.cc:551-580).
 * Any uint32 geometry runs (grids, batches and strides beyond one launch's limits are chunked internally).
 * Alignment: DXT / ETC accept any pointers and strides; PVRTC reads 16 bytes at a time and requires d_src (and
 * src_image_stride_bytes) 16-byte aligned, d_dst (and dst_image_stride_bytes) 8-byte aligned, else ICAMD_ERR_ARG. */
int icamd_encode_device(int codec, int etc_strategy, int src_components, int swap_rb,
                        uint32_t height, uint32_t width, uint32_t grid_height, uint32_t grid_width,
                        uint32_t row_stride_bytes, uint32_t n_images,
                        size_t src_image_stride_bytes, size_t dst_image_stride_bytes,
                        const void *d_src, void *d_dst, void *hip_stream);

/* ---- "next" row 8f.1: block decoders on device (Compressor::Decompress, compressor.h:85-86;
 * helper.h:218-262, dxtc.cc:167-267, etc.cc:198-289).  Writes height rows of
 * width*comps + padding_bytes_per_row bytes (comps = 4 for DXT5, PVRTC2 and PVRTC4, else 3).
 * codec ICAMD_PVRTC2 is an EXTENSION with PARITY UNPINNED: the reference has no PVRTC decoder
 * (PvrtcCompressor::Decompress returns false, pvrtc_compressor.cc:669-672, and so does icamd_decompress); this one is
 * written from the encoder's own rules (up-sampling pvrtc.cc:173-237, modulation :111-135, block layout :356-496,
 * Z order :80-86) and needs square power-of-two sizes and padding_bytes_per_row == 0.  codec ICAMD_PVRTC4 (r05) is the
 * decoder of the 4 bpp extension encoder, under the same conditions and as unpinned as that: 4 x 4 blocks, every pixel its
 * own 2-bit value (weights 0, 3, 5, 8; a block with colour-word bit 0 set -- the encoder never writes one -- takes PVRTC1's
 * punch-through weights 0, 4, 4, 8 with alpha 0 for value 2). */
int icamd_decode_device(int codec, int swap_rb, uint32_t height, uint32_t width,
                        uint32_t padding_bytes_per_row, uint32_t n_images,
                        size_t src_image_stride_bytes, size_t dst_image_stride_bytes,
                        const void *d_blocks, void *d_pixels, void *hip_stream);

/* Host-buffer drop-in for Compressor::Decompress (compressor.h:85-86) on DXTC / ETC images: `blocks` holds
 * the block grid of an image whose metadata says (uncompressed_height, uncompressed_width,
 * padding_bytes_per_row); `out` receives height rows of width*comps + padding bytes (out_size must be
 * exactly that).  PVRTC returns ICAMD_FALSE (pvrtc_compressor.cc:669-672). */
int icamd_decompress(int compressor, int format, uint32_t height, uint32_t width, uint32_t padding_bytes_per_row,
                     const uint8_t *blocks, size_t blocks_size, uint8_t *out, size_t out_size);

/* EXTENSION (parity unpinned like icamd_decode_device(ICAMD_PVRTC2)): host-buffer PVRTC 2bpp decode of a size x size
 * texture into size*size*4 RGBA bytes.  icamd_decompress itself keeps answering ICAMD_FALSE for PVRTC, like
 * PvrtcCompressor::Decompress (pvrtc_compressor.cc:669-672); the C++ class of this repo only routes here when the
 * environment variable ICAMD_PVRTC_DECOMPRESS_EXTENSION=1 is set. */
int icamd_pvrtc2_decompress(uint32_t size, const uint8_t *blocks, size_t blocks_size, uint8_t *out, size_t out_size);

/* ---- "next" rows 8f.2-4: compressed-domain operations on one image's block grid ----
 * Compressor::Pad (compressor.h:104-106; helper.h:393-477; pad functors dxtc.cc:594-696, etc.cc:645-698) for the
 * case that really pads: the source grid covers (compressed_height, compressed_width) pixels, the result
 * (padded_height, padded_width); out_size must be the result's data size.  Device pointers of the block-domain
 * operations (pad, downsample: 4-byte; transcode: 8-byte) must be aligned, else ICAMD_ERR_ARG; the decoders accept
 * any pointer.  Returns ICAMD_FALSE for PVRTC and when a
 * padded dimension has fewer blocks than the source (the reference either just duplicates the image, which the caller
 * does itself, or overruns its buffer). */
int icamd_pad_device(int compressor, int etc_strategy, int format, uint32_t compressed_height, uint32_t compressed_width,
                     const void *d_blocks, uint32_t padded_height, uint32_t padded_width, void *d_out, size_t out_size,
                     void *hip_stream);
/* Extension (r05), like icamd_downsample_batch_device: n_images equally shaped grids padded in ONE launch (ETC1 with
 * kSplit* / kHeuristic: two -- the copy, then every image's border blocks together; kSmallerError: one, its pad blocks take
 * four lanes each in the first workgroups of the launch).  Image i at d_blocks + i * src_image_stride_bytes -> d_out + i * dst_image_stride_bytes (multiples of 4). */
int icamd_pad_batch_device(int compressor, int etc_strategy, int format, uint32_t compressed_height, uint32_t compressed_width,
                           uint32_t n_images, const void *d_blocks, size_t src_image_stride_bytes, uint32_t padded_height,
                           uint32_t padded_width, void *d_out, size_t dst_image_stride_bytes, size_t out_size_per_image,
                           void *hip_stream);
int icamd_pad(int compressor, int etc_strategy, int format, uint32_t compressed_height, uint32_t compressed_width,
              const uint8_t *blocks, uint32_t padded_height, uint32_t padded_width, uint8_t *out, size_t out_size);

/* Compressor::Downsample (compressor.h:95-96; helper.h:264-391,594-636): (uncompressed_height, uncompressed_width)
 * -> ((h+1)/2, (w+1)/2); decode, 2x2 average, re-encode per output block.  ICAMD_FALSE where the reference refuses
 * (odd block counts > 1, single block with a 3-pixel side, PVRTC). */
int icamd_downsample_device(int compressor, int etc_strategy, int format, uint32_t uncompressed_height,
                            uint32_t uncompressed_width, const void *d_blocks, void *d_out, size_t out_size,
                            void *hip_stream);
/* Extension: the same on n_images equally shaped block grids in ONE launch (image i at d_blocks + i * src_image_stride_bytes ->
 * d_out + i * dst_image_stride_bytes; strides multiples of 4).  One 4096^2 DXT1 level is 10 MB of traffic -- less than the
 * fixed cost of a launch is worth -- so a mip generator that halves many textures feeds them as a batch. */
int icamd_downsample_batch_device(int compressor, int etc_strategy, int format, uint32_t uncompressed_height,
                                  uint32_t uncompressed_width, uint32_t n_images, const void *d_blocks,
                                  size_t src_image_stride_bytes, void *d_out, size_t dst_image_stride_bytes,
                                  size_t out_size_per_image, void *hip_stream);
int icamd_downsample(int compressor, int etc_strategy, int format, uint32_t uncompressed_height,
                     uint32_t uncompressed_width, const uint8_t *blocks, uint8_t *out, size_t out_size);

/* Compressor::CreateSolidImage (compressor.h:124-127; helper.h:522-543; solid blocks dxtc_compressor.cc:42-49,77-82,
 * 820-839, etc_compressor.cc:595-617,802-812): the block grid of a height x width image of one colour.  `color` is a
 * HOST pointer to 3 (kRGB/kBGR) or 4 (kRGBA/kBGRA) bytes; out_size must be blocks * block size.  ICAMD_FALSE for PVRTC
 * (pvrtc_compressor.cc:693-698), for ETC formats other than kRGB, and on a size mismatch.  The _device form fills a
 * device-resident grid (one kernel, streaming stores); the host form replicates the block on the host like the reference
 * (byte shuffling: nothing to offload). */
int icamd_create_solid_device(int compressor, int format, uint32_t height, uint32_t width, const uint8_t *color,
                              void *d_out, size_t out_size, void *hip_stream);
/* Extension (r05): n_images grids in ONE launch, image i of colour colors[i * components .. ] (HOST pointer, n_images x 3 or 4
 * bytes) at d_out + i * dst_image_stride_bytes.  A 4096^2 DXT1 grid is 8 MiB: one fill per call is launch-bound. */
int icamd_create_solid_batch_device(int compressor, int format, uint32_t height, uint32_t width, uint32_t n_images,
                                    const uint8_t *colors, void *d_out, size_t dst_image_stride_bytes, size_t out_size_per_image,
                                    void *hip_stream);
int icamd_create_solid(int compressor, int format, uint32_t height, uint32_t width, const uint8_t *color, uint8_t *out,
                       size_t out_size);

/* Compressor::CopySubimage (compressor.h:133-136; helper.h:545-592): the blocks of the height x width window at
 * (start_row, start_column) of an image whose grid covers (compressed_height, compressed_width) pixels.  ICAMD_FALSE
 * unless all four are multiples of 4 and the window lies inside the compressed image (helper.h:555-563), for PVRTC, and
 * on a size mismatch. */
int icamd_copy_subimage_device(int compressor, int format, uint32_t compressed_height, uint32_t compressed_width,
                               const void *d_blocks, uint32_t start_row, uint32_t start_column, uint32_t height,
                               uint32_t width, void *d_out, size_t out_size, void *hip_stream);
/* Extension (r05): the same window of n_images equally shaped grids in ONE launch (strides multiples of 4). */
int icamd_copy_subimage_batch_device(int compressor, int format, uint32_t compressed_height, uint32_t compressed_width,
                                     uint32_t n_images, const void *d_blocks, size_t src_image_stride_bytes, uint32_t start_row,
                                     uint32_t start_column, uint32_t height, uint32_t width, void *d_out,
                                     size_t dst_image_stride_bytes, size_t out_size_per_image, void *hip_stream);
int icamd_copy_subimage(int compressor, int format, uint32_t compressed_height, uint32_t compressed_width,
                        const uint8_t *blocks, uint32_t start_row, uint32_t start_column, uint32_t height,
                        uint32_t width, uint8_t *out, size_t out_size);

/* TranscodeDxt1ToEtc1 (public/dxtc_to_etc_transcoder.h:24; dxtc_to_etc_transcoder.cc:29-40): in place. */
int icamd_transcode_dxt1_to_etc1_device(void *d_blocks, size_t n_bytes, void *hip_stream);
int icamd_transcode_dxt1_to_etc1(uint8_t *blocks, size_t n_bytes);

/* ---- multi-GPU from one process: a batch of independent images, host buffers ----
 * Image i is compressed exactly like icamd_compress(compressor, ..., buffers[i], outs[i], out_size) on device
 * devices[i % n_devices] (HIP device ordinals; a device may be listed more than once to get several in-flight
 * streams on it).  One host worker thread per list entry drives its own stream and staging buffers, so copies and
 * kernels of different devices overlap; images are independent, so there is no inter-device exchange and the
 * results land directly in the caller's host buffers.  statuses[i] (optional) receives each image's status; the
 * return value is ICAMD_OK if all are ICAMD_OK, otherwise the first non-OK status in image order. */
int icamd_compress_batch(int compressor, int etc_strategy, int format, uint32_t height, uint32_t width,
                         uint32_t padding_bytes_per_row, uint32_t n_images, const uint8_t *const *buffers,
                         uint8_t *const *outs, size_t out_size, const int *devices, int n_devices, int *statuses);

/* The same for images that are ALREADY RESIDENT IN HBM: image i lives on device
 * devices[i % n_devices] (d_srcs[i] is a pointer on that device) and is encoded there exactly like
 * icamd_encode_device(codec, ..., height, width, height, width, row_stride_bytes, 1 image) -- Compressor::Compress
 * (compressor.h:77-80) per image, no pixel ever crosses PCIe.  One host worker thread per list entry drives two HIP
 * streams on its device.  Output:
 *   d_dsts[i] != NULL      the image's blocks are written there (a pointer on the image's own device);
 *   gather_device >= 0     every image's blocks additionally land at d_gathered + i * gathered_image_stride_bytes, a
 *                          buffer on gather_device ("rank 0"): device-to-device copies (hipMemcpyPeerAsync, xGMI between
 *                          GPUs) that overlap the next image's encode; images on gather_device itself are encoded
 *                          straight into their slot when they have no d_dsts entry.  d_dsts may be NULL altogether.
 * Returns after all streams have drained.  statuses / return value as icamd_compress_batch. */
int icamd_encode_batch_sharded_device(int codec, int etc_strategy, int src_components, int swap_rb, uint32_t height,
                                      uint32_t width, uint32_t row_stride_bytes, uint32_t n_images,
                                      const void *const *d_srcs, void *const *d_dsts, const int *devices, int n_devices,
                                      int gather_device, void *d_gathered, size_t gathered_image_stride_bytes,
                                      int *statuses);

/* ---- multi-GPU, ONE PROCESS PER GPU: the gather of the compressed output over RCCL ("Collective" column) ----
 * BASELINE.json's north star: "large texture batches shard across the 8 GPUs of one node as independent image slabs ...
 * RCCL over xGMI only to gather the compressed output".  Encoding needs no exchange (icamd_encode_device on every rank's own
 * textures, or its block-row slab / PVRTC region of one image: helper.h:202-214 makes a slab's blocks ONE contiguous byte
 * range of the final buffer); what a C / C++ caller of the drop-in classes that runs one process per GPU still needs is the
 * collective that brings the ranks' byte ranges together on one rank -- this entry point.  The reference has nothing of the
 * kind (it is a single-threaded CPU library, public/compressor.h:48-138): there is no interface to cite, only the layout rule.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1 -- the copy already loaded into the process if there is one, e.g.
 * PyTorch's), so libic_amd.so itself does not depend on it and every other entry point works without it.
 *
 * icamd_rccl_available       1 if librccl could be bound, else 0 (and icamd_last_error() says why).
 * icamd_rccl_get_unique_id   ncclGetUniqueId into id[ICAMD_RCCL_UNIQUE_ID_BYTES]; call on ONE rank and hand the bytes to the
 *                            others by any means (MPI_Bcast, a file, a socket, torch.distributed's store).
 * icamd_rccl_comm_init       ncclCommInitRank on the CURRENT HIP device; *comm is an ncclComm_t.  Collective: every rank
 *                            calls it with the same id and world.  A communicator the caller made with its own RCCL calls
 *                            (same librccl) is equally accepted by icamd_gather_blocks_rccl.
 * icamd_rccl_comm_destroy    ncclCommDestroy.
 * icamd_gather_blocks_rccl   rank r contributes counts_bytes[r] bytes at d_local (device memory of its own GPU); on rank
 *                            `root` they land at d_root_buffer + root_offsets_bytes[r] (NULL = the prefix sums of
 *                            counts_bytes: the ranks' ranges back to back, which is the final block stream when the ranks
 *                            hold consecutive slabs / texture ranges).  Unequal and zero counts are fine.  ONE grouped
 *                            ncclSend / ncclRecv exchange (ncclGroupStart ... ncclGroupEnd, ncclUint8): every peer writes
 *                            its own range of root's HBM over its own xGMI link; root's own range is a device-to-device
 *                            copy on the same stream (skipped when d_local already is its slot).  Enqueued on hip_stream,
 *                            not synchronised: run it on a second stream underneath the next batch's encode.  d_root_buffer
 *                            and root_offsets_bytes are only read on root.  Every rank passes the same counts.
 * Status: ICAMD_OK; ICAMD_ERR_ARG (bad rank / world / root, null pointers where bytes are due); ICAMD_ERR_NO_DEVICE when
 * librccl cannot be bound; ICAMD_ERR_HIP with RCCL's own error text for a failing RCCL or HIP call. */
#define ICAMD_RCCL_UNIQUE_ID_BYTES 128
int icamd_rccl_available(void);
int icamd_rccl_get_unique_id(void *id);
int icamd_rccl_comm_init(void **comm, int world, int rank, const void *id);
int icamd_rccl_comm_destroy(void *comm);
int icamd_gather_blocks_rccl(void *comm, int rank, int world, int root, const size_t *counts_bytes, const void *d_local,
                             void *d_root_buffer, const size_t *root_offsets_bytes, void *hip_stream);

/* ---- container framing (EXTENSION: SURVEY.md 8(f) row 4, tail) ----
 * The reference ends at the raw block stream (compressed_image.h:52-66); it has no file-container code, so there is
 * nothing to pin these against.  Host-side byte framing only (no device work), layouts from the public format
 * descriptions (csrc/containers.h): DDS (DXT1 / DXT5), KTX 1.1 and PVR v3 (all four codecs), PKM (ETC1, one level).
 * Level l of a height x width texture is max(1, height >> l) x max(1, width >> l) pixels, its bytes exactly what
 * icamd_compress / icamd_downsample return for that size (PVRTC: square power-of-two levels of 8 x 8 and up only). */
enum { ICAMD_CONTAINER_DDS = 0, ICAMD_CONTAINER_KTX = 1, ICAMD_CONTAINER_PKM = 2, ICAMD_CONTAINER_PVR = 3 };
/* File size for `levels` mip levels (>= 1); 0 if the container cannot hold that codec / size / level count. */
size_t icamd_container_size(int container, int codec, uint32_t height, uint32_t width, uint32_t levels);
/* Writes header + levels (largest first) into out[out_size]; out_size must equal icamd_container_size(...) and
 * level_sizes[l] the level's block-stream size, else ICAMD_FALSE.  ICAMD_ERR_ARG for an unknown container / codec. */
int icamd_container_write(int container, int codec, uint32_t height, uint32_t width, uint32_t levels,
                          const uint8_t *const *level_data, const size_t *level_sizes, uint8_t *out, size_t out_size);

/* ---- runtime ---- */
int icamd_device_count(void);             /* HIP devices visible; 0 if none */
const char *icamd_last_error(void);       /* thread-local message for the last negative status */
const char *icamd_version(void);
/* Name of the __global__ kernel a given configuration launches (for matching rocprofv3 rows). */
const char *icamd_kernel_name(int codec, int src_components);

/* ---- diagnostics (measurement aid; not part of the encode path) ----
 * Effective shader clock under load: enqueues ONE wave on hip_stream that sleeps for duration_us and then writes
 * {shader cycles elapsed (s_memtime), constant-rate ticks elapsed (s_memrealtime)} as two uint64 to d_out16.  Run it
 * on a stream of its own next to the kernels being timed; mean clock = cycles / ticks * icamd_wall_clock_rate_khz().
 * duration_us above ICAMD_CLOCK_PROBE_MAX_US is refused with ICAMD_ERR_ARG. */
#define ICAMD_CLOCK_PROBE_MAX_US 10000000u
int icamd_clock_probe_device(void *d_out16, uint32_t duration_us, void *hip_stream);
uint32_t icamd_wall_clock_rate_khz(void);  /* hipDeviceAttributeWallClockRate of the current device; 0 if unknown */

#ifdef __cplusplus
}
#endif
#endif /* IC_AMD_H_ */
