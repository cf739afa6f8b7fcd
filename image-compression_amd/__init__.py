"""image-compression_amd: MI355X (gfx950) block-encode backend -- Python plumbing over the C ABI.

This module only *binds* libic_amd.so (include/ic_amd.h) with ctypes and passes torch device
pointers / streams through it.  All encoding happens in the hand-written HIP kernels inside the
shared library; there is no Python or CPU implementation here, and importing fails loudly if the
library has not been built (python __graft_entry__.py / make -C image-compression_amd).

The directory name contains '-', so import it via `ic_amd_loader.load_package()` (repo root) or
importlib; the package registers itself as `image_compression_amd`.
"""
import ctypes
import os
import threading

import torch  # must precede CDLL: libic_amd.so then binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB_PATH = os.path.join(_HERE, "libic_amd.so")
# A/B experiments only: ICAMD_LIB_PATH swaps the product library for another build, and is honoured ONLY together with
# ICAMD_ALLOW_LIB_OVERRIDE=1 (a stray variable must not silently redirect a benchmark); bench.py records LIB_PATH and
# LIB_OVERRIDDEN in its output line.
if "ICAMD_LIB_PATH" in os.environ and os.environ.get("ICAMD_ALLOW_LIB_OVERRIDE") != "1":
    raise ImportError("ICAMD_LIB_PATH is set (%s) but ICAMD_ALLOW_LIB_OVERRIDE=1 is not: refusing to load a library "
                      "other than %s" % (os.environ["ICAMD_LIB_PATH"], _DEFAULT_LIB_PATH))
LIB_PATH = os.environ.get("ICAMD_LIB_PATH", _DEFAULT_LIB_PATH)
LIB_OVERRIDDEN = os.path.realpath(LIB_PATH) != os.path.realpath(_DEFAULT_LIB_PATH)

# enums of include/ic_amd.h
COMPRESSOR_DXTC, COMPRESSOR_ETC, COMPRESSOR_PVRTC = 0, 1, 2
RGB, BGR, RGBA, BGRA = 0, 1, 2, 3
ETC_SPLIT_HORIZONTALLY, ETC_SPLIT_VERTICALLY, ETC_SMALLER_ERROR, ETC_HEURISTIC = 0, 1, 2, 3
DXT1, DXT5, ETC1, PVRTC2 = 0, 1, 2, 3
PVRTC4 = 4  # EXTENSION, parity unpinned: PVRTC1 4 bpp (include/ic_amd.h); icamd_encode_device only
OK, FALSE = 0, 1

EXPORTS = [
    "icamd_compute_compressed_data_size", "icamd_supports_format", "icamd_encoded_size", "icamd_compress",
    "icamd_compress_and_pad", "icamd_compress_device", "icamd_compress_and_pad_device", "icamd_encode_device",
    "icamd_decode_device", "icamd_decompress", "icamd_pad_device", "icamd_pad", "icamd_downsample_device",
    "icamd_downsample", "icamd_downsample_batch_device", "icamd_pad_batch_device", "icamd_create_solid_batch_device", "icamd_copy_subimage_batch_device", "icamd_transcode_dxt1_to_etc1_device", "icamd_transcode_dxt1_to_etc1", "icamd_compress_batch", "icamd_pvrtc2_encode_region_device", "icamd_pvrtc2_workspace_size", "icamd_pvrtc4_workspace_size",
    "icamd_pvrtc2_set_workspace", "icamd_pvrtc2_tune", "icamd_host_register", "icamd_host_unregister", "icamd_pvrtc2_decompress", "icamd_device_count", "icamd_last_error", "icamd_version", "icamd_kernel_name",
    "icamd_create_solid_device", "icamd_create_solid", "icamd_copy_subimage_device", "icamd_copy_subimage",
    "icamd_encode_batch_sharded_device", "icamd_clock_probe_device", "icamd_wall_clock_rate_khz",
    "icamd_container_size", "icamd_container_write",
    "icamd_rccl_available", "icamd_rccl_get_unique_id", "icamd_rccl_comm_init", "icamd_rccl_comm_destroy", "icamd_gather_blocks_rccl",
]
RCCL_UNIQUE_ID_BYTES = 128
CONTAINER_DDS, CONTAINER_KTX, CONTAINER_PKM, CONTAINER_PVR = 0, 1, 2, 3

_u32, _sz, _vp, _ci = ctypes.c_uint32, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
_lib = None
_tls = threading.local()  # per-thread state mirrored from the C side (the PVRTC workspace override is thread-local there)


class BackendError(RuntimeError):
    """The device path could not run (negative ICAMD_ERR_* status)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libic_amd.so is not built (%s); run `python __graft_entry__.py` or "
                              "`make -C image-compression_amd` -- there is no fallback path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.icamd_compute_compressed_data_size.restype = _sz
        L.icamd_compute_compressed_data_size.argtypes = [_ci, _ci, _u32, _u32]
        L.icamd_supports_format.restype = _ci
        L.icamd_supports_format.argtypes = [_ci, _ci]
        L.icamd_encoded_size.restype = _sz
        L.icamd_encoded_size.argtypes = [_ci, _u32, _u32]
        L.icamd_compress.restype = _ci
        L.icamd_compress.argtypes = [_ci, _ci, _ci, _u32, _u32, _u32, _vp, _vp, _sz]
        L.icamd_compress_and_pad.restype = _ci
        L.icamd_compress_and_pad.argtypes = [_ci, _ci, _ci, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _sz]
        L.icamd_compress_device.restype = _ci
        L.icamd_compress_device.argtypes = [_ci, _ci, _ci, _u32, _u32, _u32, _vp, _vp, _sz, _vp]
        L.icamd_compress_and_pad_device.restype = _ci
        L.icamd_compress_and_pad_device.argtypes = [_ci, _ci, _ci, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _sz, _vp]
        L.icamd_container_size.restype = _sz
        L.icamd_container_size.argtypes = [_ci, _ci, _u32, _u32, _u32]
        L.icamd_container_write.restype = _ci
        L.icamd_container_write.argtypes = [_ci, _ci, _u32, _u32, _u32, _vp, _vp, _vp, _sz]
        L.icamd_encode_device.restype = _ci
        L.icamd_encode_device.argtypes = [_ci, _ci, _ci, _ci, _u32, _u32, _u32, _u32, _u32, _u32, _sz, _sz, _vp, _vp, _vp]
        L.icamd_decode_device.restype = _ci
        L.icamd_decode_device.argtypes = [_ci, _ci, _u32, _u32, _u32, _u32, _sz, _sz, _vp, _vp, _vp]
        L.icamd_decompress.restype = _ci
        L.icamd_decompress.argtypes = [_ci, _ci, _u32, _u32, _u32, _vp, _sz, _vp, _sz]
        L.icamd_pad.restype = _ci
        L.icamd_pad.argtypes = [_ci, _ci, _ci, _u32, _u32, _vp, _u32, _u32, _vp, _sz]
        L.icamd_pad_device.restype = _ci
        L.icamd_pad_device.argtypes = [_ci, _ci, _ci, _u32, _u32, _vp, _u32, _u32, _vp, _sz, _vp]
        L.icamd_downsample.restype = _ci
        L.icamd_downsample.argtypes = [_ci, _ci, _ci, _u32, _u32, _vp, _vp, _sz]
        L.icamd_downsample_device.restype = _ci
        L.icamd_downsample_device.argtypes = [_ci, _ci, _ci, _u32, _u32, _vp, _vp, _sz, _vp]
        if not LIB_OVERRIDDEN or hasattr(L, "icamd_downsample_batch_device"):  # r04 entry point
            L.icamd_downsample_batch_device.restype = _ci
            L.icamd_downsample_batch_device.argtypes = [_ci, _ci, _ci, _u32, _u32, _u32, _vp, _sz, _vp, _sz, _sz, _vp]
        L.icamd_transcode_dxt1_to_etc1.restype = _ci
        L.icamd_transcode_dxt1_to_etc1.argtypes = [_vp, _sz]
        L.icamd_transcode_dxt1_to_etc1_device.restype = _ci
        L.icamd_transcode_dxt1_to_etc1_device.argtypes = [_vp, _sz, _vp]
        L.icamd_compress_batch.restype = _ci
        L.icamd_compress_batch.argtypes = [_ci, _ci, _ci, _u32, _u32, _u32, _u32, _vp, _vp, _sz, _vp, _ci, _vp]
        L.icamd_pvrtc2_encode_region_device.restype = _ci
        L.icamd_pvrtc2_encode_region_device.argtypes = [_u32, _u32, _u32, _vp, _vp, _vp]
        if not LIB_OVERRIDDEN or hasattr(L, "icamd_pvrtc2_workspace_size"):  # older A/B builds lack it
            L.icamd_pvrtc2_workspace_size.restype = _sz
            L.icamd_pvrtc2_workspace_size.argtypes = [_u32, _u32]
            if hasattr(L, "icamd_pvrtc4_workspace_size"):
                L.icamd_pvrtc4_workspace_size.restype = _sz
                L.icamd_pvrtc4_workspace_size.argtypes = [_u32, _u32]
            L.icamd_pvrtc2_set_workspace.restype = _ci
            L.icamd_pvrtc2_set_workspace.argtypes = [_vp, _sz]
            L.icamd_host_register.restype = _ci
            L.icamd_host_register.argtypes = [_vp, _sz]
            L.icamd_host_unregister.restype = _ci
            L.icamd_host_unregister.argtypes = [_vp]
            L.icamd_pvrtc2_decompress.restype = _ci
            L.icamd_pvrtc2_decompress.argtypes = [_u32, _vp, _sz, _vp, _sz]
        if not LIB_OVERRIDDEN or hasattr(L, "icamd_pvrtc2_tune"):  # r05 entry point (older A/B builds lack it)
            L.icamd_pvrtc2_tune.restype = _ci
            L.icamd_pvrtc2_tune.argtypes = [_ci, _ci]
        if not LIB_OVERRIDDEN or hasattr(L, "icamd_pad_batch_device"):  # r05 entry points
            L.icamd_pad_batch_device.restype = _ci
            L.icamd_pad_batch_device.argtypes = [_ci, _ci, _ci, _u32, _u32, _u32, _vp, _sz, _u32, _u32, _vp, _sz, _sz, _vp]
            L.icamd_create_solid_batch_device.restype = _ci
            L.icamd_create_solid_batch_device.argtypes = [_ci, _ci, _u32, _u32, _u32, _vp, _vp, _sz, _sz, _vp]
            L.icamd_copy_subimage_batch_device.restype = _ci
            L.icamd_copy_subimage_batch_device.argtypes = [_ci, _ci, _u32, _u32, _u32, _vp, _sz, _u32, _u32, _u32, _u32, _vp, _sz,
                                                           _sz, _vp]
        if not LIB_OVERRIDDEN or hasattr(L, "icamd_create_solid_device"):  # r03 entry points
            L.icamd_create_solid_device.restype = _ci
            L.icamd_create_solid_device.argtypes = [_ci, _ci, _u32, _u32, _vp, _vp, _sz, _vp]
            L.icamd_create_solid.restype = _ci
            L.icamd_create_solid.argtypes = [_ci, _ci, _u32, _u32, _vp, _vp, _sz]
            L.icamd_copy_subimage_device.restype = _ci
            L.icamd_copy_subimage_device.argtypes = [_ci, _ci, _u32, _u32, _vp, _u32, _u32, _u32, _u32, _vp, _sz, _vp]
            L.icamd_copy_subimage.restype = _ci
            L.icamd_copy_subimage.argtypes = [_ci, _ci, _u32, _u32, _vp, _u32, _u32, _u32, _u32, _vp, _sz]
            L.icamd_encode_batch_sharded_device.restype = _ci
            L.icamd_encode_batch_sharded_device.argtypes = [_ci, _ci, _ci, _ci, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _ci,
                                                            _ci, _vp, _sz, _vp]
            L.icamd_clock_probe_device.restype = _ci
            L.icamd_clock_probe_device.argtypes = [_vp, _u32, _vp]
            L.icamd_wall_clock_rate_khz.restype = _u32
        if not LIB_OVERRIDDEN or hasattr(L, "icamd_gather_blocks_rccl"):  # r06 entry points
            L.icamd_rccl_available.restype = _ci
            L.icamd_rccl_available.argtypes = []
            L.icamd_rccl_get_unique_id.restype = _ci
            L.icamd_rccl_get_unique_id.argtypes = [_vp]
            L.icamd_rccl_comm_init.restype = _ci
            L.icamd_rccl_comm_init.argtypes = [ctypes.POINTER(_vp), _ci, _ci, _vp]
            L.icamd_rccl_comm_destroy.restype = _ci
            L.icamd_rccl_comm_destroy.argtypes = [_vp]
            L.icamd_gather_blocks_rccl.restype = _ci
            L.icamd_gather_blocks_rccl.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]
        L.icamd_device_count.restype = _ci
        L.icamd_last_error.restype = ctypes.c_char_p
        L.icamd_version.restype = ctypes.c_char_p
        L.icamd_kernel_name.restype = ctypes.c_char_p
        L.icamd_kernel_name.argtypes = [_ci, _ci]
        _lib = L
    return _lib


def _check(status, what):
    if status < 0:
        raise BackendError("%s failed with status %d: %s" % (what, status, lib().icamd_last_error().decode()))
    return status == OK


def compute_compressed_data_size(compressor, fmt, height, width):
    return lib().icamd_compute_compressed_data_size(compressor, fmt, height, width)


def encoded_size(codec, grid_height, grid_width):
    return lib().icamd_encoded_size(codec, grid_height, grid_width)


def kernel_name(codec, src_components):
    return lib().icamd_kernel_name(codec, src_components).decode()


def _stream_handle(stream=None):
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


def encode_device(codec, src, height, width, src_components, *, swap_rb=False, etc_strategy=ETC_SMALLER_ERROR,
                  grid_height=None, grid_width=None, row_stride_bytes=None, n_images=1,
                  src_image_stride_bytes=None, out=None, stream=None):
    """Launch the encode kernel on `src` (a torch.uint8 CUDA tensor, any shape, contiguous bytes).
    Returns the output tensor [n_images, encoded_size] (device).  No synchronisation."""
    assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
    gh = height if grid_height is None else max(grid_height, height)
    gw = width if grid_width is None else max(grid_width, width)
    stride = width * src_components if row_stride_bytes is None else row_stride_bytes
    img_stride = height * stride if src_image_stride_bytes is None else src_image_stride_bytes
    per = encoded_size(codec, gh, gw)
    if out is None:
        out = torch.empty((n_images, per), dtype=torch.uint8, device=src.device)
    st = lib().icamd_encode_device(codec, etc_strategy, src_components, int(swap_rb), height, width, gh, gw, stride,
                                   n_images, img_stride, per, ctypes.c_void_p(src.data_ptr()),
                                   ctypes.c_void_p(out.data_ptr()), _stream_handle(stream))
    if not _check(st, "icamd_encode_device"):
        return None
    return out


def compress_device(compressor, fmt, src, height, width, *, padding_bytes_per_row=0,
                    etc_strategy=ETC_SMALLER_ERROR, padded=None, out_size=None, stream=None):
    """Compressor::Compress / CompressAndPad on a device-resident image; returns a device uint8 tensor or None
    where the reference returns false."""
    assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
    if padded is None:
        n = compute_compressed_data_size(compressor, fmt, height, width) if out_size is None else out_size
    else:
        n = compute_compressed_data_size(compressor, fmt, max(height, padded[0]), max(width, padded[1])) \
            if out_size is None else out_size
    out = torch.empty((max(n, 1),), dtype=torch.uint8, device=src.device)
    if padded is None:
        st = lib().icamd_compress_device(compressor, etc_strategy, fmt, height, width, padding_bytes_per_row,
                                         ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), n,
                                         _stream_handle(stream))
    else:
        st = lib().icamd_compress_and_pad_device(compressor, etc_strategy, fmt, height, width, padded[0], padded[1],
                                                 padding_bytes_per_row, ctypes.c_void_p(src.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr()), n, _stream_handle(stream))
    if not _check(st, "icamd_compress_device"):
        return None
    return out[:n]


def host_register(array):
    """Page-locks a numpy array's memory (icamd_host_register); pair with host_unregister before it is freed."""
    return _check(lib().icamd_host_register(ctypes.c_void_p(array.ctypes.data), array.nbytes), "icamd_host_register")


def host_unregister(array):
    return _check(lib().icamd_host_unregister(ctypes.c_void_p(array.ctypes.data)), "icamd_host_unregister")


def compress_host(compressor, fmt, buffer, height, width, *, padding_bytes_per_row=0,
                  etc_strategy=ETC_SMALLER_ERROR, padded=None, out_size=None, out=None):
    """The host-buffer drop-in (H2D + kernel + D2H inside the library).  `buffer`: bytes-like / numpy uint8.
    Returns bytes, or None where the reference returns false.  `out`: a caller-owned numpy uint8 array of the exact
    output size to write into (returned as is instead of bytes)."""
    import numpy as np
    src = np.ascontiguousarray(np.frombuffer(buffer, dtype=np.uint8) if not isinstance(buffer, np.ndarray) else buffer)
    if padded is None:
        n = compute_compressed_data_size(compressor, fmt, height, width) if out_size is None else out_size
    else:
        n = compute_compressed_data_size(compressor, fmt, max(height, padded[0]), max(width, padded[1])) \
            if out_size is None else out_size
    if out is not None:
        assert out.dtype == np.uint8 and out.size == n and out.flags["C_CONTIGUOUS"]
        if padded is None:
            st = lib().icamd_compress(compressor, etc_strategy, fmt, height, width, padding_bytes_per_row,
                                      src.ctypes.data, out.ctypes.data, n)
        else:
            st = lib().icamd_compress_and_pad(compressor, etc_strategy, fmt, height, width, padded[0], padded[1],
                                              padding_bytes_per_row, src.ctypes.data, out.ctypes.data, n)
        return out if _check(st, "icamd_compress") else None
    out = np.zeros(max(n, 1), np.uint8)
    if padded is None:
        st = lib().icamd_compress(compressor, etc_strategy, fmt, height, width, padding_bytes_per_row,
                                  src.ctypes.data, out.ctypes.data, n)
    else:
        st = lib().icamd_compress_and_pad(compressor, etc_strategy, fmt, height, width, padded[0], padded[1],
                                          padding_bytes_per_row, src.ctypes.data, out.ctypes.data, n)
    if not _check(st, "icamd_compress"):
        return None
    return out[:n].tobytes()


def decode_device(codec, blocks, height, width, *, swap_rb=False, padding_bytes_per_row=0, n_images=1, stream=None):
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
    comps = 4 if codec in (DXT5, PVRTC2, PVRTC4) else 3
    per_out = height * (width * comps + padding_bytes_per_row)
    per_in = encoded_size(codec, height, width)
    out = torch.zeros((n_images, per_out), dtype=torch.uint8, device=blocks.device)
    st = lib().icamd_decode_device(codec, int(swap_rb), height, width, padding_bytes_per_row, n_images, per_in,
                                   per_out, ctypes.c_void_p(blocks.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                   _stream_handle(stream))
    if not _check(st, "icamd_decode_device"):
        return None
    return out


def _block_bytes(compressor, fmt):
    return 8 if (compressor == COMPRESSOR_ETC or fmt in (RGB, BGR)) else 16


def pad_host(compressor, fmt, blocks, compressed_height, compressed_width, padded_height, padded_width,
             etc_strategy=ETC_SMALLER_ERROR):
    """Compressor::Pad for the really-padding case, host buffers.  bytes or None (reference's false)."""
    import numpy as np
    b = np.frombuffer(blocks, np.uint8)
    n = ((padded_height + 3) // 4) * ((padded_width + 3) // 4) * _block_bytes(compressor, fmt)
    out = np.zeros(max(n, 1), np.uint8)
    st = lib().icamd_pad(compressor, etc_strategy, fmt, compressed_height, compressed_width, b.ctypes.data,
                         padded_height, padded_width, out.ctypes.data, n)
    return out[:n].tobytes() if _check(st, "icamd_pad") else None


def downsample_host(compressor, fmt, blocks, height, width, etc_strategy=ETC_SMALLER_ERROR):
    import numpy as np
    b = np.frombuffer(blocks, np.uint8)
    dh, dw = (height + 1) // 2, (width + 1) // 2
    n = ((dh + 3) // 4) * ((dw + 3) // 4) * _block_bytes(compressor, fmt)
    out = np.zeros(max(n, 1), np.uint8)
    st = lib().icamd_downsample(compressor, etc_strategy, fmt, height, width, b.ctypes.data, out.ctypes.data, n)
    return out[:n].tobytes() if _check(st, "icamd_downsample") else None


def downsample_device(compressor, fmt, blocks, height, width, *, etc_strategy=ETC_SMALLER_ERROR, n_images=1, stream=None):
    """Compressor::Downsample on device-resident block grids: `blocks` = torch.uint8 CUDA tensor [n_images, bytes of one
    height x width image] (contiguous); returns [n_images, bytes of the halved image] or None where the reference refuses."""
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
    dh, dw = (height + 1) // 2, (width + 1) // 2
    per_out = ((dh + 3) // 4) * ((dw + 3) // 4) * _block_bytes(compressor, fmt)
    if n_images < 1 or blocks.numel() % n_images:
        raise ValueError("downsample_device: %d bytes do not divide into %d images" % (blocks.numel(), n_images))
    per_in = blocks.numel() // n_images
    need = ((height + 3) // 4) * ((width + 3) // 4) * _block_bytes(compressor, fmt)
    if per_in < need:  # the kernel's 16-byte block loads would run past the tensor
        raise ValueError("downsample_device: %d bytes per image, a %d x %d image has %d" % (per_in, height, width, need))
    out = torch.empty((n_images, per_out), dtype=torch.uint8, device=blocks.device)
    st = lib().icamd_downsample_batch_device(compressor, etc_strategy, fmt, height, width, n_images,
                                             ctypes.c_void_p(blocks.data_ptr()), per_in, ctypes.c_void_p(out.data_ptr()),
                                             per_out, per_out, _stream_handle(stream))
    return out if _check(st, "icamd_downsample_batch_device") else None


def transcode_dxt1_to_etc1_host(blocks):
    import numpy as np
    b = np.frombuffer(blocks, np.uint8).copy()
    st = lib().icamd_transcode_dxt1_to_etc1(b.ctypes.data, b.size)
    return b.tobytes() if _check(st, "icamd_transcode_dxt1_to_etc1") else None


def pvrtc_encode_region_device(src, size, first_block, n_blocks, *, out=None, stream=None):
    """icamd_pvrtc2_encode_region_device: blocks [first_block, first_block + n_blocks) of the Z-order output of the
    size x size RGBA8 image `src` (torch.uint8 CUDA tensor).  Returns the [8 * n_blocks] uint8 device tensor, or None
    where the reference would refuse the size.  No synchronisation."""
    assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
    if out is None:
        out = torch.empty((8 * n_blocks,), dtype=torch.uint8, device=src.device)
    st = lib().icamd_pvrtc2_encode_region_device(size, first_block, n_blocks, ctypes.c_void_p(src.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr()), _stream_handle(stream))
    if not _check(st, "icamd_pvrtc2_encode_region_device"):
        return None
    return out


def pvrtc_decompress_host(blocks, size):
    """icamd_pvrtc2_decompress (extension): bytes of size x size RGBA8, or None where the sizes are refused."""
    import numpy as np
    b = np.frombuffer(blocks, np.uint8)
    out = np.zeros(size * size * 4, np.uint8)
    st = lib().icamd_pvrtc2_decompress(size, b.ctypes.data, b.size, out.ctypes.data, out.size)
    return out.tobytes() if _check(st, "icamd_pvrtc2_decompress") else None


def pvrtc_workspace_size(size, n_images=1):
    return lib().icamd_pvrtc2_workspace_size(size, n_images)


def pvrtc4_workspace_size(size, n_images=1):
    return lib().icamd_pvrtc4_workspace_size(size, n_images)


def pvrtc_tune(mode=0, log2_strip=-1):
    """icamd_pvrtc2_tune (extension, test / tuning hook): 0 = automatic, 1 = always morph + encode, 2 = the one-pass kernel
    wherever it is eligible; log2_strip < 0 = automatic strip height.  Process-wide; results are identical either way."""
    return _check(lib().icamd_pvrtc2_tune(int(mode), int(log2_strip)), "icamd_pvrtc2_tune")


def pvrtc_set_workspace(workspace):
    """Caller-owned PVRTC scratch for the calls that follow on this thread (a torch.uint8 CUDA tensor of at least
    pvrtc_workspace_size bytes; needed to capture PVRTC launches into a HIP graph).  None: the library's own buffer."""
    if workspace is None:
        _tls.workspace = None
        return _check(lib().icamd_pvrtc2_set_workspace(None, 0), "icamd_pvrtc2_set_workspace")
    assert workspace.is_cuda and workspace.dtype == torch.uint8 and workspace.is_contiguous()
    if workspace.device.index not in (None, torch.cuda.current_device()):
        raise ValueError("PVRTC workspace lives on %s but the current device is cuda:%d"
                         % (workspace.device, torch.cuda.current_device()))
    ok = _check(lib().icamd_pvrtc2_set_workspace(ctypes.c_void_p(workspace.data_ptr()), workspace.numel()),
                "icamd_pvrtc2_set_workspace")
    # the C side keeps only the raw pointer (per host thread): hold the tensor until the override is cleared, so that
    # a caller dropping its reference cannot leave the library writing into freed memory
    _tls.workspace = workspace if ok else None
    return ok


def compress_batch_host(compressor, fmt, images, height, width, devices, *, padding_bytes_per_row=0,
                        etc_strategy=ETC_SMALLER_ERROR):
    """icamd_compress_batch: `images` = list of numpy uint8 arrays (host), `devices` = list of HIP ordinals.
    Returns a list of bytes (None where the reference would return false)."""
    import numpy as np
    n = len(images)
    size = compute_compressed_data_size(compressor, fmt, height, width)
    srcs = [np.ascontiguousarray(im, dtype=np.uint8).reshape(-1) for im in images]
    outs = [np.zeros(max(size, 1), np.uint8) for _ in range(n)]
    in_ptrs = (ctypes.c_void_p * n)(*[s.ctypes.data for s in srcs])
    out_ptrs = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    devs = (ctypes.c_int * len(devices))(*devices)
    statuses = (ctypes.c_int * n)()
    st = lib().icamd_compress_batch(compressor, etc_strategy, fmt, height, width, padding_bytes_per_row, n, in_ptrs,
                                    out_ptrs, size, devs, len(devices), statuses)
    if st < 0:
        _check(st, "icamd_compress_batch")
    return [outs[i][:size].tobytes() if statuses[i] == OK else None for i in range(n)]


def create_solid_device(compressor, fmt, height, width, color, *, device=None, out=None, stream=None):
    """Compressor::CreateSolidImage into a device-resident block grid; returns the uint8 device tensor, or None where
    the reference returns false."""
    c = bytes(bytearray(color))
    buf = (ctypes.c_uint8 * max(len(c), 4))(*c)
    n = ((height + 3) // 4) * ((width + 3) // 4) * _block_bytes(compressor, fmt)
    if out is None:
        out = torch.empty((max(n, 1),), dtype=torch.uint8, device=device or torch.device("cuda", torch.cuda.current_device()))
    st = lib().icamd_create_solid_device(compressor, fmt, height, width, buf, ctypes.c_void_p(out.data_ptr()), n,
                                         _stream_handle(stream))
    return out[:n] if _check(st, "icamd_create_solid_device") else None


def create_solid_batch_device(compressor, fmt, height, width, colors, *, device=None, stream=None):
    """icamd_create_solid_batch_device (extension): len(colors) solid images in one launch -> [n, bytes] device tensor."""
    comps = 3 if fmt in (RGB, BGR) else 4
    flat = bytes(bytearray(b for c in colors for b in bytearray(c)[:comps]))
    buf = (ctypes.c_uint8 * max(len(flat), 4))(*flat)
    per = ((height + 3) // 4) * ((width + 3) // 4) * _block_bytes(compressor, fmt)
    out = torch.empty((len(colors), max(per, 1)), dtype=torch.uint8, device=device or torch.device("cuda", torch.cuda.current_device()))
    st = lib().icamd_create_solid_batch_device(compressor, fmt, height, width, len(colors), buf, ctypes.c_void_p(out.data_ptr()),
                                               out.shape[1], per, _stream_handle(stream))
    return out[:, :per] if _check(st, "icamd_create_solid_batch_device") else None


def pad_batch_device(compressor, fmt, blocks, compressed_height, compressed_width, padded_height, padded_width, *,
                     etc_strategy=ETC_SMALLER_ERROR, stream=None):
    """icamd_pad_batch_device (extension): blocks = [n, bytes] device tensor of equally shaped grids -> [n, bytes] padded."""
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous() and blocks.dim() == 2
    per = ((padded_height + 3) // 4) * ((padded_width + 3) // 4) * _block_bytes(compressor, fmt)
    need = ((compressed_height + 3) // 4) * ((compressed_width + 3) // 4) * _block_bytes(compressor, fmt)
    if blocks.shape[1] < need:
        raise ValueError("pad_batch_device: %d bytes per image, the source grid takes %d" % (blocks.shape[1], need))
    out = torch.empty((blocks.shape[0], max(per, 1)), dtype=torch.uint8, device=blocks.device)
    st = lib().icamd_pad_batch_device(compressor, etc_strategy, fmt, compressed_height, compressed_width, blocks.shape[0],
                                      ctypes.c_void_p(blocks.data_ptr()), blocks.shape[1], padded_height, padded_width,
                                      ctypes.c_void_p(out.data_ptr()), out.shape[1], per, _stream_handle(stream))
    return out[:, :per] if _check(st, "icamd_pad_batch_device") else None


def copy_subimage_batch_device(compressor, fmt, blocks, compressed_height, compressed_width, start_row, start_column, height,
                               width, *, stream=None):
    """icamd_copy_subimage_batch_device (extension): the same window of every grid of blocks = [n, bytes]."""
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous() and blocks.dim() == 2
    per = ((height + 3) // 4) * ((width + 3) // 4) * _block_bytes(compressor, fmt)
    need = ((compressed_height + 3) // 4) * ((compressed_width + 3) // 4) * _block_bytes(compressor, fmt)
    if blocks.shape[1] < need:
        raise ValueError("copy_subimage_batch_device: %d bytes per image, the source grid takes %d" % (blocks.shape[1], need))
    out = torch.empty((blocks.shape[0], max(per, 1)), dtype=torch.uint8, device=blocks.device)
    st = lib().icamd_copy_subimage_batch_device(compressor, fmt, compressed_height, compressed_width, blocks.shape[0],
                                                ctypes.c_void_p(blocks.data_ptr()), blocks.shape[1], start_row, start_column,
                                                height, width, ctypes.c_void_p(out.data_ptr()), out.shape[1], per,
                                                _stream_handle(stream))
    return out[:, :per] if _check(st, "icamd_copy_subimage_batch_device") else None


def create_solid_host(compressor, fmt, height, width, color):
    import numpy as np
    c = bytes(bytearray(color))
    buf = (ctypes.c_uint8 * max(len(c), 4))(*c)
    n = ((height + 3) // 4) * ((width + 3) // 4) * _block_bytes(compressor, fmt)
    out = np.zeros(max(n, 1), np.uint8)
    st = lib().icamd_create_solid(compressor, fmt, height, width, buf, out.ctypes.data, n)
    return out[:n].tobytes() if _check(st, "icamd_create_solid") else None


def container_size(container, codec, height, width, levels=1):
    return int(lib().icamd_container_size(container, codec, height, width, levels))


def container_write(container, codec, height, width, levels_data):
    """Frames the block streams of the mip levels (largest first, bytes-like each) as a DDS / KTX / PKM / PVR file image
    (extension, include/ic_amd.h: the reference has no container code); bytes, or None where the C side says false."""
    import numpy as np
    levels = [np.frombuffer(bytes(b), np.uint8) for b in levels_data]
    n = len(levels)
    total = container_size(container, codec, height, width, n)
    out = np.zeros(max(total, 1), np.uint8)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[ctypes.c_void_p(a.ctypes.data if a.size else 0) for a in levels])
    sizes = (ctypes.c_size_t * max(n, 1))(*[a.size for a in levels])
    st = lib().icamd_container_write(container, codec, height, width, n, ptrs, sizes, out.ctypes.data, total)
    return out[:total].tobytes() if _check(st, "icamd_container_write") else None


def copy_subimage_device(compressor, fmt, blocks, compressed_height, compressed_width, start_row, start_column, height,
                         width, *, stream=None):
    """Compressor::CopySubimage on a device-resident block grid (torch.uint8 CUDA tensor); device tensor or None."""
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
    n = ((height + 3) // 4) * ((width + 3) // 4) * _block_bytes(compressor, fmt)
    out = torch.empty((max(n, 1),), dtype=torch.uint8, device=blocks.device)
    st = lib().icamd_copy_subimage_device(compressor, fmt, compressed_height, compressed_width,
                                          ctypes.c_void_p(blocks.data_ptr()), start_row, start_column, height, width,
                                          ctypes.c_void_p(out.data_ptr()), n, _stream_handle(stream))
    return out[:n] if _check(st, "icamd_copy_subimage_device") else None


def copy_subimage_host(compressor, fmt, blocks, compressed_height, compressed_width, start_row, start_column, height,
                       width):
    import numpy as np
    b = np.frombuffer(blocks, np.uint8)
    n = ((height + 3) // 4) * ((width + 3) // 4) * _block_bytes(compressor, fmt)
    out = np.zeros(max(n, 1), np.uint8)
    st = lib().icamd_copy_subimage(compressor, fmt, compressed_height, compressed_width, b.ctypes.data, start_row,
                                   start_column, height, width, out.ctypes.data, n)
    return out[:n].tobytes() if _check(st, "icamd_copy_subimage") else None


def encode_batch_sharded_device(codec, srcs, height, width, src_components, devices, *, swap_rb=False,
                                etc_strategy=ETC_SMALLER_ERROR, row_stride_bytes=None, outs=None, gather_device=-1,
                                gathered=None):
    """icamd_encode_batch_sharded_device: `srcs` = list of torch.uint8 CUDA tensors, image i on device
    devices[i % len(devices)]; `outs` = optional list of per-image output tensors (same devices); gather_device >= 0
    additionally collects every image into `gathered` ([n, encoded_size] uint8 on that device, allocated if None).
    Returns (statuses, outs, gathered); synchronous."""
    n = len(srcs)
    per = encoded_size(codec, height, width)
    stride = width * src_components if row_stride_bytes is None else row_stride_bytes
    for i, s in enumerate(srcs):
        assert s.is_cuda and s.dtype == torch.uint8 and s.is_contiguous()
        assert s.device.index == devices[i % len(devices)], "image %d is not on its listed device" % i
    if gather_device >= 0 and gathered is None:
        gathered = torch.empty((n, per), dtype=torch.uint8, device=torch.device("cuda", gather_device))
    # The sources / outputs were produced on torch streams (and come from torch's caching allocator) on EVERY listed
    # device, the library uses its own streams: wait for all of them, not just the current device.
    involved = set(devices[i % len(devices)] for i in range(n))
    if gather_device >= 0:
        involved.add(gather_device)
    for d in sorted(involved):
        if 0 <= d < torch.cuda.device_count():  # (a bad ordinal is the C side's to refuse, with its own status)
            torch.cuda.synchronize(d)
    in_ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    out_ptrs = None if outs is None else (ctypes.c_void_p * n)(*[(o.data_ptr() if o is not None else None) for o in outs])
    devs = (ctypes.c_int * len(devices))(*devices)
    statuses = (ctypes.c_int * n)()
    st = lib().icamd_encode_batch_sharded_device(codec, etc_strategy, src_components, int(swap_rb), height, width, stride,
                                                 n, in_ptrs, out_ptrs, devs, len(devices), gather_device,
                                                 None if gathered is None else ctypes.c_void_p(gathered.data_ptr()),
                                                 per if gathered is None else gathered.stride(0), statuses)
    if st < 0:
        _check(st, "icamd_encode_batch_sharded_device")
    return list(statuses), outs, gathered


def clock_probe_buffer():
    """A zeroed result buffer for clock_probe, allocated and cleared NOW (synchronised): allocate the buffers of all
    probes before the launches they are to overlap, or the clearing kernel queues up behind those launches."""
    out = torch.zeros(2, dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.current_stream().synchronize()
    return out


def clock_probe(duration_us, stream, out=None):
    """Enqueues the one-wave clock probe on `stream`; returns a callable that (after a synchronize) yields the mean
    shader clock in MHz over the probe's interval, or None if the counters are unusable."""
    if out is None:
        out = clock_probe_buffer()
    khz = lib().icamd_wall_clock_rate_khz()
    st = lib().icamd_clock_probe_device(ctypes.c_void_p(out.data_ptr()), int(duration_us), _stream_handle(stream))
    _check(st, "icamd_clock_probe_device")

    def result():
        cyc, ticks = [int(v) for v in out.cpu().tolist()]
        if ticks <= 0 or khz == 0:
            return None
        return {"shader_MHz": cyc / ticks * khz / 1e3, "shader_cycles": cyc, "ref_ticks": ticks, "ref_clock_kHz": khz,
                "interval_ms": ticks / khz}
    return result


class RcclGather:
    """The product's own gather of the compressed output (icamd_gather_blocks_rccl: one grouped ncclSend / ncclRecv exchange
    into rank `root`'s HBM).  torch.distributed is only used to hand rank 0's ncclUniqueId to the other ranks (any transport
    would do: the C entry points take the 128 bytes); the communicator and the collective are the library's."""

    def __init__(self, rank, world, broadcast_bytes, agree=None):
        """broadcast_bytes(bytes or None) -> bytes: hands rank 0's argument to every rank (collective).
        agree(bool) -> bool: True iff every rank passed True (collective; None in a world of one rank).  Every rank makes the SAME
        sequence of collective calls whatever fails where -- a rank without librccl, or rank 0 without an id, must not leave the
        others waiting inside ncclCommInitRank."""
        L = lib()
        agree = agree or (lambda ok: ok)
        self.rank, self.world = rank, world
        self.comm = ctypes.c_void_p()
        available = bool(L.icamd_rccl_available())
        why = None if available else "librccl could not be bound: %s" % L.icamd_last_error().decode()
        uid = (ctypes.c_uint8 * RCCL_UNIQUE_ID_BYTES)()
        if rank == 0 and available:
            st = L.icamd_rccl_get_unique_id(uid)
            if st != OK:
                available, why = False, "icamd_rccl_get_unique_id failed with status %d: %s" % (st, L.icamd_last_error().decode())
                uid = (ctypes.c_uint8 * RCCL_UNIQUE_ID_BYTES)()
        raw = broadcast_bytes(bytes(uid) if rank == 0 else None)
        if available and not any(raw):
            available, why = False, "rank 0 could not create an ncclUniqueId"
        if not agree(available):
            raise BackendError(why or "another rank cannot use librccl")
        uid = (ctypes.c_uint8 * RCCL_UNIQUE_ID_BYTES)(*raw)
        st = L.icamd_rccl_comm_init(ctypes.byref(self.comm), world, rank, uid)
        if st != OK:
            raise BackendError("icamd_rccl_comm_init failed with status %d: %s" % (st, L.icamd_last_error().decode()))

    def gather(self, local, bufs, counts_bytes, root=0, stream=None):
        """local: this rank's uint8 device tensor (counts_bytes[rank] bytes); bufs: on `root`, one device tensor per rank (any
        placement: their offsets from the lowest address are passed on), elsewhere None.  Enqueued, not synchronised."""
        n = self.world
        counts = (ctypes.c_size_t * n)(*[int(c) for c in counts_bytes])
        base, offs = None, None
        if self.rank == root:
            ptrs = [b.data_ptr() for b in bufs]
            lo = min(p for p, c in zip(ptrs, counts_bytes) if c > 0) if any(c > 0 for c in counts_bytes) else 0
            base = ctypes.c_void_p(lo)
            offs = (ctypes.c_size_t * n)(*[(p - lo) if c > 0 else 0 for p, c in zip(ptrs, counts_bytes)])
        st = lib().icamd_gather_blocks_rccl(self.comm, self.rank, n, root, counts,
                                            ctypes.c_void_p(local.data_ptr()) if local.numel() else None, base, offs,
                                            _stream_handle(stream))
        if st != OK:
            raise BackendError("icamd_gather_blocks_rccl failed with status %d: %s" % (st, lib().icamd_last_error().decode()))

    def destroy(self):
        if self.comm:
            lib().icamd_rccl_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()

