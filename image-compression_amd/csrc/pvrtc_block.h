// pvrtc_block.h -- PVRTC1 2bpp per-block / per-pixel math (placeholder).
#ifndef ICAMD_PVRTC_BLOCK_H_
#define ICAMD_PVRTC_BLOCK_H_
#include "ic_device.h"
namespace icamd {
#if defined(ICAMD_HOST_EMULATION)
static inline int emul_pvrtc2(const uint8_t *, uint32_t, uint8_t *) { return 0; }
#endif
}  // namespace icamd
#endif
