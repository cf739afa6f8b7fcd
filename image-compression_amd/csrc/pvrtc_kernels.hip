// pvrtc_kernels.hip -- PVRTC1 2bpp encode (placeholder until the fused tile kernel lands).
#include "ic_launch.h"

namespace icamd {
const char *pvrtc2_kernel_name() { return "icamd_pvrtc2_kernel"; }
hipError_t launch_pvrtc2(const PvrtcParams &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace icamd
