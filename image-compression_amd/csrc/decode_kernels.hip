// decode_kernels.hip -- DXT1/DXT5/ETC1 block decoders (placeholder).
#include "ic_launch.h"

namespace icamd {
hipError_t launch_decode(int, const DecodeParams &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace icamd
