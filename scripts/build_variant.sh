#!/bin/bash
# A/B builds of libic_amd.so with extra -D flags: scripts/build_variant.sh <name> "<flags>" -> ab_libs/libic_amd_<name>.so
# (git-ignored through *.so, shipped to the GPU box by gpurun; delete ab_libs/ when the experiment is over).  Used with
# ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/ab_libs/libic_amd_<name>.so (scripts/ab_bench.sh).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2
O=$R/ab_libs/obj_$NAME; mkdir -p "$O"
for f in ic_capi dxt_kernels etc1_kernels pvrtc_kernels decode_kernels blockops_kernels diag_kernels rccl_gather; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -I$R/include -I$R/image-compression_amd/csrc $FLAGS \
    -c $R/image-compression_amd/csrc/$f.hip -o $O/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_libs/libic_amd_$NAME.so $O/*.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf "$O"
echo "built ab_libs/libic_amd_$NAME.so"
