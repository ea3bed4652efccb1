#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: experimental build of the library into
# superresolution_amd/lib/variants/libmmc_NAME.so (select it with MMC_HIP_LIB=...).
# -DMMC_ABLATION compiles in the MMC_DEBUG switches (bit mask: 1 no scatter, 2 no scatter atomics, 4 no A'p,
# 32 plain block order, 128 cA from memory, 256 no warp taps, 512 no data term,
# 1024 taps gathered from global memory, 2048 / 4096 generic A'p / data-term paths).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/superresolution_amd/lib/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  -o $root/superresolution_amd/lib/variants/libmmc_$name.so \
  $root/superresolution_amd/csrc/mmc_kernels.hip $root/superresolution_amd/csrc/mmc_capi.cpp $root/superresolution_amd/csrc/mmc_group.cpp $root/superresolution_amd/csrc/mmc_sf_kernels.hip $root/superresolution_amd/csrc/mmc_sf.cpp 2>&1 | grep -E "error" || true
ls -la $root/superresolution_amd/lib/variants/libmmc_$name.so
