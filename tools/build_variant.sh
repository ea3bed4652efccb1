#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: experimental build of the library into
# superresolution_amd/lib/variants/libmmc_NAME.so (select it with MMC_HIP_LIB=...).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/superresolution_amd/lib/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  -o $root/superresolution_amd/lib/variants/libmmc_$name.so \
  $root/superresolution_amd/csrc/mmc_kernels.hip $root/superresolution_amd/csrc/mmc_capi.cpp 2>&1 | grep -E "error" || true
ls -la $root/superresolution_amd/lib/variants/libmmc_$name.so
