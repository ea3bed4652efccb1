#!/bin/bash
# tools/asan_build.sh: the library sources and tools/asan_group_stress.cpp in one executable with the host
# AddressSanitizer (host code only; the device code is built as usual): tools/asan_run/asan_group_stress.
# STRESS_NAP_MS=<ms> leaves idle time around mmc_group_create.
# Run it with ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 (the HIP runtime maps memory where the
# sanitizer's shadow gap lies, and keeps allocations of its own until exit).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/asan_run
c=$root/superresolution_amd/csrc
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -std=c++17 -Wno-unused-function -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
  -o $root/tools/asan_run/asan_group_stress $root/tools/asan_group_stress.cpp \
  $c/mmc_kernels.hip $c/mmc_capi.cpp $c/mmc_group.cpp $c/mmc_sf_kernels.hip $c/mmc_sf.cpp
ls -la $root/tools/asan_run/asan_group_stress
