#!/bin/bash
# tools/kernel_regs.sh [-DFLAG ...]: vector registers, scratch bytes per lane and waves per SIMD of every instance of
# the two sweeps (compiler remarks; nothing is linked).  The instances the headline runs are
# k_primal4<0,0,false,4> and k_dual4<0,0,false,4> (model IC, update mode, no reduction, zoom 4).
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -c -Rpass-analysis=kernel-resource-usage -Wno-unused-function "$@" \
  $root/superresolution_amd/csrc/mmc_kernels.hip -o /dev/null 2>&1 | python3 -c '
import re, sys
for b in re.split(r"remark: [^\n]*Function Name: ", sys.stdin.read())[1:]:
    name = b.split()[0]
    if "k_primal4" in name or "k_dual4" in name:
        get = lambda pat: re.search(pat, b).group(1)
        print("%-60s VGPRs %3s  scratch %3s  waves/SIMD %s" % (name, get(r"VGPRs: (\d+)"), get(r"ScratchSize \[bytes/lane\]: (\d+)"), get(r"Occupancy \[waves/SIMD\]: (\d+)")))
'
