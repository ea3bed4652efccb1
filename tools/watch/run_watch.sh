#!/bin/bash
# tools/watch/run_watch.sh N [SEED0]: the group fuzz on the HIP runtime PyTorch brings along, N runs of 600
# cases with the -DMMC_WATCH build (write watchpoint on the second block's stored handle).  Stops at the
# first run that fails.  Logs: gpurun_out/watch/run<k>.log
n=${1:-6}; seed0=${2:-100}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/watch; mkdir -p $out
export MMC_HIP_LIB=$root/superresolution_amd/lib/variants/libmmc_watch.so
for k in $(seq 1 $n); do
  echo "run $k"
  timeout -k 10 170 python -u $root/tools/fuzz_group.py 600 $((seed0 + k)) > $out/run$k.log 2>&1
  rc=$?
  tail -2 $out/run$k.log
  if [ $rc -ne 0 ]; then echo "run $k: exit $rc"; grep -c "mmc_watch" $out/run$k.log; exit $rc; fi
done
