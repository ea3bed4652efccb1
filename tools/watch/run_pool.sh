#!/bin/bash
# tools/watch/run_pool.sh N [SEED0]: the group fuzz (600 cases a run) on the HIP runtime PyTorch brings along with the
# regular build of the library.  Stops at the first run that fails.  Logs: gpurun_out/pool/run<k>.log
n=${1:-6}; seed0=${2:-200}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/pool; mkdir -p $out
for k in $(seq 1 $n); do
  echo "pool run $k"
  timeout -k 10 170 python -u $root/tools/fuzz_group.py 600 $((seed0 + k)) > $out/run$k.log 2>&1
  rc=$?
  tail -1 $out/run$k.log
  if [ $rc -ne 0 ]; then echo "pool run $k: exit $rc"; exit $rc; fi
done
