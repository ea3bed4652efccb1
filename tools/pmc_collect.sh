#!/bin/bash
# tools/pmc_collect.sh OUTDIR [frames] [steps]: hardware-counter passes over tools/pmc_target.py
# (the bench workload, inputs rendered on the host), one counter set per run as rocprofv3 needs,
# each under its own time limit; stops at the first pass that fails.  Summaries: OUTDIR/<set>.json
set -u
out=$1; L=${2:-30}; steps=${3:-4}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
out=$(cd "$out" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r name counters; do
  [ -z "$name" ] && continue
  echo "== pass $name: $counters"
  timeout -s KILL 120 rocprofv3 --pmc $counters -d "$out/$name" -o vm --output-format csv -- python3 "$root/tools/pmc_target.py" "$L" "$steps" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -n 5 "$out/$name.log"; exit 1; }
  python3 "$root/tools/pmc_summary.py" "$out/$name" > "$out/$name.json" || exit 1
  rm -rf "$out/$name"
done <<'SETS'
fetch FETCH_SIZE
write WRITE_SIZE
tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
tcc3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum
sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
sq3 SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
SETS
echo "all passes done"
