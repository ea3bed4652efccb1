R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SMEM -d $R/gpurun_out/pmc_sq2 -o p --output-format csv -- python3 $R/tools/pmc_target.py 30 4 > $R/gpurun_out/pmc_sq2.log 2>&1 || exit 1
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_sq2 > gpurun_out/pmc_sq2.json
find gpurun_out/pmc_sq2 -name "*.csv" -size +2M -delete
