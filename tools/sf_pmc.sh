out=$GRAFT_REPO_ROOT/gpurun_out/r2h/sfpmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  name=${set%% *}; ctrs=${set#* }
  timeout -s KILL 120 rocprofv3 --pmc $ctrs -d $out/$name -o vm --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_singleframe.py 6 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/$name > $out/$name.json; rm -rf $out/$name
done
python3 - <<'PY'
import json, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2h/sfpmc/"
d = {}
for n in ("sq1", "sq2"):
    for k, v in json.load(open(out + n + ".json")).items():
        d.setdefault(k, {}).update(v)
for k, v in d.items():
    if "k_sf" in k and "<2" not in k:
        w = v.get("SQ_WAVES", 1)
        print(k[:40], "waves %d VALU/wave %.0f SALU/wave %.0f LDS/wave %.0f VMEMRD/wave %.0f SMEM/wave %.0f wavecyc/wave %.0f wait_any %.2f active_valu %.2f ldsconf/wave %.0f" % (
            w, v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w, v["SQ_INSTS_VMEM_RD"] / w, v["SQ_INSTS_SMEM"] / w,
            v["SQ_WAVE_CYCLES"] / w, v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_LDS_BANK_CONFLICT"] / w))
PY
