#!/bin/bash
# PMC passes over bench.py --mode align (current default protocol): gpurun_out/<tag>/pmc.json
TAG=${1:-r4p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd)
B="python3 $R/bench.py --mode align --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pf -o run -- $B > /dev/null 2>&1
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pw -o run -- $B > /dev/null 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/ps1 -o run -- $B > /dev/null 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/ps2 -o run -- $B > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $R/$OUT/pf $R/$OUT/pw $R/$OUT/ps1 $R/$OUT/ps2 > $R/$OUT/pmc.json
rm -rf $R/$OUT/pf $R/$OUT/pw $R/$OUT/ps1 $R/$OUT/ps2
cd $R
python3 - <<PY
import json
d=json.load(open("$OUT/pmc.json"))
for k,v in d.items():
    if "align" not in k: continue
    g=lambda c: v.get(c,{}).get("avg",0)
    cyc=g("GRBM_GUI_ACTIVE")/8
    print(k[:58], "cycles %.0f us %.1f"%(cyc, cyc/2400), "VALU %.1fM SALU %.1fM LDS %.2fM"%(g("SQ_INSTS_VALU")/1e6,g("SQ_INSTS_SALU")/1e6,g("SQ_INSTS_LDS")/1e6),
          "valu_issue %.2f"%(g("SQ_ACTIVE_INST_VALU")*4/1024/max(cyc,1)), "lds_conf %.2f"%(g("SQ_LDS_BANK_CONFLICT")/256/max(cyc,1)),
          "fetchMB %.0f writeMB %.0f"%(g("FETCH_SIZE")*2048/1e6, g("WRITE_SIZE")*1024/1e6), "vmem_wr %.2fM"%(g("SQ_INSTS_VMEM_WR")/1e6))
PY
