#!/bin/bash
# effective shader clock of the sweep micro-benchmark per build: GRBM_GUI_ACTIVE / 8 / kernel time (MI355X_MICROARCH.md, DVFS)
# usage: scripts/gpu_micro_clock.sh <lib or ""> <tag>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export VOF_LIB=$1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_$2 -- python3 $R/scripts/gpu_sweep_micro.py 255 5 > $R/gpurun_out/clk_$2.log 2>&1
python3 - <<PY
import csv,glob
ct=glob.glob("$R/gpurun_out/clk_$2/**/*counter_collection.csv",recursive=True)
kt=glob.glob("$R/gpurun_out/clk_$2/**/*kernel_trace.csv",recursive=True)
dur={}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]),r["Kernel_Name"])
for r in csv.DictReader(open(ct[0])):
    d,name=dur.get(r["Dispatch_Id"],(0,""))
    if "k_sweep" in name and d>0:
        print("$2", name[:40], "dur_us %.1f"%(d/1e3), "clock_GHz %.3f"%(float(r["Counter_Value"])/8/d))
PY
