#!/bin/bash
# SQ counters of the level-0 smoother pass on the fixed-work micro benchmark (two rocprofv3 --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
mkdir -p $R/gpurun_out/r3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/r3/pmc_s0_a -- python3 $R/scripts/gpu_sweep_micro.py 255 4 > $R/gpurun_out/r3/pmc_s0_a.log 2>&1
echo rc=$?
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/r3/pmc_s0_b -- python3 $R/scripts/gpu_sweep_micro.py 255 4 > $R/gpurun_out/r3/pmc_s0_b.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, os, re
from collections import defaultdict
R = os.environ["GRAFT_REPO_ROOT"]
for tag in ("a", "b"):
    d = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
    for f in glob.glob(f"{R}/gpurun_out/r3/pmc_s0_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("vof::", "").strip()
            if "sweep0" not in k: continue
            d[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in d:
        print(tag, k, "launches", len(n[k]), {c: f"{v / len(n[k]):.4g}" for c, v in sorted(d[k].items())})
PY
