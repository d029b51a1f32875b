#!/bin/bash
# SQ counters of the hot kernels (own pass, --kernel-trace only), smaller stack to keep the pass short
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --frames 64 --no-cpu-baseline --no-variants --no-end-to-end > $R/gpurun_out/pmc_sq.log 2>&1
echo rc=$?
tail -2 $R/gpurun_out/pmc_sq.log | cut -c1-200
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/bench.py --steps 1 --warmup 1 --frames 64 --no-cpu-baseline --no-variants --no-end-to-end > $R/gpurun_out/pmc_sq2.log 2>&1
echo rc=$?
ls $R/gpurun_out/pmc_sq/*/ $R/gpurun_out/pmc_sq2/*/
