#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcg
rocprofv3 -L > gpurun_out/pmcg/counters.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmcg/p1 -o p -- python scripts/microbench.py gemv > gpurun_out/pmcg/p1.log 2>&1
timeout 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --kernel-trace -d gpurun_out/pmcg/p2 -o p -- python scripts/microbench.py gemv > gpurun_out/pmcg/p2.log 2>&1
ls gpurun_out/pmcg/p1 gpurun_out/pmcg/p2
