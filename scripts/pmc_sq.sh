#!/bin/bash
# SQ counters per kernel of a short bench run (LDS conflicts / activity, VALU / MFMA busy, waits): one group per pass, --kernel-trace only.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_sq
ARGS="--steps 1 --warmup 0 --new-tokens 9 --no-cpu-baseline --no-profile-pass --no-host-frames --no-latency --no-runner --no-side"
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_sq/p$i -o pmc -- python bench.py $ARGS > gpurun_out/pmc_sq/p$i.json 2> gpurun_out/pmc_sq/p$i.err
done
python scripts/pmc_sq_summary.py gpurun_out/pmc_sq/p*/pmc_results.db > gpurun_out/pmc_sq/summary.txt 2>&1
cat gpurun_out/pmc_sq/summary.txt
rm -rf gpurun_out/pmc_sq/p*/
