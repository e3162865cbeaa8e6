#!/bin/bash
# SQ counters of the ViT attention kernel (LDS conflicts / activity, VALU / MFMA busy), one group per pass, --kernel-trace only.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_attn
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_attn/p$i -o pmc -- python scripts/microbench.py attn > gpurun_out/pmc_attn/p$i.log 2>&1
  python - gpurun_out/pmc_attn/p$i/pmc_results.db <<'PY'
import sqlite3, sys, collections
try:
    c = sqlite3.connect(sys.argv[1])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%vit_attn%'"):
        a = agg[cn]; a[0] += 1; a[1] += val
    for k, (n, v) in sorted(agg.items()):
        print(f"{k:28s} dispatches {n:4d}  avg {v / n:16.1f}")
except Exception as e:
    print("pmc read failed:", e)
PY
done
rm -rf gpurun_out/pmc_attn/p*/
