#!/bin/bash
# ViT attention cost split: PGV_ATTN_ABLATE bit sweep (see AttnArgs.abl).  Results with bits 8..64 set are numerically wrong by design.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/attn
for a in 0 1 4 5 2 8 16 32 64 48 72 120; do
  echo "== ablate $a" >> gpurun_out/attn/sweep.log
  PGV_ATTN_ABLATE=$a timeout 120 python scripts/microbench.py attn 2>&1 | grep "T=" >> gpurun_out/attn/sweep.log
done
cat gpurun_out/attn/sweep.log
