#!/bin/bash
# one GPU call: baseline numbers for all GEMM configs + ablations, then the GPU test-suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s1
for cfg in 0 1 2 3; do
  PGV_GEMM_CFG=$cfg timeout 300 python scripts/microbench.py gemm > gpurun_out/s1/gemm_cfg$cfg.log 2>&1
done
for abl in 1 2 4 3 5 6; do
  PGV_GEMM_ABLATE=$abl timeout 200 python scripts/microbench.py ablate > gpurun_out/s1/ablate_$abl.log 2>&1
done
timeout 200 python scripts/microbench.py ablate > gpurun_out/s1/ablate_0.log 2>&1
timeout 200 python scripts/microbench.py attn gemv > gpurun_out/s1/attn_gemv.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1
tail -3 gpurun_out/s1/pytest.log
