#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s10
PGV_GEMM_CFG=6 timeout 600 python -m pytest tests/test_gpu_vision.py -x -q -k "gemm" > gpurun_out/s10/pytest_gemm_cfg6.log 2>&1
tail -3 gpurun_out/s10/pytest_gemm_cfg6.log
for abl in 0 8 16 24 14 30; do
  PGV_GEMM_CFG=6 PGV_GEMM_ABLATE=$abl timeout 200 python scripts/microbench.py ablate > gpurun_out/s10/ablate_$abl.log 2>&1
done
