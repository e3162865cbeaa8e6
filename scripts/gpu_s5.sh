#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s5
PGV_GEMM_CFG=4 timeout 600 python -m pytest tests/test_gpu_vision.py -x -q -k "gemm" > gpurun_out/s5/pytest_gemm_cfg4.log 2>&1
tail -5 gpurun_out/s5/pytest_gemm_cfg4.log
PGV_GEMM_CFG=4 timeout 300 python scripts/microbench.py gemm > gpurun_out/s5/gemm_cfg4.log 2>&1
PGV_GEMM_CFG=4 PGV_GEMM_ABLATE=8 timeout 200 python scripts/lab/pp_stamps.py > gpurun_out/s5/stamps.log 2>&1
for abl in 1 3 4 6 7; do
  PGV_GEMM_CFG=4 PGV_GEMM_ABLATE=$abl timeout 200 python scripts/microbench.py ablate > gpurun_out/s5/ablate_$abl.log 2>&1
done
