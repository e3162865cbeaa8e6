#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s7
PGV_GEMM_CFG=4 timeout 600 python -m pytest tests/test_gpu_vision.py -x -q -k "gemm" > gpurun_out/s7/pytest_gemm_cfg4.log 2>&1
tail -3 gpurun_out/s7/pytest_gemm_cfg4.log
PGV_GEMM_CFG=4 timeout 300 python scripts/microbench.py gemm > gpurun_out/s7/gemm_cfg4.log 2>&1
PGV_GEMM_CFG=4 PGV_GEMM_ABLATE=9 timeout 200 python scripts/lab/pp_tile_stamps.py > gpurun_out/s7/tile_stamps.log 2>&1
