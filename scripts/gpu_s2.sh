#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2
timeout 300 python scripts/microbench.py pad > gpurun_out/s2/pad_cfg0.log 2>&1
PGV_GEMM_CFG=3 PGV_GEMM_ABLATE=6 timeout 300 python scripts/microbench.py pad > gpurun_out/s2/pad_cfg3_abl6.log 2>&1
