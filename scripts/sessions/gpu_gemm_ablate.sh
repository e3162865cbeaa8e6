#!/bin/bash
# gemm_w4 timing ablations (PGV_GEMM_ABLATE bits, see launch_w4): results are garbage by design.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/gemm
rm -f gpurun_out/gemm/ablate.log
for a in ${1:-0 32 1 4}; do
  PGV_GEMM_ABLATE=$a timeout 300 python scripts/microbench.py ablate 2>&1 | grep -v amdgpu.ids >> gpurun_out/gemm/ablate.log
done
cat gpurun_out/gemm/ablate.log
