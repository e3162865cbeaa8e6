#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/t2
timeout 900 python -m pytest tests/test_gpu_vision.py -x -q > gpurun_out/t2/pytest.log 2>&1
tail -5 gpurun_out/t2/pytest.log
timeout 300 python scripts/microbench.py attn > gpurun_out/t2/attn.log 2>&1
grep -v amdgpu gpurun_out/t2/attn.log
