#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/t3
timeout 900 python -m pytest tests/test_gpu_llm.py -x -q > gpurun_out/t3/pytest.log 2>&1
tail -15 gpurun_out/t3/pytest.log
