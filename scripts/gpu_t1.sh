#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/t1
timeout 900 python -m pytest tests/test_gpu_runners.py -x -q > gpurun_out/t1/pytest.log 2>&1
tail -30 gpurun_out/t1/pytest.log
