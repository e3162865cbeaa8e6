#!/bin/bash
# Run selected GPU tests: bash scripts/gpu_one_test.sh "<pytest args>"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/one
timeout 1200 python -m pytest $1 -x -q > gpurun_out/one/pytest.log 2>&1
tail -25 gpurun_out/one/pytest.log
