#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_llm.py -q -x -k "columns_are_independent" > $O/pytest_cols.log 2>&1; echo "cols rc=$?"; tail -15 $O/pytest_cols.log | cut -c1-220
timeout 900 python scripts/sessions/r4_i.py > $O/debug.log 2>&1; cat $O/debug.log | cut -c1-300
