#!/bin/bash
# round 4, session a: the new benched-shape parity tests alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vision.py -q -s -k "800_frames or 336_even or w_resident" > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -15 $O/pytest.log
