#!/bin/bash
# round 4, session z: full GPU suite + smoke on the round's last library (8-phase fp8 down_proj at batches <= 16 added after r4_final_d)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4z; mkdir -p $O
( time timeout 1200 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -4 $O/pytest.log | cut -c1-200; tail -3 $O/pytest.time
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
