#!/bin/bash
# Round 3 session n: full GPU suite + smoke on the round's final code, then the measurement session (scripts/gpu_r3_final.sh).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $O/pytest.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error|FAILED" $O/pytest.log | tail -5
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash scripts/gpu_r3_final.sh r3final2
