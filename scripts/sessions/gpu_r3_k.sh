#!/bin/bash
# Round 3 session k: the full GPU parity suite (incl. the new full-depth bf16 / yardstick tests) + smoke.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --durations=12 > $O/pytest.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error|\[7b|\[13b|yardstick|folded|FAILED|Error" $O/pytest.log | tail -40
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
