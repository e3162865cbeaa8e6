#!/bin/bash
# Round-6: the whole GPU suite as the driver runs it (wall clock, durations, the measurement prints of the full-depth cases) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6s}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 2400 python -m pytest tests -q -m gpu -s --durations=25 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -34 $O/pytest.log | cut -c1-200; tail -3 $O/pytest.time
grep -E "^\[|\] " $O/pytest.log | grep -E "e2e|oracle|bench call|7b\]|13b|vision" | cut -c1-700 > $O/pytest_prints.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
