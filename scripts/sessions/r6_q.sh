#!/bin/bash
# Round-6 session Q: PMC traffic passes at 64 clips per step (the wide-batch kernel shapes) + the whole GPU suite on the final tree + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6q}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f_c64 -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $Q --clips-per-gpu 64 > $O/f_c64.json 2> $O/f_c64.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w_c64 -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $Q --clips-per-gpu 64 > $O/w_c64.json 2> $O/w_c64.err
python scripts/pmc_summary.py $O/f_c64/pmc_results.db $O/w_c64/pmc_results.db $O/pmc_clips64.json > $O/pmc_clips64.txt 2>&1
rm -rf $O/f_c64 $O/w_c64
grep -E "gemv|decode_attn" $O/pmc_clips64.txt | cut -c1-140
cp $O/pmc_clips64.json profiles/r06_pmc_clips64.json
timeout 600 python bench.py --steps 2 --warmup 1 $Q --clips-per-gpu 64 > $O/bench_clips64.json 2> $O/bench_clips64.err
python - $O/bench_clips64.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("clips64 value %.3f" % d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "gemv", d.get("roofline_gemv"))
PY
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -12 $O/pytest.log | cut -c1-200; tail -3 $O/pytest.time
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
