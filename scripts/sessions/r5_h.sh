#!/bin/bash
# round 5, session h: after the 13B qkv shape change -- the 13B full-depth / end-to-end cases and the GEMV family tests on the release library,
# config 5's PMC passes + kernel trace refreshed, the default bench line (side lines included).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_fulldepth.py -m gpu -q -s -k "13b or vision_chain" ) > $O/fulldepth_13b.log 2>&1; tail -4 $O/fulldepth_13b.log | cut -c1-200
( time timeout 600 python -m pytest tests/test_gpu_llm.py -m gpu -q ) > $O/pytest_llm.log 2>&1; tail -3 $O/pytest_llm.log | cut -c1-200
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --llm 13b --weights fp8"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $Q > $O/f.json 2> $O/f.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $Q > $O/w.json 2> $O/w.err
python scripts/pmc_summary.py $O/f/pmc_results.db $O/w/pmc_results.db $O/pmc_13b_fp8.json > $O/pmc_13b_fp8.txt 2>&1; rm -rf $O/f $O/w
cp $O/pmc_13b_fp8.json profiles/r05_pmc_13b_fp8.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 1 --warmup 1 $Q > $O/bench_under_rocprof_13b_fp8.json 2> $O/rocprof.err
python scripts/rocprof_summary.py $O/prof/bench_results.db > $O/kernel_trace_13b_fp8.txt 2>&1; rm -rf $O/prof
head -12 $O/kernel_trace_13b_fp8.txt | cut -c1-160
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -2 $O/bench.time
python - $O/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default: value %.3f ms/step %.1f clip frac %.4f roofline %.4f mfma %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_frac_of_mfma_peak"], d["roofline"]["frac"], d["roofline_mfma"]["frac"]))
for k, v in d.get("side", {}).items():
    print(" side", k, v.get("value"), v.get("ms_per_step"), v.get("clip_feat_frac"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), (v.get("roofline_mfma") or {}).get("frac"), v.get("error"))
PY
