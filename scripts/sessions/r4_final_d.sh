#!/bin/bash
# Round-4 closing session after the direct GEMM epilogue: full GPU suite, the driver's bench command, rocprofv3 kernel trace of the same code,
# vision-only / 336 px / fp8 / 32-clip lines.  Everything lands in gpurun_out/$1 (summaries are copied to profiles/r04_d_* by hand).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r4d}; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -4 $O/pytest.log | cut -c1-200; tail -3 $O/pytest.time
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -c 300 $O/bench.err; tail -3 $O/bench.time
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side > $O/bench_under_rocprof.json 2> $O/rocprof.err
python scripts/rocprof_summary.py $O/prof/bench_results.db > $O/kernel_trace.txt 2>&1
rm -rf $O/prof
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 600 python bench.py $S --workload vision --steps 10 --warmup 3 > $O/bench_vision_only.json 2> $O/bench_vision_only.err
timeout 600 python bench.py $S --image 336 > $O/bench_image336.json 2> $O/bench_image336.err
timeout 600 python bench.py $S --weights fp8 > $O/bench_7b_fp8.json 2> $O/bench_7b_fp8.err
timeout 600 python bench.py $S --clips-per-gpu 64 --steps 2 > $O/bench_clips64.json 2> $O/bench_clips64.err
for L in 1 2 3; do PGV_VIT_LANES=$L timeout 300 python bench.py $S --workload vision --steps 10 --warmup 3 > $O/bench_vision_lanes$L.json 2> $O/bench_vision_lanes$L.err; done
for f in bench bench_under_rocprof bench_vision_lanes1 bench_vision_lanes2 bench_vision_lanes3 bench_vision_only bench_image336 bench_7b_fp8 bench_clips64; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f clip_ms %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), (d.get("roofline") or {}).get("frac"), (d.get("runner") or {}).get("ratio_to_value"))
    if "side" in d: print("  side:", {k: (v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("clip_feat_frac")) for k, v in d["side"].items()})
    if "cpu_baseline" in d: c=d["cpu_baseline"]; print("  cpu:", c.get("value"), c.get("cores"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -14 $O/kernel_trace.txt | cut -c1-150
