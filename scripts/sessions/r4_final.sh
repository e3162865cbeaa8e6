#!/bin/bash
# Round-4 measurement session: full GPU suite, the driver's bench command, rocprofv3 kernel trace of the same code, PMC traffic passes, side lines,
# config 5 (13B fp8) with its own kernel trace + PMC pass, the self-launched 2-rank run on one device.  Everything lands in gpurun_out/$1
# (summaries are copied to profiles/ by hand).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r4final}; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -4 $O/pytest.log | cut -c1-200; tail -3 $O/pytest.time
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -c 300 $O/bench.err; tail -3 $O/bench.time
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side > $O/bench_under_rocprof.json 2> $O/rocprof.err
python scripts/rocprof_summary.py $O/prof/bench_results.db > $O/kernel_trace.txt 2>&1
python scripts/trace_gaps.py $O/prof/bench_results.db > $O/gaps.txt 2>&1
rm -rf $O/prof
ARGS="--steps 1 --warmup 0 --new-tokens 9 --no-cpu-baseline --no-profile-pass --no-host-frames --no-latency --no-runner --no-side"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o pmc -- python bench.py $ARGS > $O/fetch.json 2> $O/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o pmc -- python bench.py $ARGS > $O/write.json 2> $O/write.err
python scripts/pmc_summary.py $O/fetch/pmc_results.db $O/write/pmc_results.db $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
rm -rf $O/fetch $O/write
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 600 python bench.py $S --weights fp8 > $O/bench_7b_fp8.json 2> $O/bench_7b_fp8.err
timeout 900 python bench.py $S --llm 13b > $O/bench_13b_bf16.json 2> $O/bench_13b_bf16.err
timeout 600 python bench.py $S --workload vision --steps 10 --warmup 3 > $O/bench_vision_only.json 2> $O/bench_vision_only.err
timeout 600 python bench.py $S --image 336 > $O/bench_image336.json 2> $O/bench_image336.err
timeout 600 python bench.py $S --clips-per-gpu 16 > $O/bench_clips16.json 2> $O/bench_clips16.err
timeout 600 python bench.py $S --clips-per-gpu 64 --steps 2 > $O/bench_clips64.json 2> $O/bench_clips64.err
timeout 600 python bench.py $S --clips-per-gpu 32 --steps 2 > $O/bench_clips32.json 2> $O/bench_clips32.err
timeout 600 python bench.py $S --clips-per-gpu 32 --weights fp8 --steps 2 > $O/bench_clips32_fp8.json 2> $O/bench_clips32_fp8.err
PGV_FP8_MFMA=1 timeout 900 python bench.py $S --weights fp8 --llm 13b > $O/bench_13b_fp8_mfma.json 2> $O/bench_13b_fp8_mfma.err
# config 5 as a measured path of its own: kernel trace + PMC of the 13B fp8 line
C5="--llm 13b --weights fp8 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof5 -o bench -- python bench.py --steps 1 --warmup 1 $C5 > $O/bench_13b_fp8_under_rocprof.json 2> $O/rocprof5.err
python scripts/rocprof_summary.py $O/prof5/bench_results.db > $O/kernel_trace_13b_fp8.txt 2>&1
rm -rf $O/prof5
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch5 -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $C5 > $O/fetch5.json 2> $O/fetch5.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write5 -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $C5 > $O/write5.json 2> $O/write5.err
python scripts/pmc_summary.py $O/fetch5/pmc_results.db $O/write5/pmc_results.db $O/pmc_traffic_13b_fp8.json > $O/pmc_traffic_13b_fp8.txt 2>&1
rm -rf $O/fetch5 $O/write5
(PGV_BENCH_SHARE_DEVICE=1 PGV_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-host-frames --no-latency --no-profile-pass > $O/bench_gpus2_shared.log 2>&1; echo rc=$? >> $O/bench_gpus2_shared.log)
for f in bench bench_7b_fp8 bench_13b_bf16 bench_vision_only bench_image336 bench_clips16 bench_clips32 bench_clips64 bench_clips32_fp8 bench_13b_fp8_mfma; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f clip_ms %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), (d.get("roofline") or {}).get("frac"), (d.get("runner") or {}).get("ratio_to_value"))
    if "side" in d: print("  side:", {k: (v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("clip_feat_frac")) for k, v in d["side"].items()})
    if "cpu_baseline" in d: c=d["cpu_baseline"]; print("  cpu:", c.get("value"), c.get("cores"), c.get("threads_used"), c.get("decode_step_s_per_layer"), c.get("child_wall_s"), c.get("cgroup_cpu_quota"), c.get("error"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -14 $O/kernel_trace.txt | cut -c1-150
grep '^{' $O/bench_gpus2_shared.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks shared device:', d['value'], d['n_gpus'], d.get('collective'))"
