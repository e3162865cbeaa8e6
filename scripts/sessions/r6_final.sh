#!/bin/bash
# Round-6 closing session on the final code: full GPU suite, smoke, PMC traffic passes (7B default / 13B fp8 / 336 px -- copied into profiles/ of
# the box's tree so the bench lines that follow read THIS code's counters), the driver's bench command, rocprofv3 kernel traces of the three
# configurations, side lines, SQ counters.  Everything lands in gpurun_out/$1 (summaries are copied to profiles/r05_* afterwards).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6f}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 1500 python -m pytest tests -q -m gpu -s --durations=15 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -3 $O/pytest.log | cut -c1-200; tail -3 $O/pytest.time
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
pmc() {   # name, extra bench args
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f_$1 -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $Q $2 > $O/f_$1.json 2> $O/f_$1.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w_$1 -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $Q $2 > $O/w_$1.json 2> $O/w_$1.err
  python scripts/pmc_summary.py $O/f_$1/pmc_results.db $O/w_$1/pmc_results.db $O/pmc_$1.json > $O/pmc_$1.txt 2>&1
  rm -rf $O/f_$1 $O/w_$1
  cp $O/pmc_$1.json profiles/r06_pmc_$1.json
}
pmc traffic ""
pmc 13b_fp8 "--llm 13b --weights fp8"
pmc image336 "--image 336"
ls -la $O/pmc_*.json
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -c 300 $O/bench.err; tail -3 $O/bench.time
trace() {   # name, extra args
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$1 -o bench -- python bench.py --steps 1 --warmup 1 $Q $2 > $O/bench_under_rocprof_$1.json 2> $O/rocprof_$1.err
  python scripts/rocprof_summary.py $O/prof_$1/bench_results.db > $O/kernel_trace_$1.txt 2>&1
  rm -rf $O/prof_$1
}
trace 7b ""
trace 13b_fp8 "--llm 13b --weights fp8"
trace image336 "--image 336"
trace clips64 "--clips-per-gpu 64 --no-profile-pass"
trace clips32 "--clips-per-gpu 32 --no-profile-pass"
S="--steps 3 --warmup 1 $Q"
timeout 600 python bench.py $S --workload vision --steps 10 --warmup 3 > $O/bench_vision_only.json 2> $O/bench_vision_only.err
timeout 600 python bench.py $S --image 336 --workload vision --steps 6 --warmup 2 > $O/bench_vision_only_336.json 2> $O/bench_vision_only_336.err
timeout 600 python bench.py $S --weights fp8 > $O/bench_7b_fp8.json 2> $O/bench_7b_fp8.err
timeout 600 python bench.py $S --llm 13b > $O/bench_13b_bf16.json 2> $O/bench_13b_bf16.err
timeout 600 python bench.py $S --clips-per-gpu 16 > $O/bench_clips16.json 2> $O/bench_clips16.err
timeout 600 python bench.py $S --clips-per-gpu 32 --steps 2 > $O/bench_clips32.json 2> $O/bench_clips32.err
timeout 600 python bench.py $S --clips-per-gpu 64 --steps 2 > $O/bench_clips64.json 2> $O/bench_clips64.err
timeout 600 python bench.py $S --clips-per-gpu 64 --weights fp8 --steps 2 > $O/bench_clips64_fp8.json 2> $O/bench_clips64_fp8.err
# SQ counters per kernel at 336 px (the attention kernel's 8-wave form) and at 224 px
sq() {   # name, extra args
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $grp --kernel-trace -d $O/sq_$1/p$i -o pmc -- python bench.py --steps 1 --warmup 0 --new-tokens 9 --no-profile-pass $Q $2 > $O/sq_$1_p$i.json 2> $O/sq_$1_p$i.err
  done
  python scripts/pmc_sq_summary.py $O/sq_$1/p*/pmc_results.db > $O/pmc_sq_$1.txt 2>&1
  rm -rf $O/sq_$1 $O/sq_$1_p*.json
}
sq image336 "--image 336"
for f in bench bench_under_rocprof_7b bench_under_rocprof_13b_fp8 bench_under_rocprof_image336 bench_under_rocprof_clips64 bench_vision_only bench_vision_only_336 bench_7b_fp8 bench_13b_bf16 bench_clips16 bench_clips32 bench_clips64 bench_clips64_fp8; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f clip_ms %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), (d.get("roofline") or {}).get("frac"), (d.get("roofline_mfma") or {}).get("frac"), (d.get("runner") or {}).get("ratio_to_value"))
    if "side" in d: print("  side:", {k: (v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("clip_feat_frac"), v.get("error")) for k, v in d["side"].items()})
    if "cpu_baseline" in d: c=d["cpu_baseline"]; print("  cpu:", c.get("value"), c.get("cores"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -12 $O/kernel_trace_7b.txt | cut -c1-150
grep -E "gemv|decode_attn" $O/kernel_trace_clips64.txt | cut -c1-160
grep -E "gemv|decode_attn" $O/kernel_trace_13b_fp8.txt | cut -c1-160
head -16 $O/pmc_sq_image336.txt | cut -c1-150
