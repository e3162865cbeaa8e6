#!/bin/bash
# Round-6 session D: blocked activation layout for the wide-batch consumers in the product path: the decoder tests (batch invariance at 32 / 64 is
# bitwise), wide-batch bench lines and a kernel trace at 64 clips; RCCL log capture after the flush fix.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6d}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_runners.py tests/test_gpu_sampling.py -q -s -x --durations=8 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -14 $O/pytest.log | cut -c1-250; tail -3 $O/pytest.time
grep -E "shaped 2-layer" $O/pytest.log | cut -c1-400
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 600 python bench.py --steps 2 --warmup 1 $Q --clips-per-gpu 64 > $O/bench_clips64.json 2> $O/bench_clips64.err
timeout 600 python bench.py --steps 2 --warmup 1 $Q --clips-per-gpu 32 > $O/bench_clips32.json 2> $O/bench_clips32.err
timeout 600 python bench.py --steps 3 --warmup 1 $Q > $O/bench_clips8.json 2> $O/bench_clips8.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof64 -o bench -- python bench.py --steps 1 --warmup 1 $Q --clips-per-gpu 64 --no-profile-pass > $O/bench_under_rocprof_clips64.json 2> $O/rocprof64.err
python scripts/rocprof_summary.py $O/prof64/bench_results.db > $O/kernel_trace_clips64.txt 2>&1; rm -rf $O/prof64
for f in bench_clips64 bench_clips32 bench_clips8; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), (d.get("roofline") or {}).get("frac"), {k: round(v["avg_us"],1) for k,v in d["families"].items()}, (d.get("roofline") or {}).get("token_check"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep -E "gemv|decode_attn|argmax|embed_tok" $O/kernel_trace_clips64.txt | cut -c1-200
