#!/bin/bash
# round 5, session c: the attention kernel with the half-swap fix under the vision suite, the two repaired runner tests, config 5 with the adopted
# 8-phase shapes (10 row blocks per workgroup, o_proj in that form), the 336-px line, the default bench line with the new side line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
python -c "from video_llava_amd import build; print(build.build()); print(build.build(lab=True))" > $O/build.log 2>&1; tail -2 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_vision.py tests/test_gpu_runners.py tests/test_gpu_loader.py -m gpu -q --durations=8 ) > $O/pytest_a.log 2>&1; tail -14 $O/pytest_a.log
( time timeout 600 python -m pytest tests/test_gpu_llm.py tests/test_gpu_sampling.py -m gpu -q -x --durations=5 ) > $O/pytest_b.log 2>&1; tail -10 $O/pytest_b.log
echo "== attention microbench (release library) ==" > $O/attn.txt
timeout 120 python scripts/microbench.py attn >> $O/attn.txt 2>&1; cat $O/attn.txt
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames"
timeout 300 python $B --image 336 > $O/bench_336.json 2> $O/bench_336.err; python scripts/sessions/r5_pick.py $O/bench_336.json
timeout 300 python $B --llm 13b --weights fp8 > $O/cfg5.json 2> $O/cfg5.err; echo cfg5; python scripts/sessions/r5_pick.py $O/cfg5.json
PGV_GEMV_K8_NWB10=0 timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_lab.so $B --llm 13b --weights fp8 > $O/cfg5_nwb8.json 2> $O/cfg5_nwb8.err; echo "cfg5 nwb8 (round-4 shapes)"; python scripts/sessions/r5_pick.py $O/cfg5_nwb8.json
( time timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default: value %.3f ms/step %.1f clip frac %.4f roofline %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_frac_of_mfma_peak"], d["roofline"]["frac"]))
print("roofline_mfma:", {k: d.get("roofline_mfma", {}).get(k) for k in ("achieved", "frac", "traffic", "algorithmic_bytes_per_launch", "launches_per_step")})
for k, v in d.get("side", {}).items():
    print(" side", k, v.get("value"), v.get("ms_per_step"), v.get("clip_feat_frac"), (v.get("roofline") or {}).get("frac"), (v.get("roofline_mfma") or {}).get("frac"), v.get("error"))
PY
