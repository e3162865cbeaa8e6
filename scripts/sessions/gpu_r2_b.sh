#!/bin/bash
# Round-2 trip B: folded RMSNorm in decode (numerics vs goldens + full-depth 7B), decode-attention wave count A/B, fp8 conversion cut.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_sampling.py tests/test_gpu_loader.py tests/test_gpu_runners.py "tests/test_gpu_fulldepth.py::test_7b_full_depth_fp16_token_exact" -m gpu -q -s > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300; grep -n "rel err\|margins" $O/pytest.log | cut -c1-260 | head -20
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency"
timeout 600 $B > $O/bench_w8.json 2> $O/bench_w8.err
PGV_DATTN_WAVES=16 timeout 600 $B > $O/bench_w16.json 2> $O/bench_w16.err
timeout 600 $B --weights fp8 > $O/bench_7b_fp8.json 2> $O/bench_7b_fp8.err
timeout 900 $B --weights fp8 --llm 13b > $O/bench_13b_fp8.json 2> $O/bench_13b_fp8.err
for f in w8 w16 7b_fp8 13b_fp8; do python - $O/bench_$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    f=d.get("families",{})
    print(sys.argv[1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), {k:(round(v["avg_us"],2), round(v.get("gbs",0))) for k,v in f.items() if k.startswith("decode")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -c 300 $O/bench_w8.err
