#!/bin/bash
# round 4, session f: K-order rotation of the decode GEMVs -- parity + A/B in the chain labs (fp8 and 16-bit) + bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/video_llava_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_gpu_llm.py -q -x > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
for r in 1 0; do
echo "== fp8 chain, PGV_GEMV_ROT=$r =="; PGV_GEMV_ROT=$r timeout 200 scripts/lab/gemv8_chain.exe 2>&1 | grep -v "bare" | tee $O/chain8_rot$r.log
echo "== 16-bit chain, PGV_GEMV_ROT=$r =="; PGV_GEMV_ROT=$r timeout 200 scripts/lab/gemv_chain.exe 2>&1 | grep -v "plain resid\|side loads\|LDS reduce" | tee $O/chain16_rot$r.log
done
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-runner --no-side --no-latency"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    f=d.get("families",{})
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), "dattn %.2f us" % f.get("decode_attn",{}).get("avg_us",-1), "gemv %.2f us" % f.get("decode_gemv",{}).get("avg_us",-1), "roof", d["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for r in 1 0; do
  PGV_GEMV_ROT=$r timeout 300 python bench.py $S > $O/b7_rot$r.json 2> $O/b7_rot$r.err; show $O/b7_rot$r.json
  PGV_GEMV_ROT=$r timeout 300 python bench.py $S --llm 13b --weights fp8 > $O/b13_rot$r.json 2> $O/b13_rot$r.err; show $O/b13_rot$r.json
done
