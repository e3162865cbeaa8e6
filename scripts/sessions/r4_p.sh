#!/bin/bash
# round 4, session p: 8-phase wide residual producers -- bitwise parity at 7B (wide vs single), ragged wide batches, side lines at 32 / 64 clips with the switch on / off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -q -x -k "wide_batch or ragged" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.log | cut -c1-220
S="--steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-runner --no-side --no-latency"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    f=d.get("families",{})
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), "dattn %.2f us" % f.get("decode_attn",{}).get("avg_us",-1), "gemv %.2f us x %d" % (f.get("decode_gemv",{}).get("avg_us",-1), f.get("decode_gemv",{}).get("launches_per_step",0)))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for k in 4 8; do for c in 32 64; do
  PGV_GEMV_K8_WAVES=$k timeout 600 python bench.py $S --clips-per-gpu $c > $O/b7_clips${c}_w$k.json 2> $O/b7_clips${c}_w$k.err; show $O/b7_clips${c}_w$k.json
done; done
PGV_GEMV_K8_WAVES=4 timeout 600 python bench.py $S --clips-per-gpu 32 --weights fp8 > $O/b7_fp8_clips32_k81.json 2> $O/b7_fp8_clips32_k81.err; show $O/b7_fp8_clips32_k81.json
