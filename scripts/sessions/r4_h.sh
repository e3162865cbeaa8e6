#!/bin/bash
# round 4, session h: decode batches up to 64 -- parity (column independence across tiles, ragged batches, 7B wide invariance) + side lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -q -x -k "columns_are_independent or ragged or wide_batch or 7b_batch_invariance or golden or fp8_bit_equal" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.log
S="--steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-runner --no-side --no-latency"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    f=d.get("families",{})
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f clip_ms %.1f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"]), "dattn %.2f us" % f.get("decode_attn",{}).get("avg_us",-1), "gemv %.2f us" % f.get("decode_gemv",{}).get("avg_us",-1), "roof", d["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for c in 16 32 64; do
  timeout 600 python bench.py $S --clips-per-gpu $c > $O/b7_clips$c.json 2> $O/b7_clips$c.err; show $O/b7_clips$c.json
done
timeout 600 python bench.py $S --clips-per-gpu 32 --weights fp8 > $O/b7_fp8_clips32.json 2> $O/b7_fp8_clips32.err; show $O/b7_fp8_clips32.json
