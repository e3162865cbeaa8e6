#!/bin/bash
# Round 3 session j: persistent double-buffered X1 attention: parity, per-kernel time, vision bench A/B (PGV_ATTN_X1: 1 persistent, 3 X1 one unit per WG, 0 padded).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vision.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -5 $O/pytest.log
for x in 1 3 0; do echo "PGV_ATTN_X1=$x"; PGV_ATTN_X1=$x timeout 120 python scripts/microbench.py attn 2>&1 | tail -4; done | tee $O/attn_micro.txt
V="--workload vision --steps 10 --warmup 3 --no-host-frames"
for x in 1 0 3 1 0; do
  PGV_ATTN_X1=$x timeout 120 python bench.py $V > $O/vis_x1_${x}_$RANDOM.json 2> $O/vis.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3j/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        fam = d.get("families", {})
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"],
              {k: (round(v["ms_per_step_est"], 2), round(v["avg_us"], 1)) for k, v in fam.items() if k in ("gemm", "vit_attn")})
    except Exception as e:
        print(f, "ERR", e)
PY
