#!/bin/bash
# Round 3 session i: X1 ViT attention (the odd token handled apart): parity, per-kernel time, vision bench A/B (PGV_ATTN_X1 = 0 padded, 1 CB 4, 2 CB 2).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vision.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -3 $O/pytest.log
for x in 0 1 2; do echo "PGV_ATTN_X1=$x"; PGV_ATTN_X1=$x timeout 120 python scripts/microbench.py attn 2>&1 | tail -4; done | tee $O/attn_micro.txt
V="--workload vision --steps 10 --warmup 3 --no-host-frames"
for x in 1 0 2 1 0; do
  PGV_ATTN_X1=$x timeout 120 python bench.py $V > $O/vis_x1_${x}_$RANDOM.json 2> $O/vis.err
done
PGV_ATTN_X1=1 timeout 120 python bench.py $V --image 336 > $O/vis336_x1_1.json 2> $O/vis.err
PGV_ATTN_X1=0 timeout 120 python bench.py $V --image 336 > $O/vis336_x1_0.json 2> $O/vis.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3i/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        fam = d.get("families", {})
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"],
              {k: (round(v["ms_per_step_est"], 2), round(v["avg_us"], 1)) for k, v in fam.items() if k in ("gemm", "vit_attn")})
    except Exception as e:
        print(f, "ERR", e)
PY
