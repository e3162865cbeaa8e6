#!/bin/bash
# Round 3 session p: cache-policy bits on the GEMM's operand DMA (lab variants aux 1 = sc0, 2 = nt, 3 = sc0 nt) A/B on the vision bench; vision yardstick test.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
timeout 120 python bench.py $V > $O/vis_aux0_a.json 2> $O/vis.err
for a in 1 2 3; do timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_aux$a.so bench.py $V > $O/vis_aux$a.json 2> $O/vis.err; done
timeout 120 python bench.py $V > $O/vis_aux0_b.json 2> $O/vis.err
timeout 300 python -m pytest tests/test_gpu_vision.py -q -s -k "config1" > $O/pytest.log 2>&1; grep -E "passed|failed|HF CLIP" $O/pytest.log | tail -4
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3p/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"])
    except Exception as e:
        print(f, "ERR", e)
PY
