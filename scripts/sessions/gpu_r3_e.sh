#!/bin/bash
# Round 3 session e: epilogue with batched LDS reads (release / lab) against the previous epilogue (libpgv_prev.so): per-shape times,
# vision bench, parity tests.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
for lib in video_llava_amd/libpgv_prev.so video_llava_amd/libpgv_lab.so video_llava_amd/libpgv_prev.so video_llava_amd/libpgv_lab.so; do
  PGV_LAB_LIB=$lib timeout 200 python scripts/lab/gemm_epi_decomp.py >> $O/decomp.txt 2>> $O/decomp.err
done
cat $O/decomp.txt
timeout 900 python -m pytest tests/test_gpu_vision.py tests/test_gpu_llm.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -3 $O/pytest.log
V="--workload vision --steps 10 --warmup 3 --no-host-frames"
timeout 120 python bench.py $V > $O/vis_new1.json 2> $O/vis_new1.err
timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_prev.so bench.py $V > $O/vis_prev1.json 2> $O/vis_prev1.err
timeout 120 python bench.py $V > $O/vis_new2.json 2> $O/vis_new2.err
timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_prev.so bench.py $V > $O/vis_prev2.json 2> $O/vis_prev2.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3e/vis_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        fam = d.get("families", {})
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"],
              {k: round(v["ms_per_step_est"], 2) for k, v in fam.items() if k in ("gemm", "vit_attn")})
    except Exception as e:
        print(f, "ERR", e)
PY
