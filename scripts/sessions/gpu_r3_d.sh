#!/bin/bash
# Round 3 session d: is the GEMM K loop bound by DMA latency?  (1) ablations 8 (no counted DMA wait) and 12 (+ no MFMA) on the lab library;
# (2) the W-early DMA schedule (libpgv_wearly.so): parity tests, per-shape times, vision bench.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
for abl in 0 8 12; do
  PGV_GEMM_ABLATE=$abl timeout 200 python scripts/lab/gemm_epi_decomp.py >> $O/decomp.txt 2>> $O/decomp.err
done
for abl in 0 32 4; do
  PGV_LAB_LIB=video_llava_amd/libpgv_wearly.so PGV_GEMM_ABLATE=$abl timeout 200 python scripts/lab/gemm_epi_decomp.py >> $O/decomp.txt 2>> $O/decomp.err
done
cat $O/decomp.txt
timeout 600 python scripts/lab/with_lib.py video_llava_amd/libpgv_wearly.so -m pytest tests/test_gpu_vision.py -x -q > $O/pytest_vision_wearly.log 2>&1; echo "wearly vision tests rc=$?"
tail -3 $O/pytest_vision_wearly.log
V="--workload vision --steps 10 --warmup 3 --no-host-frames"
timeout 120 python bench.py $V > $O/vis_release.json 2> $O/vis_release.err
timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_wearly.so bench.py $V > $O/vis_wearly.json 2> $O/vis_wearly.err
timeout 120 python bench.py $V > $O/vis_release2.json 2> $O/vis_release2.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3d/vis_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        fam = d.get("families", {})
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"],
              {k: round(v["ms_per_step_est"], 2) for k, v in fam.items() if k in ("gemm", "vit_attn")})
    except Exception as e:
        print(f, "ERR", e)
PY
