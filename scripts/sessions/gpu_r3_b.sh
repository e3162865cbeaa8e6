#!/bin/bash
# Round 3 session b: pipelined GEMM epilogue -- parity (vision tests), A/B against the round-2 epilogue (libpgv_nopipe.so) and the
# start-stagger experiment (libpgv_lab.so, PGV_GEMM_STAGGER in 10 ns ticks), vision-only bench (BASELINE configs[1]).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vision.py -x -q > $O/pytest_vision.log 2>&1; echo "vision tests rc=$?" | tee -a $O/pytest_vision.log
V="--workload vision --steps 10 --warmup 3 --no-host-frames"
for rep in 1 2; do
  timeout 120 python bench.py $V > $O/vis_pipe_$rep.json 2> $O/vis_pipe_$rep.err
  timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_nopipe.so bench.py $V > $O/vis_nopipe_$rep.json 2> $O/vis_nopipe_$rep.err
done
for st in 0 600 1100 2800; do
  PGV_GEMM_STAGGER=$st timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_lab.so bench.py $V > $O/vis_lab_stagger_$st.json 2> $O/vis_lab_stagger_$st.err
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o vis -- python $GRAFT_REPO_ROOT/bench.py --workload vision --steps 3 --warmup 1 --no-host-frames --no-profile-pass > $GRAFT_REPO_ROOT/$O/vis_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT
ls $O/prof | head; DB=$(ls $O/prof/*results.db $O/prof/*/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocprof_summary.py "$DB" > $O/vis_kernel_trace.txt 2>&1
rm -rf $O/prof
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3b/vis_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        fam = d.get("families", {})
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"],
              {k: round(v["ms_per_step_est"], 2) for k, v in fam.items() if k in ("gemm", "vit_attn")})
    except Exception as e:
        print(f, "ERR", e)
PY
