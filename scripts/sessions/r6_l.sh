#!/bin/bash
# Round-6 session L: phase-split ViT attention kernel (336 px) against the one-stream kernel (same source, -DPGV_LAB_ATTN_PS=0): parity tests,
# microbenchmark, vision-only bench at 336 px.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6l}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_vision.py -q -x --durations=3 -k "attention or 336" > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -12 $O/pytest.log | cut -c1-250
for rep in 1 2; do
for v in libpgv attn_nops; do
  lib=video_llava_amd/libpgv_$v.so; [ $v = libpgv ] && lib=video_llava_amd/libpgv.so
  timeout 200 python scripts/lab/with_lib.py $lib scripts/microbench.py attn > $O/attn_${v}_$rep.txt 2>&1
  echo "--- $v ($rep)"; grep "N=577" $O/attn_${v}_$rep.txt
done
done
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
for v in libpgv attn_nops; do
  lib=video_llava_amd/libpgv_$v.so; [ $v = libpgv ] && lib=video_llava_amd/libpgv.so
  timeout 600 python scripts/lab/with_lib.py $lib bench.py $Q --image 336 --workload vision --steps 6 --warmup 2 > $O/bench_vision336_$v.json 2> $O/bench_vision336_$v.err
done
for f in bench_vision336_libpgv bench_vision336_attn_nops; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f clip_ms %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), {k: (round(v["avg_us"],1), round(v.get("tflops",0),1)) for k,v in d["families"].items() if k in ("gemm","vit_attn")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
