#!/bin/bash
# Round-6 session B: (1) which rows differ between one-call and cut prefill; (2) what the wide-batch consumer GEMVs spend their time on
# (lab library, PGV_GEMV_ABLATE: 1 no x loads, 2 no MFMA, 4 no weight loads); (3) the 8-phase producers with the x slice staged by LDS-DMA.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6b}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
timeout 300 python scripts/lab/append_debug.py > $O/append_debug.txt 2>&1; cat $O/append_debug.txt | tail -40
for a in 0 1 2 4 3 6; do
  PGV_LIB=lab PGV_GEMV_ABLATE=$a timeout 300 python scripts/microbench.py gemvwide > $O/gemvwide_abl$a.txt 2>&1
  echo "--- ablate $a"; cat $O/gemvwide_abl$a.txt | grep -v "^$"
done
timeout 900 python -m pytest tests/test_gpu_llm.py -q -x -k "wide or 8_phase or k8 or batch or gemv" > $O/pytest_gemv.log 2>&1; tail -5 $O/pytest_gemv.log
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 600 python bench.py --steps 2 --warmup 1 $Q --clips-per-gpu 64 > $O/bench_clips64.json 2> $O/bench_clips64.err
timeout 600 python bench.py --steps 2 --warmup 1 $Q --clips-per-gpu 32 > $O/bench_clips32.json 2> $O/bench_clips32.err
for f in bench_clips64 bench_clips32; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), (d.get("roofline") or {}).get("frac"), {k: round(v["avg_us"],1) for k,v in d["families"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
