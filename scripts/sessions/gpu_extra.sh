#!/bin/bash
# Side measurements for DESIGN.md: PCIe-inclusive rate, 16 resident clips per GPU, fp8-weight decoders.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/extra
run() { name=$1; shift; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/extra/$name.json 2> gpurun_out/extra/$name.err; tail -c 200 gpurun_out/extra/$name.err; }
run host_frames --host-frames
run clips16 --clips-per-gpu 16
run 7b_fp8 --weights fp8
run 13b_fp8 --llm 13b --weights fp8
for f in gpurun_out/extra/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], round(d["value"], 3), "videos/s", round(d["ms_per_step"], 1), "ms/step", d.get("pcie_inclusive"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
