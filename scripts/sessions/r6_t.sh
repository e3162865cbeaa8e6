#!/bin/bash
# Session r6_t: 13B wide decode batches, release kernels: the committed r06_z library (video_llava_amd/libpgv_r06z.so, build_variant from the
# r06_z sources) against this tree (qkv with four row blocks, gate/up with four pairs per workgroup).  (The lab library spills in these shapes --
# 456 B of scratch with the ablation code compiled in -- so scripts/microbench.py does not measure them; r6_s.)
O=gpurun_out/r6_t; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
Q="--llm 13b --steps 2 --warmup 1 --no-side --no-latency --no-cpu-baseline --no-runner"
for c in 64 32 24; do
  timeout 900 python bench.py $Q --clips-per-gpu $c > $O/new_$c.json 2> $O/new_$c.err
  timeout 900 python scripts/lab/with_lib.py video_llava_amd/libpgv_r06z.so bench.py $Q --clips-per-gpu $c > $O/old_$c.json 2> $O/old_$c.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_t/*.json")):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], round(j["value"], 3), round(j["ms_per_step"], 1), {k: j.get(k) for k in ("token_check",)}, j.get("roofline_gemv", j.get("roofline")))
PY
