#!/bin/bash
# round 5, session m: evidence only -- kernel trace of the 64-clip configuration (the runners' default regime with --batch auto), and clocks / power over a whole
# headline step (vision at the power cap vs the HBM-bound decode).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 1 --warmup 1 --clips-per-gpu 64 $Q > $O/bench_clips64_under_rocprof.json 2> $O/rocprof.err
python scripts/rocprof_summary.py $O/prof/bench_results.db > $O/kernel_trace_clips64.txt 2>&1; rm -rf $O/prof
head -24 $O/kernel_trace_clips64.txt | cut -c1-170
( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null >> $O/smi_step.jsonl; echo >> $O/smi_step.jsonl; sleep 0.1; done ) &
SMI=$!
timeout 300 python bench.py --steps 12 --warmup 2 $Q --no-profile-pass > $O/bench_smi.json 2> $O/bench_smi.err
kill $SMI; wait $SMI 2>/dev/null
python - $O/smi_step.jsonl <<'PY'
import json, re, sys
rows = []
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("{"):
        try: rows.append(next(iter(json.loads(l).values())))
        except Exception: pass
def val(r, key):
    for k, v in r.items():
        if key in k.lower():
            m = re.search(r"([0-9.]+)", str(v)); return float(m.group(1)) if m else None
pts = [(val(r, "sclk clock speed"), val(r, "power")) for r in rows]
pts = [p for p in pts if p[0] and p[1]]
hi = [p for p in pts if p[1] > 1200]; mid = [p for p in pts if 500 < p[1] <= 1200]
for name, g in (("power > 1200 W (vision / prefill)", hi), ("500 - 1200 W (decode)", mid)):
    if g: print(name, "samples", len(g), "sclk mean %.0f MHz, power mean %.0f W" % (sum(a for a, _ in g) / len(g), sum(b for _, b in g) / len(g)))
print("all samples", len(pts), "power histogram (100 W bins):", {int(b // 100 * 100): sum(1 for _, p in pts if p // 100 == b // 100) for b in sorted({p // 100 * 100 for _, p in pts})})
PY
