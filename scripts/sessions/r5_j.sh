#!/bin/bash
# round 5, session j: is the fp16-vs-bf16 gap of the CLIP stage the clock?  Vision-only bench in both dtypes with rocm-smi sampled every 0.25 s.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for dt in bf16 fp16 bf16 fp16; do
  tag=${dt}_$RANDOM
  ( while true; do rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null >> $O/smi_$tag.jsonl; echo >> $O/smi_$tag.jsonl; sleep 0.25; done ) &
  SMI=$!
  timeout 300 python bench.py --workload vision --dtype $dt --steps 60 --warmup 5 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames --no-profile-pass > $O/bench_$tag.json 2> $O/bench_$tag.err
  kill $SMI; wait $SMI 2>/dev/null
  python - $O/bench_$tag.json $O/smi_$tag.jsonl $dt <<'PY'
import json, sys, re
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rows = []
for l in open(sys.argv[2]):
    l = l.strip()
    if not l.startswith("{"):
        continue
    try:
        j = json.loads(l)
    except Exception:
        continue
    card = next(iter(j.values()))
    rows.append(card)
def nums(key_part):
    out = []
    for r in rows:
        for k, v in r.items():
            if key_part in k.lower():
                m = re.search(r"([0-9.]+)", str(v))
                if m:
                    out.append(float(m.group(1)))
                break
    return out
keys = sorted(rows[0].keys()) if rows else []
s, p = nums("sclk"), nums("power")
print(sys.argv[3], "clip ms %.2f frac %.4f;" % (d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), "samples", len(rows),
      "sclk mean %.0f min %.0f max %.0f;" % (sum(s) / max(len(s), 1), min(s or [0]), max(s or [0])), "power mean %.0f max %.0f" % (sum(p) / max(len(p), 1), max(p or [0])))
if rows: print("   keys:", keys[:12])
PY
done
