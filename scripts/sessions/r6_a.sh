#!/bin/bash
# Round-6 session A: the new parity cases (prefill append / forward with a cache, KV reuse across chat turns, 336-px production config end to end,
# context horizon 4096) + the whole GPU suite's wall clock, host-thread experiment for the oracle (cgroup quota vs torch's default thread count),
# and the refactored bench line (token_check, second-turn latency).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6a}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; free -g | head -2
python - > $O/threads.log 2>&1 <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
from oracle import synth, vision as ovis
ccfg = synth.CLIP_L14_224
cw = synth.quantize_weights(synth.make_clip_weights(ccfg, seed=0), "float16")
px = ovis.clip_preprocess(synth.make_frames(25, 224, seed=0))
print("default threads", torch.get_num_threads())
for n in (torch.get_num_threads(), 64, 32, 16, 8):
    torch.set_num_threads(n)
    t0 = time.time()
    with torch.no_grad():
        ovis.clip_select_features(px, cw, ccfg)
    print(n, "threads: 25 frames x 23 layers", round(time.time() - t0, 1), "s", flush=True)
PY
cat $O/threads.log
( time timeout 2400 python -m pytest tests -q -m gpu --durations=30 -x > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -45 $O/pytest.log | cut -c1-220; tail -3 $O/pytest.time
grep -E "^\[|shaped 2-layer|frames of" $O/pytest.log | cut -c1-400 > $O/pytest_prints.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
( time timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -c 600 $O/bench.err; tail -3 $O/bench.time
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), "roofline", d["roofline"]["frac"], "token_check", d["roofline"].get("token_check"))
print("latency", d.get("latency_b1"))
PY
