#!/bin/bash
# round 4, session c: fp8 GEMV lab (x-operand cost), fixed runner test, the pinned CPU-baseline child alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_runners.py -q -x > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest.log
export LD_LIBRARY_PATH=$PWD/video_llava_amd:$LD_LIBRARY_PATH
timeout 300 scripts/lab/gemv8_chain.exe > $O/gemv8_chain.log 2>&1; cat $O/gemv8_chain.log
python - <<'PY' > $O/cpu_child.log 2>&1
import json, os, subprocess, sys, time
sys.path.insert(0, ".")
import bench
node, cpus = bench._numa_node_cpus()
print("node", node, "phys cpus", len(cpus), cpus[:4], "...", "logical", os.cpu_count())
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cpu.max n/a", e)
class A: llm="7b"; frames=100; new_tokens=256
t=time.time(); r=bench.cpu_baseline_reference(A); print("wall", time.time()-t)
print(json.dumps({k: r.get(k) for k in ("value","cores","threads_tried","threads_used","config1_by_threads_s","prefill_by_threads_s","decode_step_by_threads_s","decode_step_s_per_layer","decode_per_layer_vs_survey_probe","pinning","child_wall_s")}))
PY
cat $O/cpu_child.log
