#!/bin/bash
# Quick GPU check: LLM tests + decode GEMV microbench + bench line (no rocprof).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests/test_gpu_llm.py -x -q > gpurun_out/quick/pytest_llm.log 2>&1
tail -3 gpurun_out/quick/pytest_llm.log
timeout 300 python scripts/microbench.py gemv > gpurun_out/quick/gemv.log 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err
tail -c 300 gpurun_out/quick/bench.err
