#!/bin/bash
# Vision-side GPU check: vision tests + bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/vision
timeout 900 python -m pytest tests/test_gpu_vision.py tests/test_gpu_runners.py -x -q > gpurun_out/vision/pytest.log 2>&1
tail -3 gpurun_out/vision/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/vision/bench.json 2> gpurun_out/vision/bench.err
timeout 600 python bench.py --workload vision --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/vision/bench_vision.json 2> gpurun_out/vision/bench_vision.err
tail -c 300 gpurun_out/vision/bench_vision.err
