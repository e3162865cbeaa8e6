#!/bin/bash
# round 5, session a: the restructured library (ABI 300) under the whole GPU suite, with the new end-to-end cases in data-gathering mode
# (loose pins, prefix 0: the numbers printed here set the pins), then the 8-rank shape of the N > 1 path on one shared device.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
python -c "from video_llava_amd import build; print(build.build()); print(build.build(lab=True))" > $O/build.log 2>&1
( time PGV_E2E_PREFIX=0 PGV_E2E_PIN=1.0 timeout 900 python -m pytest tests/test_gpu_fulldepth.py -m gpu -q -s --durations=15 ) > $O/fulldepth.log 2>&1
tail -5 $O/fulldepth.log
( time timeout 900 python -m pytest tests -m gpu -q --ignore=tests/test_gpu_fulldepth.py --durations=15 ) > $O/pytest_rest.log 2>&1
tail -30 $O/pytest_rest.log
( time PGV_BENCH_SHARE_DEVICE=1 PGV_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames --no-profile-pass ) > $O/gpus8_shared.log 2>&1
tail -c 1500 $O/gpus8_shared.log
