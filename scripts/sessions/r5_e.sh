#!/bin/bash
# round 5, session e: GPU tests under the sanitizer build of the host shim (UBSan non-recoverable + libstdc++ container assertions on the host code of
# libpgv, release kernels) after the round's host-side changes (GEMV dispatch table, forward / all-position logits, loader), + the runner's --batch auto on the GPU.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r5e}; mkdir -p $O
python -c "from video_llava_amd import build; print(build.build())" > $O/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_runners.py -m gpu -q -x -k "qa_runner or rccl" > $O/pytest_runner.log 2>&1; echo "runner tests rc=$?"; tail -3 $O/pytest_runner.log | cut -c1-200
RT=$(python -c "from video_llava_amd import build; print(build.sanitizer_runtime())")
[ -f video_llava_amd/libpgv_ubsan.so ] || python -c "from video_llava_amd import build; build.build_sanitizer()"
export UBSAN_OPTIONS=print_stacktrace=1
LD_PRELOAD=$RT timeout 1500 python scripts/lab/with_lib.py video_llava_amd/libpgv_ubsan.so -m pytest tests/test_gpu_loader.py tests/test_gpu_llm.py tests/test_gpu_runners.py tests/test_gpu_sampling.py tests/test_gpu_vision.py -q -x \
    -k "not 800_frames and not w_resident and not switches and not 8_phase_form and not wide_batch_invariance and not full_7b and not rccl and not production_shape" > $O/pytest_ubsan.log 2>&1
echo "ubsan tests rc=$?"; tail -4 $O/pytest_ubsan.log | cut -c1-200
echo "reports: $(grep -c 'runtime error' $O/pytest_ubsan.log)"; grep -m5 "runtime error" $O/pytest_ubsan.log | cut -c1-300
