#!/bin/bash
# round 4: GPU tests under the sanitizer build of the host shim (UBSan, non-recoverable, + libstdc++ container assertions on the host code of libpgv; release kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r4ubsan}; mkdir -p $O
RT=$(python -c "from video_llava_amd import build; print(build.sanitizer_runtime())")
[ -f video_llava_amd/libpgv_ubsan.so ] || python -c "from video_llava_amd import build; build.build_sanitizer()"
export UBSAN_OPTIONS=print_stacktrace=1
LD_PRELOAD=$RT timeout 1500 python scripts/lab/with_lib.py video_llava_amd/libpgv_ubsan.so -m pytest tests/test_gpu_loader.py tests/test_gpu_llm.py tests/test_gpu_runners.py tests/test_gpu_sampling.py tests/test_gpu_vision.py -q -x \
    -k "not 800_frames and not w_resident and not switches and not fp8_mfma_decode and not wide_batch_invariance and not full_7b" > $O/pytest_ubsan.log 2>&1
echo "ubsan tests rc=$?"; tail -4 $O/pytest_ubsan.log | cut -c1-200
echo "reports: $(grep -c 'runtime error' $O/pytest_ubsan.log)"; grep -m5 "runtime error" $O/pytest_ubsan.log | cut -c1-300
