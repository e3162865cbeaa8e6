#!/bin/bash
# round 4, session j: wide-batch GEMVs by shape (16-bit chain lab, B = 8 / 16 / 32 / 64) + the wide parity tests after the side-array fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/video_llava_amd:$LD_LIBRARY_PATH
O=gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -q -x -k "ragged or wide_batch" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.log | cut -c1-200
timeout 200 scripts/lab/gemv_chain.exe 2>&1 | grep -v "plain resid\|side loads\|LDS reduce\|consumer\|producer\|whole layer" | tee $O/chain16_wide.log
