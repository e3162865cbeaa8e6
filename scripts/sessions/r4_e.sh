#!/bin/bash
# round 4, session e: fp8 GEMV variants (consumer TL, producer K-split) in the chain lab + parity of the forced variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/video_llava_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_gpu_llm.py -q -x -k "switches or batch_invariance or fp8 or 13b or 7b" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
echo "== default (prod TL4/KS4, gate/up pairs 2, qkv TL3) =="; timeout 200 scripts/lab/gemv8_chain.exe 2>&1 | grep -v "B=4" | tee $O/chain_default.log
echo "== PGV_GEMV_PROD_TL=1 PGV_GEMV_GU_PAIRS=1 PGV_GEMV_TL3=0 (round-3 shapes) =="; PGV_GEMV_PROD_TL=1 PGV_GEMV_GU_PAIRS=1 PGV_GEMV_TL3=0 timeout 200 scripts/lab/gemv8_chain.exe 2>&1 | grep -v "bare" | tee $O/chain_r3.log
echo "== PGV_GEMV_PROD_TL=2 =="; PGV_GEMV_PROD_TL=2 timeout 200 scripts/lab/gemv8_chain.exe 2>&1 | grep "producer" | tee $O/chain_tl2.log
