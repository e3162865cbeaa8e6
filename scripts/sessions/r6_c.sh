#!/bin/bash
# Round-6 session C: the activation operand of the wide-batch consumer GEMVs -- row padding (L2 channel spread) and the fragment-blocked layout
# (one contiguous 1 KiB wave-load per MFMA B fragment), microbenchmark on the lab library; then the new parity tests that failed / were not reached in r6a.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6c}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
for pad in 0 64 128 32 520; do
  PGV_LIB=lab PGV_XPAD=$pad timeout 300 python scripts/microbench.py gemvwide > $O/gemvwide_pad$pad.txt 2>&1
  echo "--- pad $pad"; grep -E "B=(32|64)" $O/gemvwide_pad$pad.txt
done
PGV_LIB=lab PGV_GEMV_XBLK=1 timeout 300 python scripts/microbench.py gemvwide > $O/gemvwide_xblk.txt 2>&1; echo "--- xblk"; grep -v "^$" $O/gemvwide_xblk.txt
PGV_LIB=lab PGV_GEMV_XBLK=1 PGV_GEMV_ABLATE=4 timeout 300 python scripts/microbench.py gemvwide > $O/gemvwide_xblk_abl4.txt 2>&1; echo "--- xblk, no weight loads"; grep -E "B=(32|64)" $O/gemvwide_xblk_abl4.txt
( time timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_runners.py tests/test_gpu_sampling.py tests/test_gpu_loader.py tests/test_gpu_vision.py -q -s --durations=15 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -30 $O/pytest.log | cut -c1-250; tail -3 $O/pytest.time
grep -E "shaped 2-layer" $O/pytest.log | cut -c1-400
