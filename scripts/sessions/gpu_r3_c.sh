#!/bin/bash
# Round 3 session c: where the ViT GEMM time goes (epilogue vs main loop, lab library) + the batched / prefetching eval runner.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
for abl in 0 32 1 4; do
  PGV_GEMM_ABLATE=$abl timeout 200 python scripts/lab/gemm_epi_decomp.py >> $O/gemm_epi_decomp.txt 2>> $O/gemm_epi_decomp.err
done
cat $O/gemm_epi_decomp.txt
timeout 600 python -m pytest tests/test_gpu_runners.py tests/test_gpu_loader.py -x -q > $O/pytest_runners.log 2>&1; echo "runner tests rc=$?" | tee -a $O/pytest_runners.log
tail -5 $O/pytest_runners.log
