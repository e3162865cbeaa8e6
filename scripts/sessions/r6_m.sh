#!/bin/bash
# Round-6 session M: phase-split attention kernel, what the barriers and the one-segment skew cost (variant libraries), microbenchmark
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6m}; mkdir -p $O
for v in libpgv ps_nobar ps_noskew ps_nobar_noskew attn_nops; do
  lib=video_llava_amd/libpgv_$v.so; [ $v = libpgv ] && lib=video_llava_amd/libpgv.so
  timeout 200 python scripts/lab/with_lib.py $lib scripts/microbench.py attn > $O/attn_${v}.txt 2>&1
  echo "--- $v"; grep "N=577" $O/attn_${v}.txt
done
