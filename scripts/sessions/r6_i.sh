#!/bin/bash
# Round-6 session I: ViT attention A/B of the round-6 changes one by one (variant libraries built with -D switches), same box.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6i}; mkdir -p $O
for rep in 1 2; do
for v in libpgv attn_r5 attn_nosum2 attn_nosplit attn_nosum2_nosplit; do
  lib=video_llava_amd/libpgv_$v.so; [ $v = libpgv ] && lib=video_llava_amd/libpgv.so
  timeout 200 python scripts/lab/with_lib.py $lib scripts/microbench.py attn > $O/attn_${v}_$rep.txt 2>&1
  echo "--- $v ($rep)"; grep "N=" $O/attn_${v}_$rep.txt
done
done
