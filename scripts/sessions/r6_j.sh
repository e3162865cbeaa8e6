#!/bin/bash
# Round-6 session J: ViT attention, wave arbitration experiments (s_setprio around the MFMA clusters, static priority for the younger half,
# a start offset for the upper half of the waves), variant libraries, same box, two rounds.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6j}; mkdir -p $O
for rep in 1 2; do
for v in libpgv attn_prio1 attn_prio2 attn_prio3 attn_stag8 attn_stag14 attn_stag22; do
  lib=video_llava_amd/libpgv_$v.so; [ $v = libpgv ] && lib=video_llava_amd/libpgv.so
  timeout 200 python scripts/lab/with_lib.py $lib scripts/microbench.py attn > $O/attn_${v}_$rep.txt 2>&1
  echo "--- $v ($rep)"; grep "N=" $O/attn_${v}_$rep.txt
done
done
