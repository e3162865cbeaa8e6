#!/bin/bash
# round 4, session g: activation row stride (L2 channel conflicts of the x operand?) in the fp8 chain lab
cd "$GRAFT_REPO_ROOT" || exit 1
export LD_LIBRARY_PATH=$PWD/video_llava_amd:$LD_LIBRARY_PATH
O=gpurun_out/r4g; mkdir -p $O
for xp in 0 64 128 192 264; do
echo "== XPAD=$xp =="; XPAD=$xp PGV_GEMV_ROT=0 timeout 200 scripts/lab/gemv8_chain.exe 2>&1 | grep -v "bare\|B=1 " | tee $O/chain8_xpad$xp.log
done
