#!/bin/bash
# round 5, session b: (1) ViT attention A/B (8 waves at 336 px, lazy rescale, exact last chunk) on the microbenchmark and the 336-px bench;
# (2) config 5 (13B fp8) with the 8-phase producers at 10 row blocks per workgroup / fused finish / o_proj in that form; (3) wide batches with the fused finish.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
python - > $O/build.log 2>&1 <<'PY'
from video_llava_amd import build
print(build.build()); print(build.build(lab=True))
print(build.build_variant("attn_nolazy", ["PGV_LAB", "PGV_LAB_ATTN_LAZY_TH=-1.0f"]))
PY
tail -2 $O/build.log
LAB=video_llava_amd/libpgv_lab.so
W="python scripts/lab/with_lib.py"
echo "== attention microbench ==" > $O/attn.txt
for v in "new:$LAB:" "nw4:$LAB:PGV_ATTN_NW4=1" "nolazy:video_llava_amd/libpgv_attn_nolazy.so:" "nolazy_nw4:video_llava_amd/libpgv_attn_nolazy.so:PGV_ATTN_NW4=1"; do
  IFS=: read name lib env <<< "$v"
  echo "--- $name" >> $O/attn.txt
  env $env timeout 120 $W $lib scripts/microbench.py attn >> $O/attn.txt 2>&1
done
cat $O/attn.txt
# parity of the new attention kernel before anything else is believed
timeout 600 python -m pytest tests/test_gpu_vision.py -m gpu -q -x > $O/pytest_vision.log 2>&1; tail -3 $O/pytest_vision.log
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames"
timeout 300 python $B --image 336 > $O/bench_336.json 2> $O/bench_336.err; python scripts/sessions/r5_pick.py $O/bench_336.json
PGV_ATTN_NW4=1 timeout 300 $W $LAB $B --image 336 > $O/bench_336_nw4.json 2> $O/bench_336_nw4.err; python scripts/sessions/r5_pick.py $O/bench_336_nw4.json
# config 5
C5="$B --llm 13b --weights fp8"
i=0
for env in "" "PGV_GEMV_K8_NWB10=1" "PGV_GEMV_K8_FUSED=1" "PGV_GEMV_K8_NWB10=1 PGV_GEMV_K8_FUSED=1" "PGV_GEMV_K8_NWB10=1 PGV_GEMV_K8_FUSED=1 PGV_GEMV_K8_NARROW_MINK=4096" "PGV_GEMV_K8_NWB10=1 PGV_GEMV_K8_NARROW_MINK=4096"; do
  i=$((i+1))
  env $env timeout 300 $W $LAB $C5 > $O/cfg5_$i.json 2> $O/cfg5_$i.err
  echo "cfg5 [$env]"; python scripts/sessions/r5_pick.py $O/cfg5_$i.json
done
# wide batches, 7B bf16
for n in 32 64; do
  for env in "" "PGV_GEMV_K8_FUSED=1"; do
    f=$O/clips${n}_$(echo "$env" | tr -c 'A-Z0-9\n' '_').json
    env $env timeout 300 $W $LAB $B --steps 1 --clips-per-gpu $n > $f 2> $f.err
    echo "clips $n [$env]"; python scripts/sessions/r5_pick.py $f
  done
done
