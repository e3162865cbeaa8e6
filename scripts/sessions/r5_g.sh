#!/bin/bash
# round 5, session g: 13B qkv GEMV with FOUR row blocks per workgroup (240 workgroups, one round) against three (320 = 1 1/4 rounds); fp8 and 16-bit weights.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
python -c "from video_llava_amd import build; print(build.build()); print(build.build(lab=True))" > $O/build.log 2>&1; tail -1 $O/build.log
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames"
W="python scripts/lab/with_lib.py video_llava_amd/libpgv_lab.so"
for cfg in "--llm 13b --weights fp8" "--llm 13b"; do
  for env in "PGV_GEMV_TL4=0" "PGV_GEMV_TL4=1" "PGV_GEMV_TL4=0" "PGV_GEMV_TL4=1"; do
    f=$O/$(echo "$cfg $env" | tr -c 'a-zA-Z0-9\n' '_')_$RANDOM.json
    env $env timeout 300 $W $B $cfg > $f 2> $f.err
    echo "[$cfg] [$env]"; python scripts/sessions/r5_pick.py $f
  done
done
PGV_GEMV_TL4=1 timeout 300 $W -m pytest tests/test_gpu_llm.py -q -x -k "13b or fp8 or batch_columns" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
