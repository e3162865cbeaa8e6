#!/bin/bash
# round 5, session n: split decode attention with three workgroups per CU (80 VGPRs: the fresh token's block after the key loop) against two (lab variant, 5 waves per SIMD).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
python -c "
from video_llava_amd import build; print(build.build()); print(build.build(lab=True)); print(build.build_variant('dattn_w5', ['PGV_LAB', 'PGV_LAB_DATTN_WAVES_PER_EU=5']))" > $O/build.log 2>&1; tail -1 $O/build.log
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames --llm 13b --weights fp8"
for v in "w6:video_llava_amd/libpgv_lab.so" "w5:video_llava_amd/libpgv_dattn_w5.so" "w6:video_llava_amd/libpgv_lab.so" "w5:video_llava_amd/libpgv_dattn_w5.so"; do
  IFS=: read name lib <<< "$v"
  f=$O/cfg5_${name}_$RANDOM.json
  timeout 300 python scripts/lab/with_lib.py $lib $B > $f 2> $f.err
  echo "cfg5 [$name]"; python scripts/sessions/r5_pick.py $f
done
( time timeout 600 python -m pytest tests/test_gpu_llm.py -m gpu -q ) > $O/pytest_llm.log 2>&1; tail -3 $O/pytest_llm.log | cut -c1-200
