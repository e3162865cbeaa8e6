#!/bin/bash
# round 5, session d: gate/up with two (gate, up) pairs per workgroup at four column tiles (64 clips), A/B in one session + the bitwise tests of the GEMV family.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
python -c "from video_llava_amd import build; print(build.build()); print(build.build(lab=True))" > $O/build.log 2>&1; tail -2 $O/build.log
( time timeout 600 python -m pytest tests/test_gpu_llm.py -m gpu -q -x -k "gemv or batch or wide or tile" ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames"
W="python scripts/lab/with_lib.py video_llava_amd/libpgv_lab.so"
for n in 64 48; do
  for env in "PGV_GEMV_WIDE_TL4=1" "PGV_GEMV_WIDE_TL4=0"; do
    f=$O/clips${n}_$(echo "$env" | tr -c 'A-Z0-9\n' '_').json
    env $env timeout 300 $W $B --clips-per-gpu $n > $f 2> $f.err
    echo "clips $n [$env]"; python scripts/sessions/r5_pick.py $f
  done
done
timeout 300 python $B --clips-per-gpu 64 --weights fp8 > $O/clips64_fp8.json 2> $O/clips64_fp8.err; echo "clips 64 fp8"; python scripts/sessions/r5_pick.py $O/clips64_fp8.json
