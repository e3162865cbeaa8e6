#!/bin/bash
# Control-flow check of the N > 1 bench path on a 1-GPU box: two ranks share cuda:0, the collation goes over gloo.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/two
PGV_DIST_BACKEND=gloo PGV_BENCH_SHARE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 1 --warmup 1 --new-tokens 17 > gpurun_out/two/bench.json 2> gpurun_out/two/bench.err
echo "rc=$?"; tail -c 600 gpurun_out/two/bench.err | grep -v amdgpu.ids; cut -c1-400 gpurun_out/two/bench.json
