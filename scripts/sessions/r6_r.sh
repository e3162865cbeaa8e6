#!/bin/bash
# Session r6_r: the 13B launch shapes of the wide decode batches (qkv with four row blocks, gate/up with four pairs per workgroup), A/B by the lab
# switches, then the whole 13B job at 32 and 64 clips per GPU.
O=gpurun_out/r6_r; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
export PGV_LIB=lab PGV_WIDE_13B=1 PGV_GEMV_XBLK=1
python scripts/microbench.py gemvwide > $O/wide13_new.txt 2>&1
PGV_GEMV_WIDE_QKV4=0 PGV_GEMV_WIDE_TL8=0 python scripts/microbench.py gemvwide > $O/wide13_old.txt 2>&1
unset PGV_LIB PGV_WIDE_13B PGV_GEMV_XBLK
Q="--no-side --no-latency --no-cpu-baseline --no-runner"
timeout 900 python bench.py --llm 13b --steps 2 --warmup 1 $Q --clips-per-gpu 64 > $O/bench13_clips64.json 2> $O/bench13_clips64.err
timeout 900 python bench.py --llm 13b --steps 2 --warmup 1 $Q --clips-per-gpu 32 > $O/bench13_clips32.json 2> $O/bench13_clips32.err
tail -n 12 $O/wide13_new.txt $O/wide13_old.txt
cut -c1-400 $O/bench13_clips64.json $O/bench13_clips32.json
