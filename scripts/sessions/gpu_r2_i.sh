#!/bin/bash
# Round-2 trip I: software-pipelined GEMV weight stream + merged x-fragment loads (PGV_GEMV_PIPE bits) -- bit identity, per-kernel durations, bench A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r2i; mkdir -p $O
for m in 0 45 61; do PGV_GEMV_PIPE=$m timeout 300 python scripts/lab/pipe_ab.py /tmp/ab_$m.npy 2>&1 | tail -1; done
cmp /tmp/ab_0.npy /tmp/ab_45.npy && echo "mask 45 bit-identical"
cmp /tmp/ab_0.npy /tmp/ab_61.npy && echo "mask 61 bit-identical"
for m in 45 61; do
  PGV_GEMV_PIPE=$m timeout 600 rocprofv3 --kernel-trace -d $O/prof$m -o t -- python bench.py --steps 1 --warmup 0 --new-tokens 65 --no-cpu-baseline --no-profile-pass --no-host-frames --no-latency > $O/bench_trace_$m.json 2> $O/err_$m.log
  python scripts/rocprof_summary.py $O/prof$m/t_results.db > $O/summary_$m.txt 2>&1
  rm -rf $O/prof$m
  echo "== mask $m"; grep -E "gemv_mfma_kernel|decode_attn" $O/summary_$m.txt | cut -c1-130
done
PGV_GEMV_PIPE=61 timeout 600 rocprofv3 --kernel-trace -d $O/prof8 -o t -- python bench.py --weights fp8 --steps 1 --warmup 0 --new-tokens 65 --no-cpu-baseline --no-profile-pass --no-host-frames --no-latency > $O/bench_trace_fp8.json 2> $O/err_fp8.log
python scripts/rocprof_summary.py $O/prof8/t_results.db > $O/summary_fp8.txt 2>&1; rm -rf $O/prof8
echo "== fp8 mask 61"; grep -E "gemv_mfma_kernel|decode_attn" $O/summary_fp8.txt | cut -c1-130
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-profile-pass"
for m in 0 13 29 61 0 61; do PGV_GEMV_PIPE=$m timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mask $m value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))"; done | tee $O/ab.txt
for m in 0 61; do PGV_GEMV_PIPE=$m timeout 600 $B --weights fp8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp8 mask $m value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))"; done | tee -a $O/ab.txt
