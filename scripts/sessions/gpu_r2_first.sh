#!/bin/bash
# Round-2 first GPU trip: host facts, the whole GPU suite (no -x: collect every failure), smoke, one bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
{ nproc; free -g; python -c "import torch,os;print('threads',torch.get_num_threads(),'cpus',os.cpu_count())"; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $O/host.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q -s --durations=15 > $O/pytest.log 2>&1
tail -40 $O/pytest.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err; head -c 3000 $O/bench.json
