#!/bin/bash
# HBM traffic of every kernel of a short bench run from the rocprofv3 PMC counters (MI355X_MICROARCH.md "HBM" section):
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), --kernel-trace only.
# Output: gpurun_out/pmc/{fetch,write}/..._results.db  ->  scripts/pmc_summary.py -> profiles/<name>.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
ARGS="--steps 1 --warmup 0 --new-tokens 9 --no-cpu-baseline --no-profile-pass --no-host-frames --no-latency --no-runner --no-side"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/fetch -o pmc -- python bench.py $ARGS > gpurun_out/pmc/fetch.json 2> gpurun_out/pmc/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/write -o pmc -- python bench.py $ARGS > gpurun_out/pmc/write.json 2> gpurun_out/pmc/write.err
ls -la gpurun_out/pmc/fetch gpurun_out/pmc/write
python scripts/pmc_summary.py gpurun_out/pmc/fetch/pmc_results.db gpurun_out/pmc/write/pmc_results.db gpurun_out/pmc/pmc_traffic.json | head -30
rm -rf gpurun_out/pmc/fetch gpurun_out/pmc/write
