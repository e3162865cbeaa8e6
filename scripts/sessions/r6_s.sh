#!/bin/bash
# Session r6_s: why is gate/up with four pairs per workgroup slow at four column tiles (13B, 48 / 64 sequences)?  Ablations + kernel trace.
O=gpurun_out/r6_s; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
export PGV_LIB=lab PGV_WIDE_13B=1 PGV_GEMV_XBLK=1
for a in 1 2 4 3 6; do PGV_GEMV_ABLATE=$a python scripts/microbench.py gemvwide 2>&1 | grep "gate/up13.*B=\(32\|64\)" | sed "s/^/abl$a /" >> $O/abl.txt; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/scripts/microbench.py gemvwide > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat $O/abl.txt
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r6_s/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:110], r["Calls"], r["AverageNs"])
PY
