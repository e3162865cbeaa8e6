#!/usr/bin/env python3
"""Per-kernel SQ counter table from the passes of scripts/pmc_sq.sh.  Usage: pmc_sq_summary.py <pass1.db> <pass2.db> ...
Values are per-dispatch averages summed over the chip as rocprofv3 reports them; ratios are what to read:
  lds_conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of LDS-active cycles lost to bank conflicts)
  mfma     = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES-per-CU estimate) is not derivable without the CU count per SE,
             so the table prints MFMA-busy cycles per wave cycle instead (mfma/wave) next to wait/wave = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import collections
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::|_GLOBAL__N_1\d*)(\w+?)(?:<|\(|I[0-9A-Z])", name)
    base = m.group(1) if m else name.split("(")[0][-50:]
    t = re.search(r"<(TBF16|TF16)(?:, (\d+))?", name)
    if t:
        base += f"<{t.group(1)}" + (f",{t.group(2)}" if t.group(2) else "") + ">"
    return base


def main(dbs):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for db in dbs:
        c = sqlite3.connect(db)
        seen = set()
        for name, cn, val, did, st, en in c.execute("select kernel_name, counter_name, value, dispatch_id, start, end from counters_collection"):
            k = short(name)
            a = agg[k][cn]; a[0] += 1; a[1] += val
            if (db, did) not in seen:
                seen.add((db, did)); d = dur[k]; d[0] += 1; d[1] += en - st
    rows = []
    for k, cs in agg.items():
        g = lambda n: (cs[n][1] / cs[n][0]) if n in cs and cs[n][0] else float("nan")
        rows.append((dur[k][1], k, dur[k][0], dur[k][1] / max(dur[k][0], 1) / 1e3, g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
                     g("SQ_LDS_IDX_ACTIVE") / max(g("SQ_BUSY_CYCLES"), 1.0), g("SQ_VALU_MFMA_BUSY_CYCLES") / max(g("SQ_WAVE_CYCLES"), 1.0),
                     g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0), g("SQ_WAIT_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1.0),
                     g("SQ_INSTS_VALU") / max(g("SQ_INSTS_MFMA"), 1.0), g("SQ_LDS_UNALIGNED_STALL")))
    print(f"{'kernel':34s} {'calls':>6s} {'avg_us':>9s} {'lds_conf':>8s} {'lds/busy':>8s} {'mfma/wave':>9s} {'wait/wave':>9s} {'ldswait':>8s} {'valu/mfma':>9s} {'unalign':>8s}")
    for r in sorted(rows, reverse=True)[:24]:
        print(f"{r[1][:34]:34s} {r[2]:6d} {r[3]:9.1f} {r[4]:8.3f} {r[5]:8.3f} {r[6]:9.3f} {r[7]:9.3f} {r[8]:8.3f} {r[9]:9.1f} {r[10]:8.0f}")


if __name__ == "__main__":
    main(sys.argv[1:])
