#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; scripts/pmc_traffic.sh).

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts a wide
coalesced streaming read (16 B/lane, plain or LDS-DMA) at exactly half its bytes (128-B requests tallied at 64 B), so the corrected
read traffic is 2x the raw value; WRITE_SIZE is uncalibrated there, so this script prints a calibration against kernels whose
algorithmic byte count is known exactly (decode GEMVs read N*K*2 weight bytes once; the fp32->16-bit cast writes M*C*2 bytes)
and applies the read correction only.  Usage: pmc_summary.py <fetch.db> <write.db> [out.json]"""
import json
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::|_GLOBAL__N_1\d*)(\w+?)(?:<|\(|I[0-9A-Z])", name)
    base = m.group(1) if m else name.split("(")[0][-60:]
    t = re.search(r"<(TBF16|TF16)(?:, (\d+))?", name)
    if t:
        base += f"<{t.group(1)}" + (f",{t.group(2)}" if t.group(2) else "") + ">"
    return base


def load(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, val, dur in c.execute("select kernel_name, value, (end - start) from counters_collection where counter_name = ?", (counter,)):
        e = out.setdefault(short(name), {"calls": 0, "kib": 0.0, "ns": 0.0})
        e["calls"] += 1; e["kib"] += val; e["ns"] += dur
    return out


def main(fetch_db, write_db, out=None):
    f, w = load(fetch_db, "FETCH_SIZE"), load(write_db, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, {"ns": 0})["ns"])):
        fe, we = f.get(k), w.get(k)
        calls = (fe or we)["calls"]
        rows[k] = {"calls": calls,
                   "avg_us_under_pmc": (fe or we)["ns"] / calls / 1e3,
                   "fetch_bytes_raw": fe["kib"] * 1024 / fe["calls"] if fe else None,
                   "fetch_bytes_corrected_x2": 2 * fe["kib"] * 1024 / fe["calls"] if fe else None,
                   "write_bytes_raw": we["kib"] * 1024 / we["calls"] if we else None}
    print(f"{'kernel':42s} {'calls':>6s} {'avg_us':>8s} {'fetch raw MB':>13s} {'fetch x2 MB':>12s} {'write raw MB':>13s}")
    for k, r in list(rows.items())[:30]:
        fm = lambda x: f"{x / 1e6:13.3f}" if x is not None else f"{'-':>13s}"
        print(f"{k:42s} {r['calls']:6d} {r['avg_us_under_pmc']:8.1f} {fm(r['fetch_bytes_raw'])} {fm(r['fetch_bytes_corrected_x2'])[1:]} {fm(r['write_bytes_raw'])}")
    if out:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bytes per dispatch",
                   "correction": "fetch_bytes_corrected_x2 = 2 x FETCH_SIZE (MI355X_MICROARCH.md, gfx950: 128-B requests tallied at 64 B); WRITE_SIZE raw",
                   "kernels": rows}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
