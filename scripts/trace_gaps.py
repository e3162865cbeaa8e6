#!/usr/bin/env python3
"""Inter-kernel gaps in a rocprofv3 kernel trace: for the longest run of consecutive decode-step kernels (graph replays), the share of the span
covered by kernel execution and the gap statistics by (previous kernel -> next kernel).  Usage: trace_gaps.py <results.db>"""
import collections
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::|_GLOBAL__N_1\d*)(\w+?)(?:<|\(|I[0-9A-Z])", name)
    base = m.group(1) if m else name.split("(")[0][-40:]
    t = re.search(r"<(TBF16|TF16)(?:, (\d+))?", name)
    if t and t.group(2):
        base += f"<{t.group(2)}>"
    return base


def main(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    ks = [(short(n), s, e) for n, s, e in rows]
    dec = {"gemv_mfma_kernel", "decode_attn_kernel", "rms_resid_kernel", "argmax_kernel", "embed_tok_kernel"}
    best = (0, 0, 0)
    i = 0
    while i < len(ks):
        j = i
        while j < len(ks) and ks[j][0].split("<")[0] in dec:
            j += 1
        if j - i > best[0]:
            best = (j - i, i, j)
        i = max(j, i + 1)
    n, i, j = best
    if n < 10:
        print("no decode run found"); return
    span = ks[j - 1][2] - ks[i][1]
    busy = sum(e - s for _, s, e in ks[i:j])
    print(f"decode run: {n} kernels, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms = {100.0 * busy / span:.1f} % of the span")
    gaps = collections.defaultdict(list)
    for a, b in zip(ks[i:j - 1], ks[i + 1:j]):
        gaps[(a[0], b[0])].append(b[1] - a[2])
    print(f"{'previous -> next':60s} {'n':>6s} {'avg gap us':>10s} {'min':>7s} {'max':>7s}")
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0] + ' -> ' + k[1]:60s} {len(v):6d} {sum(v) / len(v) / 1e3:10.2f} {min(v) / 1e3:7.2f} {max(v) / 1e3:7.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
