#!/usr/bin/env python3
"""Where the GPU idles inside a bench step: union of the kernel intervals of a rocprofv3 kernel trace (all streams), then every idle gap between
`lo` and `hi` microseconds with the kernels on either side, grouped by (previous -> next).  Usage: trace_idle.py <results.db> [lo_us=15] [hi_us=20000]"""
import collections
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_gaps import short  # noqa: E402


def main(db, lo=15.0, hi=20000.0):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    ks = [(short(n), s, e) for n, s, e in rows]
    gaps = collections.defaultdict(list)
    cur_end, cur_name = ks[0][2], ks[0][0]
    busy = 0
    seg_start = ks[0][1]
    for name, s, e in ks[1:]:
        if s > cur_end:
            busy += cur_end - seg_start
            g = (s - cur_end) / 1e3
            if lo <= g <= hi:
                gaps[(cur_name, name)].append(g)
            seg_start = s
            cur_end, cur_name = e, name
        elif e > cur_end:
            cur_end, cur_name = e, name
    busy += cur_end - seg_start
    span = ks[-1][2] - ks[0][1]
    tot = sum(sum(v) for v in gaps.values())
    print(f"trace span {span / 1e6:.1f} ms, GPU busy (union over streams) {busy / 1e6:.1f} ms; idle gaps of {lo:g}..{hi:g} us: {tot / 1e3:.2f} ms in {sum(len(v) for v in gaps.values())} gaps")
    print(f"{'previous -> next':70s} {'n':>6s} {'total ms':>9s} {'avg us':>8s} {'max us':>8s}")
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print(f"{(k[0] + ' -> ' + k[1])[:70]:70s} {len(v):6d} {sum(v) / 1e3:9.3f} {sum(v) / len(v):8.1f} {max(v):8.1f}")


if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:4]))
