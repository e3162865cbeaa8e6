#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite `*_results.db`) into a per-kernel table
(count, total ms, share, avg/min/max us) -- the same figures `--stats` prints, in a diff-able text file for profiles/."""
import sqlite3
import sys


def main(db, out=None, top=40):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 kernel-trace summary of {db}", f"# total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'total_ms':>10} {'share':>6} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>10}  kernel"]
    for r in rows[:top]:
        lines.append(f"{r[2]:10.2f} {100 * r[2] / tot:5.1f}% {r[1]:7d} {r[3]:10.1f} {r[4]:9.1f} {r[5]:10.1f}  {r[0][:140]}")
    # kernels whose launches fall into two clearly separate duration clusters (one kernel, two shapes: out_proj / fc2 share gemm_w4<.,10>)
    for r in rows[:top]:
        if r[1] >= 4 and r[5] > 1.8 * r[4]:
            d = [x[0] / 1e3 for x in c.execute("select end-start from kernels where name = ? order by 1", (r[0],))]
            gaps = [(d[i + 1] / d[i], i) for i in range(len(d) - 1)]
            ratio, cut = max(gaps)
            if ratio > 1.4:
                lo, hi = d[:cut + 1], d[cut + 1:]
                lines.append(f"#   split {r[0][:70]}: {len(lo)} launches avg {sum(lo) / len(lo):.1f} us | {len(hi)} launches avg {sum(hi) / len(hi):.1f} us")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
