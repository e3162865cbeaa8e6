#!/usr/bin/env python3
"""Which HOST calls sit inside the GPU's idle stretches of a bench step: rocprofv3 `--hip-trace --kernel-trace --output-format csv` ->
union of the kernel intervals (all streams) -> every idle gap of at least `lo` microseconds with the kernels on either side and the HIP API
calls that overlap it (longest first).  Usage: trace_idle_api.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv> [lo_us=300]"""
import csv
import glob
import os
import sys


def col(header, *cands):
    low = [h.lower() for h in header]
    for c in cands:
        for i, h in enumerate(low):
            if c in h:
                return i
    raise KeyError((cands, header))


def read(path, name_cands):
    with open(path, newline="") as f:
        r = csv.reader(f)
        header = next(r)
        ni, si, ei = col(header, *name_cands), col(header, "start"), col(header, "end")
        out = []
        for row in r:
            try:
                out.append((row[ni], int(row[si]), int(row[ei])))
            except (ValueError, IndexError):
                continue
    return out, header


def main(d, lo=300.0):
    kf = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    af = sorted(glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True))
    print("files:", kf, af)
    ks, kh = read(kf[-1], ("kernel_name",))
    api, ah = read(af[-1], ("function",))
    print("kernel columns:", kh)
    print("api columns:", ah)
    ks.sort(key=lambda t: t[1])
    api.sort(key=lambda t: t[1])
    t0 = ks[0][1]
    gaps = []
    cur_end, cur_name = ks[0][2], ks[0][0]
    for name, s, e in ks[1:]:
        if s > cur_end:
            if (s - cur_end) / 1e3 >= lo:
                gaps.append((cur_end, s, cur_name, name))
            cur_end, cur_name = e, name
        elif e > cur_end:
            cur_end, cur_name = e, name
    print(f"{len(ks)} kernels, {len(api)} HIP API calls, span {(ks[-1][2] - t0) / 1e6:.1f} ms; idle gaps >= {lo:g} us: {len(gaps)}, total {sum(b - a for a, b, _, _ in gaps) / 1e6:.2f} ms")
    for a, b, pn, nn in gaps:
        print(f"\n[gap {(b - a) / 1e3:8.1f} us at +{(a - t0) / 1e6:9.3f} ms]  {pn[:60]}  ->  {nn[:60]}")
        inside = [(n, max(s, a), min(e, b), s, e) for n, s, e in api if e > a and s < b]
        inside.sort(key=lambda t: -(t[2] - t[1]))
        for n, s2, e2, s, e in inside[:10]:
            print(f"    {n[:44]:44s} overlaps {(e2 - s2) / 1e3:8.1f} us (call {(e - s) / 1e3:8.1f} us, starts {(s - a) / 1e3:+9.1f} us from the gap's start)")
        print(f"    ... {len(inside)} calls overlap the gap")


if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:3]))
