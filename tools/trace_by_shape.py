#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV by kernel and launch grid, with the idle time between dispatches.
    python tools/trace_by_shape.py gpurun_out/prof_x/*/*_kernel_trace.csv --steps 8 [--top 60] [--skip-setup-ms 0]
Only the last `--steps` PC steps' worth of dispatches are counted when --tail-frac is given (setup kernels dropped).
"""
import argparse
import collections
import csv
import re


def short(name):
    return name.split("(")[0].replace("void t2p::", "").replace("t2p::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--from-kernel", default="timestep_embedding_kernel", help="start counting at the first dispatch of this kernel")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    start = next((i for i, r in enumerate(rows) if a.from_kernel in r["Kernel_Name"]), 0)
    rows = rows[start:]
    per = collections.defaultdict(lambda: [0, 0.0])
    fam = collections.defaultdict(lambda: [0, 0.0])
    tot = gap = 0.0
    hist = collections.Counter()
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = (e - s) / 1e3
        tot += d
        if prev_end is not None and s > prev_end and (s - prev_end) < 2e6:
            gap += (s - prev_end) / 1e3
        prev_end = max(prev_end or 0, e)
        nm = short(r["Kernel_Name"])
        key = (nm, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        per[key][0] += 1
        per[key][1] += d
        f = re.sub(r"<.*", "", nm)
        fam[f][0] += 1
        fam[f][1] += d
        hist[min(int(d // 10) * 10, 200)] += d
    n = a.steps
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print(f"dispatches/step {len(rows) / n:.0f}; kernel time {tot / n / 1e3:.2f} ms/step; idle between dispatches {gap / n / 1e3:.2f} ms/step; "
          f"span {span / n / 1e3:.2f} ms/step\n")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {k:34s} {v[0] / n:7.1f} launches  {v[1] / n / 1e3:7.2f} ms/step")
    print("\n  time in kernels by duration bucket (us): " + ", ".join(f"{b}+: {hist[b] / n / 1e3:.2f}" for b in sorted(hist)))
    print("\n| kernel | grid (workgroups) | launches/step | ms/step | avg us |\n|---|---|---|---|---|")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"| `{k[0]}` | {k[1]}x{k[2]}x{k[3]} | {v[0] / n:.1f} | {v[1] / n / 1e3:.2f} | {v[1] / v[0]:.1f} |")


if __name__ == "__main__":
    main()
