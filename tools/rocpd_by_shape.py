#!/usr/bin/env python3
"""Aggregate a rocprofv3 rocpd database (default output format) by kernel and launch grid.
    python tools/rocpd_by_shape.py gpurun_out/prof_tmp/t_results.db --steps 8 [--top 50]
"""
import argparse
import collections
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--top", type=int, default=50)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    names = {r[0]: r[1] for r in cur.execute(f"select id, display_name from {ks}")}
    per = collections.defaultdict(lambda: [0, 0.0])
    fam = collections.defaultdict(float)
    tot = 0.0
    q = f"select kernel_id, start, end, workgroup_size_x, grid_size_x, grid_size_y, grid_size_z from {kd}"
    for kid, s, e, wx, gx, gy, gz in cur.execute(q):
        d = (e - s) / 1e3
        tot += d
        nm = names[kid].split("(")[0].replace("void t2p::", "").replace("t2p::", "")
        per[(nm, gx // wx, gy, gz)][1] += d
        per[(nm, gx // wx, gy, gz)][0] += 1
        fam[re.sub(r"<.*", "", nm)] += d
    n = a.steps
    print(f"GPU time per PC step: {tot / n / 1e3:.2f} ms\n")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:20]:
        print(f"  {k:32s} {v / n / 1e3:7.2f} ms/step")
    print("\n| kernel | grid (workgroups) | launches/step | ms/step | avg us |\n|---|---|---|---|---|")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"| `{k[0]}` | {k[1]}x{k[2]}x{k[3]} | {v[0] / n:.1f} | {v[1] / n / 1e3:.2f} | {v[1] / v[0]:.1f} |")


if __name__ == "__main__":
    main()
