#!/usr/bin/env python3
"""Per-block times of one PC step (bench.py --layers FILE) as a table by block and by (kind, map side).

    python tools/layer_table.py gpurun_out/layers_cfg2.csv > profiles/rNN_layers_cfg2.md
"""
import collections
import sys


def main():
    path = sys.argv[1]
    rows, head = [], ""
    for line in open(path):
        if line.startswith("#"):
            head = line[1:].strip()
            continue
        f = line.strip().split(",")
        if len(f) == 6:
            rows.append((f[0], f[1], int(f[2]), int(f[3]), int(f[4]), float(f[5])))
    total = sum(r[5] for r in rows)
    print(f"# per-block times, {head}\n\ntotal inside blocks: {total:.2f} ms per PC step\n")
    by = collections.OrderedDict()
    for p, k, h, ci, co, ms in rows:
        key = (k, h)
        a = by.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms
    print("| kind | map side | blocks/step | ms/step | avg us | share |\n|---|---|---|---|---|---|")
    for (k, h), (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {h} | {n} | {ms:.3f} | {ms / n * 1e3:.1f} | {ms / total * 100:.1f} % |")
    lvl = collections.defaultdict(float)
    for p, k, h, ci, co, ms in rows:
        lvl[h] += ms
    print("\n| map side | ms/step | share |\n|---|---|---|")
    for h, ms in sorted(lvl.items(), key=lambda kv: -kv[0]):
        print(f"| {h} | {ms:.3f} | {ms / total * 100:.1f} % |")
    print("\n| block | kind | side | Cin | Cout | us (first evaluation) |\n|---|---|---|---|---|---|")
    seen = set()
    for p, k, h, ci, co, ms in rows:
        if p in seen:
            continue
        seen.add(p)
        print(f"| {p} | {k} | {h} | {ci} | {co} | {ms * 1e3:.1f} |")


if __name__ == "__main__":
    main()
