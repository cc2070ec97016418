"""Markdown table of the GEMM / convolution launches of a profiled bench run by operand shape
(bench.py --shapes FILE): where the time of the 'other GEMM' class goes.
usage: python tools/shapes_table.py gpurun_out/shapes_cfg2.csv [--top 40]"""
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    lines = open(path).read().splitlines()
    head = lines[0].lstrip("# ")
    nprof = int(head.split(":")[1].split()[0])
    rows = []
    for ln in lines[2:]:
        kind, M, N, K, taps, z, n, ms, fl = ln.split(",")
        rows.append((float(ms) / nprof, int(kind), int(M), int(N), K, int(taps), int(z), float(n) / nprof, float(fl)))
    rows.sort(reverse=True)
    names = {0: "conv3x3 (LDS-DMA)", 1: "GEMM", 2: "conv3x3 (register-staged)"}
    print(f"# GEMM / convolution launches by operand shape: {head}\n")
    print("| class | M | N | K | taps | batch | launches/step | ms/step | avg us | TFLOP/s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    tot = {}
    for ms, kind, M, N, K, taps, z, n, fl in rows:
        tot[kind] = tot.get(kind, 0.0) + ms
    for ms, kind, M, N, K, taps, z, n, fl in rows[:top]:
        tf = fl / nprof / (ms * 1e-3) / 1e12 if ms > 0 else 0
        print(f"| {names[kind]} | {M} | {N} | {K} | {taps} | {z} | {n:.0f} | {ms:.3f} | {ms / n * 1e3:.1f} | {tf:.0f} |")
    print()
    for k, v in sorted(tot.items()):
        print(f"total {names[k]}: {v:.2f} ms/step")


if __name__ == "__main__":
    main()
