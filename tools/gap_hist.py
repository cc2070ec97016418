"""Inter-kernel gaps of the steady-state PC steps in a rocprofv3 --kernel-trace CSV.

    python tools/gap_hist.py <kernel_trace.csv> [label]

A PC step is delimited by consecutive predictor_update_kernel dispatches; for the last 4 complete steps prints the step time, the
sum of kernel durations, the sum of the gaps between consecutive kernels (start[i+1] - end[i], clamped at 0) and a histogram of
the gaps.  Used to compare eager launches with hipGraph replay (bench.py --graph 1)."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
    marks = [i for i, e in enumerate(ev) if "predictor_update_kernel" in e[2]]
    if len(marks) < 6:
        print(label, "too few steps in the trace")
        return
    steps = list(zip(marks[-5:-1], marks[-4:]))
    tot = busy = gaps = n = 0
    hist = {"<0.5us": 0, "0.5-1": 0, "1-2": 0, "2-4": 0, "4-8": 0, ">8us": 0}
    gsum = dict.fromkeys(hist, 0.0)
    for a, b in steps:
        seg = ev[a:b + 1]
        tot += seg[-1][0] - seg[0][0]
        for (s0, e0, _), (s1, _e1, _n) in zip(seg[:-1], seg[1:]):
            busy += e0 - s0
            g = max(0, s1 - e0) / 1e3
            gaps += g
            n += 1
            k = "<0.5us" if g < 0.5 else "0.5-1" if g < 1 else "1-2" if g < 2 else "2-4" if g < 4 else "4-8" if g < 8 else ">8us"
            hist[k] += 1
            gsum[k] += g
    big = []
    for a, b in steps[-1:]:
        seg = ev[a:b + 1]
        for (s0, e0, n0), (s1, _e1, n1) in zip(seg[:-1], seg[1:]):
            if s1 - e0 > 4000:
                big.append(((s1 - e0) / 1e3, n0.split("(")[0][-60:], n1.split("(")[0][-60:]))
    ns = len(steps)
    print(f"{label}: {ns} steps, {n / ns:.0f} dispatches/step, step {tot / ns / 1e6:.3f} ms, kernels {busy / ns / 1e6:.3f} ms, gaps {gaps / ns / 1e3:.3f} ms")
    for g, n0, n1 in big:
        print(f"   gap {g:7.1f} us after {n0} before {n1}")
    print("   gap histogram (count per step, ms per step): " + ", ".join(f"{k}: {hist[k] / ns:.0f} / {gsum[k] / ns / 1e3:.3f}" for k in hist))


if __name__ == "__main__":
    main()
