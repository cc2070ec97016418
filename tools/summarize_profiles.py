#!/usr/bin/env python3
"""Turn rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

    python tools/summarize_profiles.py --tag r01 --stats gpurun_out/prof_r01 \
        --fetch gpurun_out/pmc_r01_fetch --write gpurun_out/pmc_r01_write --steps-in-trace 7
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_DEMANGLED = {}


def short(name):
    if name.startswith("_Z"):          # rocprofv3 leaves names with _Float16 pointers mangled: ask llvm-cxxfilt
        if name not in _DEMANGLED:
            import shutil
            import subprocess
            tool = shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
            try:
                _DEMANGLED[name] = subprocess.run([tool, name], capture_output=True, text=True, timeout=10).stdout.strip() or name
            except Exception:
                _DEMANGLED[name] = name
            if _DEMANGLED[name].startswith("_Z"):      # (this llvm-cxxfilt does not know DF16_ either): kernel name + template arguments by hand
                import re
                m = re.match(r"_ZN3t2p(\d+)", name)
                if m:
                    n0 = m.end()
                    base = name[n0:n0 + int(m.group(1))]
                    rest, args = name[n0 + int(m.group(1)):], []
                    if rest.startswith("I"):
                        rest = rest[1:]
                        while rest and not rest.startswith("E"):
                            mi = re.match(r"Li(\d+)E", rest) or re.match(r"Lb([01])E", rest)
                            mt = re.match(r"NS_(\d+)", rest)
                            if mi:
                                args.append(("true" if mi.group(1) == "1" else "false") if rest.startswith("Lb") else mi.group(1))
                                rest = rest[mi.end():]
                            elif mt:
                                k = mt.end()
                                args.append(rest[k:k + int(mt.group(1))])
                                rest = rest[k + int(mt.group(1)):]
                                rest = rest[1:] if rest.startswith("E") and not rest.startswith("EE") else rest
                            else:
                                break
                    _DEMANGLED[name] = base + ("<" + ", ".join(args) + ">" if args else "")
        name = _DEMANGLED[name]
    return name.split("(")[0].replace("void t2p::", "").replace("t2p::", "")


def newest(pattern):
    """Scratch directories may hold files of earlier calls: take the most recent match."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--steps-in-trace", type=int, required=True, help="PC steps executed under the profiler (warmup + timed + roofline leg)")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"), help="output directory")
    a = ap.parse_args()
    out = a.out
    os.makedirs(out, exist_ok=True)
    ks = newest(os.path.join(a.stats, "*", "*kernel_stats.csv"))
    sfx = "" if a.workload == "cfg2" else "_" + a.workload
    shutil.copy(ks, os.path.join(out, f"{a.tag}_kernel_stats{sfx}.csv"))
    rows = list(csv.DictReader(open(newest(os.path.join(a.stats, "*", "*kernel_trace.csv")))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # steady state = from the first dispatch of the first score evaluation (timestep_embedding_kernel); everything before it is set-up:
    # weight uploads (one staged host-to-device copy = one __amd_rocclr_copyBuffer dispatch per tensor), text K/V projection, prior
    first = next((i for i, r in enumerate(rows) if "timestep_embedding_kernel" in r["Kernel_Name"]), 0)
    setup, rows = rows[:first], rows[first:]
    copies_setup = sum("copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"] for r in setup)
    copies_steady = sum("copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"] for r in rows)
    per = collections.defaultdict(lambda: [0, 0.0])
    tot = 0.0
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        per[key][0] += 1
        per[key][1] += d
    n = a.steps_in_trace
    with open(os.path.join(out, f"{a.tag}_kernels_by_shape{sfx}.md"), "w") as f:
        f.write(f"# {a.tag}: kernels by launch shape (rocprofv3 --kernel-trace, {n} PC steps, {a.workload} f16)\n\n")
        f.write(f"Dispatches before the first score evaluation (set-up, not counted below): {len(setup)}, of which runtime copy / fill kernels "
                f"(`__amd_rocclr_copyBuffer` / `fillBuffer`: the staged uploads of the weights): {copies_setup}.  Runtime copy / fill kernels "
                f"inside the {n} PC steps: {copies_steady}.  Dispatches per PC step: {len(rows) / n:.1f}.\n\n")
        f.write(f"GPU time per PC step: {tot / n / 1e3:.2f} ms\n\n| kernel | grid (workgroups) | launches/step | ms/step | avg us |\n|---|---|---|---|---|\n")
        for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]:
            f.write(f"| `{k[0]}` | {k[1]}x{k[2]}x{k[3]} | {v[0] / n:.1f} | {v[1] / n / 1e3:.2f} | {v[1] / v[0]:.1f} |\n")
    traffic = {}
    for kind, d in (("fetch", a.fetch), ("write", a.write)):
        if not d:
            continue
        rr = list(csv.DictReader(open(newest(os.path.join(d, "*", "*counter_collection.csv")))))
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in rr:
            nm = short(r["Kernel_Name"])
            acc[nm][0] += 1
            acc[nm][1] += float(r["Counter_Value"])
        for nm, (c, v) in acc.items():
            traffic.setdefault(nm, {})[kind + "_kb_per_launch"] = v / c
            traffic[nm]["launches_" + kind] = c
    if traffic:
        # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled
        # (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane stores.  Units: KiB.
        for nm, t in traffic.items():
            t["hbm_bytes_per_launch"] = (2.0 * t.get("fetch_kb_per_launch", 0.0) + t.get("write_kb_per_launch", 0.0)) * 1024.0
        json.dump(traffic, open(os.path.join(out, f"{a.tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print("wrote summaries for", a.tag)


if __name__ == "__main__":
    main()
