#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (csv output) -> profiles/<tag>_pmc_sq_tcc.json.

    python tools/summarize_pmc.py --tag r01 gpurun_out/pmc_r01_sq1 gpurun_out/pmc_r01_sq2 gpurun_out/pmc_r01_tcc

Each directory holds one pass (`rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 bench.py ...`).
Derived: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
(MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs); l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS).
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("gemm_dxs_kernel", "gemm_dma_kernel", "attn_flash_kernel", "attn_strip_kernel", "gemm_kernel", "gn_apply_kernel", "gn_apply16_kernel", "splitk_reduce_gn_kernel")


def short(name):
    return name.split("(")[0].replace("void t2p::", "").replace("t2p::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"), help="output directory")
    ap.add_argument("dirs", nargs="+")
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                nm = short(r["Kernel_Name"])
                if not nm.startswith(KEEP):
                    continue
                c = acc[nm][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
    out = {}
    for nm, cs in sorted(acc.items()):
        o = {k: v[1] / v[0] for k, v in cs.items()}
        o["dispatches"] = max(v[0] for v in cs.values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o:
            o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * o["GRBM_GUI_ACTIVE"] / 8.0)
        if "TCC_HIT_sum" in o and "TCC_MISS_sum" in o:
            o["l2_hit_rate"] = o["TCC_HIT_sum"] / max(o["TCC_HIT_sum"] + o["TCC_MISS_sum"], 1.0)
        out[nm] = o
    os.makedirs(a.out, exist_ok=True)
    path = os.path.join(a.out, f"{a.tag}_pmc_sq_tcc.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, len(out), "kernels")


if __name__ == "__main__":
    main()
