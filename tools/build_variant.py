#!/usr/bin/env python3
"""Build a measurement variant of the library with extra -D flags (in-call A/B against the shipped build: bench.py --lib).
    python tools/build_variant.py nt -DT2P_C_AUX=2        ->  variants/libt2p_nt.so
Only the units named with --units (default: the gemm.hip parts) are recompiled with the flags; the other objects are the
shipped ones.  The output is git-ignored (*.so) and travels to the GPU box with the snapshot."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2protein_amd import build as B


def main():
    name = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith("-D")]
    units = [a.split("=", 1)[1].split(",") for a in sys.argv[2:] if a.startswith("--units=")]
    units = units[0] if units else ["gemm.hip"]
    B.build(verbose=False)
    out_dir = os.path.join(os.path.dirname(B.CSRC), "..", "variants")
    out_dir = os.path.abspath(out_dir)
    os.makedirs(out_dir, exist_ok=True)
    hipcc = B._hipcc()
    objs, procs = [], []
    for src in B.SOURCES:
        parts = list(range(B.GEMM_PARTS)) if src == "gemm.hip" else [None]
        for part in parts:
            stem = os.path.splitext(src)[0] + ("" if part is None else f"_part{part}")
            if src not in units:
                objs.append(os.path.join(B.CSRC, stem + ".o"))
                continue
            obj = os.path.join(out_dir, f"{name}_{stem}.o")
            objs.append(obj)
            cmd = [hipcc, *B.FLAGS, *defs, "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj]
            if part is not None:
                cmd[1:1] = [f"-DT2P_GEMM_PART={part}"]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(" ".join(cmd) + "\n" + out)
    lib = os.path.join(out_dir, f"libt2p_{name}.so")
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib, *objs], check=True)
    subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({lib!r})"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
