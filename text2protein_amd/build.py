"""Build libt2p_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to the sources
so that it travels to the GPU box with the repository snapshot."""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libt2p_hip.so")
LIB_ABLATION = os.path.join(CSRC, "libt2p_hip_ablation.so")   # measurement builds only (tools/bench_conv.py --ablation)
SOURCES = ["gemm.hip", "kernels.hip", "attention.hip", "smallconv.hip", "stfuse.hip", "engine.cpp", "capi.cpp",
           "train_kernels.hip", "train.cpp", "train_capi.cpp"]
GEMM_PARTS = 7          # gemm.hip is compiled once per -DT2P_GEMM_PART=k: the LDS-DMA instantiations build in parallel
HEADERS = ["t2p_common.h", "t2p_kernels.h", "engine.h", "train_kernels.h", "train.h", os.path.join("..", "..", "include", "t2p.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


STAMP = LIB + ".srchash"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
_COMPILER_ID = None


def compiler_id() -> str:
    """Version text of the compiler that would build the library: part of every hash, so objects of another ROCm release are
    never linked or loaded as if they were current.  Empty when no hipcc is installed (a box that only loads the prebuilt library)."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        try:
            _COMPILER_ID = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout.strip()
        except (OSError, RuntimeError, subprocess.SubprocessError):
            _COMPILER_ID = ""
    return _COMPILER_ID


def source_hash(ablation: bool = False) -> str:
    """sha256 over every source, header and compiler flag: the identity of what libt2p_hip.so was built from.
    (File times do not survive the copy to the GPU box; contents do.)"""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + (["-DT2P_ABLATION"] if ablation else [])).encode())
    h.update(compiler_id().encode())
    for f in SOURCES + HEADERS:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def unit_hash(src: str, flags) -> str:
    """Identity of one object file: its source, every header and the flags (unchanged units are not recompiled)."""
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    h.update(compiler_id().encode())
    seen, todo = [], [os.path.join(CSRC, src)]
    while todo:                                   # the source and the local headers it includes, transitively
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f, "rb") as fh:
            data = fh.read()
        h.update(os.path.relpath(f, CSRC).encode())
        h.update(data)
        for inc in re.findall(rb'^\s*#\s*include\s+"([^"]+)"', data, re.M):
            todo.append(os.path.join(os.path.dirname(f), inc.decode()))
    return h.hexdigest()


def _read(path) -> str:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def built_hash() -> str:
    return _read(STAMP)


def needs_build() -> bool:
    """True when the library is missing or was built from other sources than the ones in the tree."""
    return not os.path.exists(LIB) or built_hash() != source_hash()


def scratch_users(remarks: str, kernel_prefix: str = "_kernel"):
    """Kernels whose -Rpass-analysis=kernel-resource-usage remarks report scratch (private memory).

    The LDS-DMA GEMM keeps 128 accumulator registers per lane; if a loop over them is left rolled
    (a `#pragma unroll` body that grew too large is skipped silently) they move to scratch and the
    kernel runs several times slower while still producing correct results.  Caught here, on the CPU."""
    bad, name = [], None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and kernel_prefix in name and int(m.group(1)) > 0:
            bad.append((name, int(m.group(1))))
    return bad


def build(force: bool = False, verbose: bool = True, ablation: bool = False) -> str:
    """Compile every HIP/C++ source for gfx950 and link the shared library; returns its path.

    ``ablation`` adds -DT2P_ABLATION and writes a SEPARATE library (libt2p_hip_ablation.so): the timing-only
    switches of the LDS-DMA GEMM that skip work (and so give wrong results) exist in that build only; the product
    library is never built with it."""
    lib, stamp = (LIB_ABLATION, LIB_ABLATION + ".srchash") if ablation else (LIB, STAMP)
    if not force and os.path.exists(lib) and _read(stamp) == source_hash(ablation):
        return lib
    # one builder at a time: the ranks of a multi-process job all arrive here when the library is stale; the first builds,
    # the others wait on the lock and then find the stamp in place
    import fcntl
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and os.path.exists(lib) and _read(stamp) == source_hash(ablation):
                return lib
            return _build_locked(lib, stamp, verbose, ablation)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(lib: str, stamp: str, verbose: bool, ablation: bool) -> str:
    hipcc = _hipcc()
    objs = []
    procs = []
    pending = {}
    flags = FLAGS + (["-DT2P_ABLATION"] if ablation else [])
    if os.path.exists(stamp):
        os.remove(stamp)
    units = [(src, None) for src in SOURCES if src != "gemm.hip"] + [("gemm.hip", k) for k in range(GEMM_PARTS)]
    for src, part in sorted(units, key=lambda u: u[1] is None):       # the long gemm.hip units first
        stem = os.path.splitext(src)[0] + ("" if part is None else f"_part{part}") + ("_abl" if ablation else "")
        obj = os.path.join(CSRC, stem + ".o")
        objs.append(obj)
        cmd = [hipcc, *flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if part is not None:
            cmd[1:1] = ["-Rpass-analysis=kernel-resource-usage", f"-DT2P_GEMM_PART={part}"]
        uh = unit_hash(src, cmd[1:-4])
        if os.path.exists(obj) and _read(obj + ".hash") == uh:
            continue
        if os.path.exists(obj + ".hash"):
            os.remove(obj + ".hash")
        pending[obj] = uh
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if src == "gemm.hip":
            bad = scratch_users(out)
            if bad:
                raise RuntimeError("LDS-DMA GEMM kernels spill to scratch memory (accumulator loop left rolled?): "
                                   + ", ".join(f"{n} {b} B/lane" for n, b in bad[:4]))
            keep, skip = [], 0          # drop the remarks and the two source-context lines that follow each of them
            for l in out.splitlines():
                if "remark:" in l:
                    skip = 2
                elif skip and re.match(r"^\s*(\d+\s*)?\|", l):
                    skip -= 1
                else:
                    skip = 0
                    keep.append(l)
            out = "\n".join(keep)
        with open(obj + ".hash", "w") as f:
            f.write(pending[obj] + "\n")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    # every kernel's host stub must be there (a template body the host pass rejects is dropped silently): dlopen it
    r = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({lib!r})"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{lib} does not load:\n{r.stderr[-2000:]}")
    with open(stamp, "w") as f:
        f.write(source_hash(ablation) + "\n")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, ablation="--ablation" in sys.argv)
    print(LIB)
