"""Does a device buffer cross a process boundary on this host with / without HSA_ENABLE_IPC_MODE_LEGACY=0?

    python tools/ipc_probe.py          # runs the child twice: variable = "0", then variable unset, then "1"

RCCL's intra-node transport and torch's CUDA-tensor sharing both export device allocations with hipIpcGetMemHandle.  One GPU is
enough to see whether that call works: the parent process allocates a tensor and a spawned child opens it through
torch.multiprocessing's CUDA IPC path.  Prints one line per setting; the log is committed under profiles/ as the evidence behind
text2protein_amd/distributed.py's launcher default."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
import torch.multiprocessing as mp

def reader(q, out):
    t = q.get(timeout=30)
    out.put(float(t.sum().item()))

if __name__ == "__main__":
    mp.set_start_method("spawn")
    x = torch.arange(1024, device="cuda", dtype=torch.float32)
    q, out = mp.Queue(), mp.Queue()
    p = mp.Process(target=reader, args=(q, out))
    p.start()
    q.put(x)
    v = out.get(timeout=40)
    p.join(10)
    print("shared tensor sum", v, "expected", float(x.sum().item()))
'''


def main():
    import tempfile
    d = tempfile.mkdtemp()
    child_path = os.path.join(d, "ipc_child.py")          # spawn re-imports the main module: it has to be a file
    with open(child_path, "w") as f:
        f.write(CHILD)
    for setting in ("0", None, "1"):
        env = dict(os.environ)
        env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
        if setting is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = setting
        try:
            r = subprocess.run([sys.executable, child_path], env=env, capture_output=True, text=True, timeout=120)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.splitlines() if "Ipc" in l or "ipc" in l or "Error" in l or "error" in l]
            print(f"HSA_ENABLE_IPC_MODE_LEGACY={'<unset>' if setting is None else setting}: rc={r.returncode} | {tail} | {' / '.join(err[-3:])}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"HSA_ENABLE_IPC_MODE_LEGACY={'<unset>' if setting is None else setting}: TIMEOUT (child hung)", flush=True)


if __name__ == "__main__":
    main()
