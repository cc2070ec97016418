"""The sampling_6d.py entry point end to end on the GPU: reference-layout checkpoint in,
reference-layout pickles out (sampling_6d.py:41-53, 160-162)."""
import os
import pickle
import subprocess
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_roundtrip(tmp_path):
    from text2protein_amd import checkpoint
    from text2protein_amd.config import tiny_config
    cfg = tiny_config(**{"model.num_scales": 4, "model.condition": ["length"]})
    cfg_path = tmp_path / "tiny.yml"
    with open(cfg_path, "w") as f:
        yaml.safe_dump(yaml.safe_load(__import__("json").dumps(cfg)), f)
    ckpt_dir = tmp_path / "training" / "tiny" / "run0" / "checkpoints"
    ckpt_dir.mkdir(parents=True)
    ckpt = checkpoint.save_synthetic_checkpoint(str(ckpt_dir / "best.pth"), cfg, seed=2)
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "sampling_6d.py"), str(cfg_path), ckpt, "--batch_size", "3", "--tag", "t",
           "--select_length", "1", "--length_index", "9", "--dtype", "f32", "--context_tokens", "4", "--outdir", str(out),
           "--ids", "a,b,c"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    files = sorted(os.listdir(out))
    assert files == ["sampled_a.pkl", "sampled_b.pkl", "sampled_c.pkl"]
    with open(out / "sampled_a.pkl", "rb") as f:
        t = pickle.load(f)
    L = cfg.data.max_res_num
    assert isinstance(t, torch.Tensor) and tuple(t.shape) == (1, cfg.data.num_channels, L, L) and t.dtype == torch.float32
    n = cfg.data.min_res_num + 9 - 1                      # --length_index is 1-based (sampling_6d.py:145)
    m = torch.zeros(L, L)
    m[:n, :n] = 1
    assert torch.equal(t[0, -1], m)                       # last channel carries the length mask
    assert torch.isfinite(t).all()
