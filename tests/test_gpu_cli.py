"""The sampling_6d.py entry point end to end on the GPU: reference-layout checkpoint in,
reference-layout pickles out (sampling_6d.py:41-53, 160-162)."""
import os
import pickle
import subprocess
import sys

import pytest
import torch
import yaml

from test_cpu_edges import tiny_tokenizer_dir  # noqa: F401  (fixture)

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


def test_reference_written_checkpoint_restores_the_ema_weights():
    """tests/golden/tiny_checkpoint.pth was written by the reference's own save_checkpoint (DataParallel keys, Adam
    state, ExponentialMovingAverage.state_dict() whose shadow differs from the live weights); the expected score is
    what the reference computes after restore_checkpoint + ema.copy_to (sampling_6d.py:64-73)."""
    import numpy as np
    from helpers import GOLDEN, cfg_ckpt, load_golden, rel_l2
    from text2protein_amd import synth
    from text2protein_amd.checkpoint import restore_checkpoint
    from text2protein_amd.model import HipScoreModel
    g = load_golden("tiny_checkpoint_expected")
    cfg = cfg_ckpt()
    cfg.device = "cuda"
    x, labels, ctx = (torch.from_numpy(g[k]).cuda() for k in ("x", "labels", "context"))
    m = HipScoreModel(cfg, dtype="f32")
    step = restore_checkpoint(os.path.join(GOLDEN, "tiny_checkpoint.pth"), m, cfg)
    got = m(x, labels, ctx).cpu()
    assert step == int(g["step"]) == 1234
    e = rel_l2(got, g["score"])
    print(f"checkpoint-restored (EMA) score vs the reference's: {e:.3e}")
    assert e < 1e-5
    assert rel_l2(got, g["score_live"]) > 1e-2                      # not the live (non-EMA) weights
    direct = HipScoreModel(cfg, dtype="f32")
    direct.load_state_dict(synth.synth_state_dict(cfg, int(g["ema_seed"])))
    assert torch.equal(direct(x, labels, ctx).cpu(), got)           # == loading the EMA tensors by name
    live = HipScoreModel(cfg, dtype="f32")
    restore_checkpoint(os.path.join(GOLDEN, "tiny_checkpoint.pth"), live, cfg, use_ema=False)
    assert rel_l2(live(x, labels, ctx).cpu(), g["score_live"]) < 1e-5
    assert np.isfinite(g["score"]).all()


def test_cli_captions_and_decode(tmp_path, tiny_tokenizer_dir):
    """captions -> local tokenizer + embedding table -> context (sampling_6d.py:121-137) and --decode
    (sampling_rosetta.py:69-96): the decoded maps equal the oracle's decode of the pickled sample."""
    import numpy as np
    from oracle import t2p_oracle as O
    from text2protein_amd import checkpoint
    from text2protein_amd.config import tiny_config
    cfg = tiny_config(**{"model.num_scales": 3, "model.condition": ["length"]})
    cfg_path = tmp_path / "tiny.yml"
    with open(cfg_path, "w") as f:
        yaml.safe_dump(yaml.safe_load(__import__("json").dumps(cfg)), f)
    ckpt_dir = tmp_path / "training" / "tiny" / "run0" / "checkpoints"
    ckpt_dir.mkdir(parents=True)
    ckpt = checkpoint.save_synthetic_checkpoint(str(ckpt_dir / "best.pth"), cfg, seed=2)
    table = torch.randn(80, cfg.model.context_dim, generator=torch.Generator().manual_seed(7))
    torch.save({"model.embed_tokens.weight": table}, tmp_path / "pytorch_model.bin")
    (tmp_path / "caps.txt").write_text("1abc_A\tthe protein binds atp\n2xyz_B\tmembrane transporter with twelve helices\n")
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "sampling_6d.py"), str(cfg_path), ckpt, "--batch_size", "2", "--dtype", "f32",
           "--select_length", "1", "--length_index", "5", "--outdir", str(out), "--captions", str(tmp_path / "caps.txt"),
           "--tokenizer_path", tiny_tokenizer_dir, "--embed_table", str(tmp_path / "pytorch_model.bin"), "--decode"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(out)) == ["decoded_1abc_A.npz", "decoded_2xyz_B.npz", "sampled_1abc_A.pkl", "sampled_2xyz_B.pkl"]
    with open(out / "sampled_2xyz_B.pkl", "rb") as f:
        t = pickle.load(f)
    want = O.decode_6d(t.numpy())
    got = np.load(out / "decoded_2xyz_B.npz")
    assert want["L"] == cfg.data.min_res_num + 5 - 1
    for k in ("dist", "omega", "theta", "phi", "dist_abs", "omega_abs", "theta_abs", "phi_abs"):
        assert np.array_equal(got[k], want[k]), k


def test_cli_inpainting_from_known_maps(tmp_path):
    """--inpaint_coords + --mask_info: the conditions the reference builds from --pdb (utils.py:84-137 -> sampling.py:259-287),
    driven from the entry point on a model trained with ["length", "inpainting"]."""
    from text2protein_amd.config import tiny_config
    cfg = tiny_config(**{"model.num_scales": 4, "model.condition": ["length", "inpainting"], "data.num_channels": 8})
    cfg_path = tmp_path / "tiny_inp.yml"
    with open(cfg_path, "w") as f:
        yaml.safe_dump(yaml.safe_load(__import__("json").dumps(cfg)), f)
    L, C = cfg.data.max_res_num, 8
    n = L - 3
    coords = torch.rand(1, C, L, L) * 2 - 1
    torch.save({"coords_6d": coords, "lengths": [n]}, tmp_path / "known.pt")
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "sampling_6d.py"), str(cfg_path), "synthetic", "--batch_size", "2", "--dtype", "f32",
           "--context_tokens", "4", "--outdir", str(out), "--inpaint_coords", str(tmp_path / "known.pt"), "--mask_info", "1:2,5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    with open(out / "sampled_1.pkl", "rb") as f:
        t = pickle.load(f)[0]
    sel = torch.zeros(L, dtype=torch.bool)
    sel[[1, 2, 5]] = True
    inside = torch.arange(L) < n
    free = (inside[:, None] & inside[None, :]) & (sel[:, None] | sel[None, :])      # conditional_mask of sampling.py:259-275
    assert torch.isfinite(t).all()
    assert torch.equal(t[:-1][:, ~free], coords[0, :-1][:, ~free])                   # everything else is the known map
    assert torch.equal(t[-1], coords[0, -1])                                        # the last channel is never free
    assert float((t[:-1][:, free] - coords[0, :-1][:, free]).abs().max()) > 1e-3    # the selected residues were sampled
    # without a source of known maps the flag is an error, not a silent no-op
    r2 = subprocess.run(cmd[:-4] + ["--mask_info", "1:2"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r2.returncode != 0 and "--inpaint_coords" in r2.stderr + r2.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL path (runs the first time a multi-GPU box is seen)")
def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` as the driver starts it: two rank processes, nccl (= RCCL) backend, the all_gather of the run."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-roofline", "--no-cfg3"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("T2P_DIST_BACKEND", "T2P_FORCE_DEVICE")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["finite"] is True and rec["value"] > 0


def test_cli_runs_a_vp_model_in_the_fused_loop(tmp_path):
    """training.sde: vpsde through the entry point (sampling_6d.py:80-82): the fused sampler with its per-step VP tables."""
    from text2protein_amd.config import tiny_config
    cfg = tiny_config(**{"model.num_scales": 40, "training.sde": "vpsde"})     # (beta_max / N must stay below 1: N >= 21)
    cfg_path = tmp_path / "tiny_vp.yml"
    with open(cfg_path, "w") as f:
        yaml.safe_dump(yaml.safe_load(__import__("json").dumps(cfg)), f)
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "sampling_6d.py"), str(cfg_path), "synthetic", "--batch_size", "2", "--dtype", "f32",
           "--context_tokens", "4", "--outdir", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    with open(out / "sampled_0.pkl", "rb") as f:
        t = pickle.load(f)
    assert tuple(t.shape) == (1, cfg.data.num_channels, cfg.data.max_res_num, cfg.data.max_res_num) and torch.isfinite(t).all()
    assert "80 score evaluations" in r.stdout
