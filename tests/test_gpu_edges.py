"""GPU parity of the stages either side of the sampler against the oracle: the 6D decode
kernel (bit-exact, incl. scattered / improper / empty masks) and the text-context embedding gather."""
import json

import numpy as np
import pytest
import torch

from oracle import t2p_oracle as O
from test_cpu_edges import _sample, tiny_tokenizer_dir  # noqa: F401  (fixture re-used)

pytestmark = pytest.mark.gpu


def test_decode_6d_matches_oracle_bitwise():
    from text2protein_amd.decode import NAMES, decode_6d, decode_6d_batch
    L = 128
    sels = []
    s = np.zeros(L, bool); s[:100] = True; sels.append(s)                       # length mask (cfg3)
    s = np.zeros(L, bool); s[np.random.default_rng(1).choice(L, 57, replace=False)] = True; sels.append(s)   # scattered
    s = np.ones(L, bool); sels.append(s)                                        # full map
    s = np.zeros(L, bool); sels.append(s)                                       # empty mask: L = 0
    s = np.zeros(L, bool); s[5] = True; sels.append(s)                          # a single residue
    xs = np.stack([_sample(L, sel, seed=i, C=5) for i, sel in enumerate(sels)])
    got = decode_6d(torch.from_numpy(xs).cuda())
    for b in range(len(sels)):
        want = O.decode_6d(xs[b])
        assert got[b]["L"] == want["L"] == int(sels[b].sum())
        for nm in NAMES:
            assert np.array_equal(got[b][nm], want[nm]), (b, nm)
            assert np.array_equal(got[b][nm + "_abs"], want[nm + "_abs"]), (b, nm)
    # 8-channel samples (cfg5): channels 0..3 decoded, mask is the last channel
    x8 = _sample(64, np.arange(64) < 40, seed=9, C=8)
    g8 = decode_6d(torch.from_numpy(x8).cuda())[0]
    w8 = O.decode_6d(x8)
    assert g8["L"] == 40 and all(np.array_equal(g8[k], w8[k]) for k in w8 if k != "L")
    # improper mask: the batch call flags it, the reference-shaped call raises like sampling_rosetta.py:72-73
    bad = xs.copy()
    bad[1, -1, 0, 0] = 1.0 if bad[1, -1, 0, 0] < 0.5 else 0.0
    lengths, _, _ = decode_6d_batch(torch.from_numpy(bad).cuda())
    assert lengths.tolist()[1] == -1 and lengths.tolist()[0] == 100
    with pytest.raises(ValueError, match="improper masking"):
        decode_6d(torch.from_numpy(bad).cuda())


def test_decode_6d_large_map():
    from text2protein_amd.decode import decode_6d
    L = 256
    sel = np.zeros(L, bool); sel[3:203] = True
    x = _sample(L, sel, seed=4)
    g = decode_6d(torch.from_numpy(x).cuda())[0]
    w = O.decode_6d(x)
    assert g["L"] == 200 and all(np.array_equal(g[k], w[k]) for k in w if k != "L")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_embedding_gather_matches_oracle(dtype):
    from text2protein_amd.text_context import TextContextProducer
    g = torch.Generator().manual_seed(3)
    table = torch.randn(1000, 4096, generator=g)
    ids = torch.randint(0, 1000, (3, 37), generator=g)
    ids[0, -5:] = 0                                                            # padded tail embeds the pad row
    p = TextContextProducer(tokenizer=None, table=table, table_dtype=dtype)
    got = p.embed(ids)
    want = O.embed_tokens(table.to(dtype), ids)
    assert got.shape == (3, 37, 4096) and got.dtype == torch.float32
    assert torch.equal(got.cpu(), want)
    ids[1, 4] = 1000
    with pytest.raises(IndexError):
        p.embed(ids)


def test_text_context_end_to_end(tiny_tokenizer_dir, tmp_path):   # noqa: F811
    """captions -> context through the local tokenizer + table, fed to the score network's set_context."""
    from text2protein_amd.text_context import TextContextProducer
    table = torch.randn(80, 32, generator=torch.Generator().manual_seed(5))
    torch.save({"model.embed_tokens.weight": table}, tmp_path / "pytorch_model.bin")
    prod = TextContextProducer.from_local(tiny_tokenizer_dir, str(tmp_path / "pytorch_model.bin"))
    caps = ["a beta sheet enzyme that hydrolyses peptides", "dna binding zinc finger domain"]
    ctx = prod(caps)
    toks = prod.tokens(caps)
    assert ctx.shape == (2, toks.shape[1], 32)
    assert torch.equal(ctx.cpu(), O.embed_tokens(table, toks))
    # the tiny model of the golden fixtures has context_dim 32: the produced context drives a score evaluation
    from helpers import cfg_tiny
    from text2protein_amd import synth
    from text2protein_amd.model import HipScoreModel
    cfg = cfg_tiny()
    m = HipScoreModel(cfg, dtype="f32")
    m.load_state_dict(synth.synth_state_dict(cfg, 0))
    x = torch.randn(2, cfg.data.num_channels, 16, 16, generator=torch.Generator().manual_seed(1)).cuda()
    out = m(x, torch.tensor([0, 3]).cuda(), ctx)
    assert torch.isfinite(out).all()
