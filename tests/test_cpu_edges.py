"""CPU checks of the stages either side of the sampler: the oracle's 6D decode on
hand-made cases, the tokenizer call semantics of the text-context producer and the embedding-table loader."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import t2p_oracle as O


def _sample(L, sel, seed=0, C=5):
    """(C, L, L) sample whose mask channel is outer(sel, sel) with a little noise around 0 / 1."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.6, 1.6, (C, L, L)).astype(np.float32)
    m = np.outer(sel, sel).astype(np.float32)
    x[-1] = m + rng.uniform(-0.3, 0.3, (L, L)).astype(np.float32)
    return x


def test_oracle_decode_top_left_and_scattered_masks():
    L = 12
    sel = np.zeros(L, bool); sel[:7] = True
    x = _sample(L, sel)
    d = O.decode_6d(x)
    assert d["L"] == 7
    assert np.array_equal(d["dist"], np.clip(x[0][:7, :7], -1, 1))
    sel2 = np.zeros(L, bool); sel2[[0, 3, 4, 9, 11]] = True
    x2 = _sample(L, sel2, seed=1)
    d2 = O.decode_6d(x2[None])                      # (1, C, L, L) as pickled by sampling_6d.py:160-162
    assert d2["L"] == 5
    assert np.array_equal(d2["theta"], np.clip(x2[2][np.ix_(sel2, sel2)], -1, 1))
    assert d2["dist_abs"].dtype == np.float32 and d2["dist_abs"].min() >= 0 and d2["dist_abs"].max() <= 20
    assert np.allclose(d2["phi_abs"], (d2["phi"] + 1) * math.pi / 2, rtol=1e-6)
    assert abs(float(d2["omega_abs"].max())) <= math.pi + 1e-6


def test_oracle_decode_improper_mask_and_ties():
    x = np.zeros((5, 4, 4), np.float32)
    x[-1, 0, :3] = 1                                 # 3 ones: not a perfect square
    with pytest.raises(ValueError, match="improper masking"):
        O.decode_6d(x)
    y = np.zeros((5, 4, 4), np.float32)
    y[-1, 0, 0] = 0.5                                # round half to even: 0.5 -> 0, 1.5 -> 2, neither is 1
    y[-1, 0, 1] = 1.5
    y[-1, 1, 1] = 0.50001
    d = O.decode_6d(y)
    assert d["L"] == 1 and d["dist"].shape == (1, 1)
    assert O.decode_6d(np.zeros((5, 4, 4), np.float32))["L"] == 0


@pytest.fixture(scope="module")
def tiny_tokenizer_dir(tmp_path_factory):
    import sentencepiece as spm
    d = tmp_path_factory.mktemp("tok")
    corpus = d / "corpus.txt"
    lines = ["the protein binds atp and forms a helix bundle", "a beta sheet enzyme that hydrolyses peptides",
             "membrane transporter with twelve helices", "dna binding zinc finger domain"]
    corpus.write_text("\n".join(lines * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(d / "tokenizer"), vocab_size=80, model_type="bpe",
                                   unk_id=0, bos_id=1, eos_id=2, pad_id=-1, minloglevel=2)
    json.dump({"tokenizer_class": "LlamaTokenizer", "unk_token": "<unk>", "bos_token": "<s>", "eos_token": "</s>",
               "padding_side": "right", "legacy": True}, open(d / "tokenizer_config.json", "w"))
    return str(d)


def test_tokenizer_call_semantics(tiny_tokenizer_dir):
    from text2protein_amd.text_context import load_tokenizer, tokenize_captions
    tok = load_tokenizer(tiny_tokenizer_dir)
    assert tok.pad_token == tok.unk_token            # padding=True needs a pad token; vicuna uses <unk>
    caps = ["the protein binds atp", "a beta sheet enzyme that hydrolyses peptides and forms a helix bundle", "dna"]
    ids = tokenize_captions(tok, caps, max_length=512)
    lens = [len(tok.encode(c, add_special_tokens=False)) for c in caps]
    assert ids.shape == (3, max(lens)) and ids.dtype == torch.int64      # padded to the longest caption of the batch
    for i, n in enumerate(lens):
        assert ids[i, :n].tolist() == tok.encode(caps[i], add_special_tokens=False)
        assert (ids[i, n:] == tok.pad_token_id).all()
        assert tok.bos_token_id not in ids[i, :n].tolist()               # add_special_tokens=False
    short = tokenize_captions(tok, caps, max_length=6)                   # truncation=True at max_length
    assert short.shape == (3, 6) and short[1].tolist() == ids[1, :6].tolist()


def test_embedding_table_loader(tmp_path):
    from safetensors.torch import save_file
    from text2protein_amd.text_context import load_embedding_table
    t = torch.arange(40, dtype=torch.float32).reshape(10, 4)
    torch.save({"model.embed_tokens.weight": t, "lm_head.weight": t * 2}, tmp_path / "pytorch_model.bin")
    assert torch.equal(load_embedding_table(str(tmp_path / "pytorch_model.bin")), t)
    sh = tmp_path / "sharded"
    sh.mkdir()
    save_file({"model.embed_tokens.weight": t.half()}, str(sh / "model-00001-of-00002.safetensors"))
    save_file({"lm_head.weight": t}, str(sh / "model-00002-of-00002.safetensors"))
    json.dump({"weight_map": {"model.embed_tokens.weight": "model-00001-of-00002.safetensors",
                              "lm_head.weight": "model-00002-of-00002.safetensors"}}, open(sh / "model.safetensors.index.json", "w"))
    got = load_embedding_table(str(sh))
    assert got.dtype == torch.float16 and torch.equal(got.float(), t)
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(FileNotFoundError):
        load_embedding_table(str(empty))
