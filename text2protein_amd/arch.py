"""Structure of the score network: block list and parameter table.

This is a *description* (no tensors, no compute) of what reference
``score_sde_pytorch/models/ncsnpp.py:74-217`` (``UNetModel.__init__``) builds for a config: the
ordered U-Net stages and, for each, the state-dict names and shapes of its parameters, in
``model.parameters()`` registration order (that order is what the reference's EMA
``shadow_params`` list follows, ema.py:51-64, 90-93).  The HIP engine builds the same structure
on its own from the flat config (csrc/engine.cpp); tests check both agree.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class ParamSpec:
    name: str                 # state-dict key without the DataParallel "module." prefix
    shape: Tuple[int, ...]
    kind: str                 # "weight" | "bias" | "norm_scale" | "norm_shift"
    fan_in: int = 1


@dataclass
class Layer:
    kind: str                 # "res" | "attn" | "st"
    prefix: str               # e.g. "input_blocks.3.0"
    in_ch: int
    out_ch: int
    up: bool = False
    down: bool = False
    res_in: int = 0           # map side at the layer's input


@dataclass
class Stage:
    """One TimestepEmbedSequential (reference ncsnpp.py:49-69)."""
    prefix: str
    layers: List[Layer] = field(default_factory=list)
    skip_ch: int = 0          # channels popped from the skip stack and concatenated (out stages)


@dataclass
class Arch:
    nf: int
    channels: int
    temb_dim: int
    resolutions: List[int]
    input_stages: List[Stage]
    mid_stage: Stage
    out_stages: List[Stage]
    final_ch: int

    def all_layers(self):
        for st in self.input_stages + [self.mid_stage] + self.out_stages:
            for l in st.layers:
                yield l


def gn_groups(c: int) -> int:
    """nn.GroupNorm(num_groups=min(C // 4, 32)) -- reference layers.py:282,292, ncsnpp.py:214."""
    return min(c // 4, 32)


def build_arch(config) -> Arch:
    m = config.model
    if m.resblock_type.lower() != "biggan":
        raise ValueError("only resblock_type 'biggan' is on the sampling path (no shipped config uses 'ddpm')")
    if m.embedding_type.lower() != "positional":
        raise ValueError("only embedding_type 'positional' is supported")
    nf = m.nf
    ch_mult = list(m.ch_mult)
    nrb = m.num_res_blocks
    attn_res = list(m.attn_resolutions)
    nres = len(ch_mult)
    L = config.data.max_res_num
    resolutions = [L // (2 ** i) for i in range(nres)]

    input_stages: List[Stage] = []
    skip_channels = [nf]
    in_ch = nf

    def attn_pair(prefix, idx0, ch, res):
        return [Layer("attn", f"{prefix}.{idx0}", ch, ch, res_in=res),
                Layer("st", f"{prefix}.{idx0 + 1}", ch, ch, res_in=res)]

    for lvl in range(nres):
        res = resolutions[lvl]
        for _ in range(nrb):
            out_ch = nf * ch_mult[lvl]
            prefix = f"input_blocks.{len(input_stages)}"
            st = Stage(prefix, [Layer("res", f"{prefix}.0", in_ch, out_ch, res_in=res)])
            in_ch = out_ch
            if res in attn_res:
                st.layers += attn_pair(prefix, 1, in_ch, res)
            input_stages.append(st)
            skip_channels.append(in_ch)
        if lvl != nres - 1:
            prefix = f"input_blocks.{len(input_stages)}"
            input_stages.append(Stage(prefix, [Layer("res", f"{prefix}.0", in_ch, in_ch, down=True, res_in=res)]))
            skip_channels.append(in_ch)

    mid_ch = skip_channels[-1]
    low = resolutions[-1]
    mid = Stage("mid_blocks", [Layer("res", "mid_blocks.0", mid_ch, mid_ch, res_in=low)]
                + attn_pair("mid_blocks", 1, mid_ch, low)
                + [Layer("res", "mid_blocks.3", mid_ch, mid_ch, res_in=low)])

    out_stages: List[Stage] = []
    in_ch = mid_ch
    for lvl in reversed(range(nres)):
        res = resolutions[lvl]
        for blk in range(nrb + 1):
            out_ch = nf * ch_mult[lvl]
            prefix = f"out_blocks.{len(out_stages)}"
            skip = skip_channels.pop()
            st = Stage(prefix, [Layer("res", f"{prefix}.0", in_ch + skip, out_ch, res_in=res)], skip_ch=skip)
            in_ch = out_ch
            if res in attn_res:
                st.layers += attn_pair(prefix, 1, in_ch, res)
            if lvl != 0 and blk == nrb:
                st.layers.append(Layer("res", f"{prefix}.{len(st.layers)}", in_ch, in_ch, up=True, res_in=res))
            out_stages.append(st)
    assert not skip_channels
    return Arch(nf=nf, channels=config.data.num_channels, temb_dim=4 * nf, resolutions=resolutions,
                input_stages=input_stages, mid_stage=mid, out_stages=out_stages, final_ch=in_ch)


def _res_params(l: Layer, temb_dim: int) -> List[ParamSpec]:
    p, ci, co = l.prefix, l.in_ch, l.out_ch
    out = [ParamSpec(f"{p}.GroupNorm_0.weight", (ci,), "norm_scale"),
           ParamSpec(f"{p}.GroupNorm_0.bias", (ci,), "norm_shift"),
           ParamSpec(f"{p}.Conv_0.weight", (co, ci, 3, 3), "weight", ci * 9),
           ParamSpec(f"{p}.Conv_0.bias", (co,), "bias"),
           ParamSpec(f"{p}.Dense_0.weight", (co, temb_dim), "weight", temb_dim),
           ParamSpec(f"{p}.Dense_0.bias", (co,), "bias"),
           ParamSpec(f"{p}.GroupNorm_1.weight", (co,), "norm_scale"),
           ParamSpec(f"{p}.GroupNorm_1.bias", (co,), "norm_shift"),
           ParamSpec(f"{p}.Conv_1.weight", (co, co, 3, 3), "weight", co * 9),
           ParamSpec(f"{p}.Conv_1.bias", (co,), "bias")]
    if ci != co or l.up or l.down:
        out += [ParamSpec(f"{p}.Conv_2.weight", (co, ci, 1, 1), "weight", ci),
                ParamSpec(f"{p}.Conv_2.bias", (co,), "bias")]
    return out


def _attn_params(l: Layer) -> List[ParamSpec]:
    p, c = l.prefix, l.in_ch
    out = [ParamSpec(f"{p}.GroupNorm_0.weight", (c,), "norm_scale"),
           ParamSpec(f"{p}.GroupNorm_0.bias", (c,), "norm_shift")]
    for i in range(4):
        out += [ParamSpec(f"{p}.NIN_{i}.W", (c, c), "weight", c),
                ParamSpec(f"{p}.NIN_{i}.b", (c,), "bias")]
    return out


def _st_params(l: Layer, ctx_dim: int) -> List[ParamSpec]:
    p, c = l.prefix, l.in_ch
    t = f"{p}.transformer_blocks.0"
    out = [ParamSpec(f"{p}.norm.weight", (c,), "norm_scale"),
           ParamSpec(f"{p}.norm.bias", (c,), "norm_shift"),
           ParamSpec(f"{p}.proj_in.weight", (c, c, 1, 1), "weight", c),
           ParamSpec(f"{p}.proj_in.bias", (c,), "bias")]
    for a, kd in (("attn1", c),):
        out += [ParamSpec(f"{t}.{a}.to_q.weight", (c, c), "weight", c),
                ParamSpec(f"{t}.{a}.to_k.weight", (c, kd), "weight", kd),
                ParamSpec(f"{t}.{a}.to_v.weight", (c, kd), "weight", kd),
                ParamSpec(f"{t}.{a}.to_out.0.weight", (c, c), "weight", c),
                ParamSpec(f"{t}.{a}.to_out.0.bias", (c,), "bias")]
    out += [ParamSpec(f"{t}.ff.net.0.proj.weight", (8 * c, c), "weight", c),
            ParamSpec(f"{t}.ff.net.0.proj.bias", (8 * c,), "bias"),
            ParamSpec(f"{t}.ff.net.2.weight", (c, 4 * c), "weight", 4 * c),
            ParamSpec(f"{t}.ff.net.2.bias", (c,), "bias")]
    out += [ParamSpec(f"{t}.attn2.to_q.weight", (c, c), "weight", c),
            ParamSpec(f"{t}.attn2.to_k.weight", (c, ctx_dim), "weight", ctx_dim),
            ParamSpec(f"{t}.attn2.to_v.weight", (c, ctx_dim), "weight", ctx_dim),
            ParamSpec(f"{t}.attn2.to_out.0.weight", (c, c), "weight", c),
            ParamSpec(f"{t}.attn2.to_out.0.bias", (c,), "bias")]
    for i in (1, 2, 3):
        out += [ParamSpec(f"{t}.norm{i}.weight", (c,), "norm_scale"),
                ParamSpec(f"{t}.norm{i}.bias", (c,), "norm_shift")]
    out += [ParamSpec(f"{p}.proj_out.weight", (c, c, 1, 1), "weight", c),
            ParamSpec(f"{p}.proj_out.bias", (c,), "bias")]
    return out


def param_specs(config) -> List[ParamSpec]:
    """Learnable tensors of the reference ``UNetModel`` in ``parameters()`` order.

    The float64 buffer ``sigmas`` (ncsnpp.py:78) is not a parameter; it is derived from the
    config (``get_sigmas``, models/utils.py:50-60)."""
    a = build_arch(config)
    nf, td, ch = a.nf, a.temb_dim, a.channels
    specs = [ParamSpec("pre_blocks.0.weight", (td, nf), "weight", nf),
             ParamSpec("pre_blocks.0.bias", (td,), "bias"),
             ParamSpec("pre_blocks.1.weight", (td, td), "weight", td),
             ParamSpec("pre_blocks.1.bias", (td,), "bias"),
             ParamSpec("pre_conv.weight", (nf, ch, 3, 3), "weight", ch * 9),
             ParamSpec("pre_conv.bias", (nf,), "bias")]
    for l in a.all_layers():
        if l.kind == "res":
            specs += _res_params(l, td)
        elif l.kind == "attn":
            specs += _attn_params(l)
        else:
            specs += _st_params(l, config.model.context_dim)
    fc = a.final_ch
    specs += [ParamSpec("out.0.weight", (fc,), "norm_scale"),
              ParamSpec("out.0.bias", (fc,), "norm_shift"),
              ParamSpec("out.2.weight", (ch, fc, 3, 3), "weight", fc * 9),
              ParamSpec("out.2.bias", (ch,), "bias")]
    return specs
