"""Per-operator parity of the HIP kernels (called through the C ABI) against torch fp32 on CPU."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2

pytestmark = pytest.mark.gpu

TOL = {0: 2e-6, 1: 6e-3, 2: 8e-4}     # rel-L2 per op: exact-f32 MFMA, bf16, fp16 operands
TDT = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


@pytest.fixture(scope="module")
def lib():
    from text2protein_amd import _lib
    return _lib.load()


_KEEP = []


def dev(t):
    """Copy to the GPU and keep the tensor alive until the test ends (raw pointers are passed to C)."""
    d = t.contiguous().to("cuda")
    _KEEP.append(d)
    return d


@pytest.fixture(autouse=True)
def _release_device_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def P(t):
    return C.c_void_p(t.data_ptr())


def check(lib, rc):
    assert rc == 0, lib.t2p_last_error().decode()


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 72, 40), (1000, 515, 264), (64, 5, 32), (7, 130, 8),
                                   (512, 256, 128), (1000, 130, 192), (4096, 512, 1024), (300, 64, 72)])
def test_gemm(lib, dt, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = (a.double() @ w.double().T + bias.double() + res.double()) * 0.5
    da, dw = dev(a), dev(w.to(TDT[dt]))
    out = torch.empty(M, N, device="cuda")
    check(lib, lib.t2p_op_gemm(dt, P(da), 1, P(dw), P(out), 1, M, N, K, K, K, N, P(dev(bias)), P(dev(res)), 0.5, None))
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref) < TOL[dt]
    if dt:   # operands already in the compute dtype, compute-dtype output
        da16 = dev(a.to(TDT[dt]))
        out16 = torch.empty(M, N, device="cuda", dtype=TDT[dt])
        check(lib, lib.t2p_op_gemm(dt, P(da16), 0, P(dw), P(out16), 0, M, N, K, K, K, N, None, None, 1.0, None))
        torch.cuda.synchronize()
        assert rel_l2(out16.float().cpu(), a.double() @ w.double().T) < 2 * TOL[dt]


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("B,H,W,Cin,Cout,up", [(2, 16, 16, 32, 64, 0), (1, 8, 12, 96, 32, 0), (2, 8, 8, 64, 40, 1),
                                              (3, 4, 4, 8, 5, 0), (1, 32, 32, 264, 136, 0),
                                              # shapes that take the LDS-DMA kernel (16-bit, Cin % 64 == 0, M >= 256)
                                              (2, 32, 32, 64, 128, 0), (1, 16, 16, 128, 64, 1), (2, 16, 24, 192, 72, 0),
                                              (3, 20, 12, 256, 256, 0), (1, 64, 64, 64, 200, 1)])
def test_conv3x3(lib, dt, B, H, W, Cin, Cout, up):
    g = torch.Generator().manual_seed(H * 31 + Cin)
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, Cin, hs, ws, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    xin = x.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else x
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    dx = dev(x.permute(0, 2, 3, 1))
    dw = dev(w.permute(0, 2, 3, 1).to(TDT[dt]))
    out = torch.empty(B, H, W, Cout, device="cuda")
    check(lib, lib.t2p_op_conv3x3(dt, P(dx), 1, P(dw), P(dev(b)), P(out), B, H, W, Cin, Cout, up, None))
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref) < TOL[dt]
    if dt:   # activations already in the compute dtype (what the engine feeds): LDS-DMA kernel where eligible
        x16 = x.to(TDT[dt])
        xin16 = x16.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else x16
        ref16 = F.conv2d(xin16.double(), w.to(TDT[dt]).double(), b.double(), padding=1).permute(0, 2, 3, 1)
        out2 = torch.full((B, H, W, Cout), float("nan"), device="cuda")
        check(lib, lib.t2p_op_conv3x3(dt, P(dev(x16.permute(0, 2, 3, 1))), 0, P(dw), P(dev(b)), P(out2), B, H, W, Cin, Cout, up, None))
        torch.cuda.synchronize()
        assert rel_l2(out2.cpu(), ref16) < 3e-6      # same rounded operands, fp32 accumulation


@pytest.mark.parametrize("C0,C1,G,silu,down", [(32, 0, 8, 1, 0), (64, 32, 24, 1, 0), (512, 256, 32, 0, 0),
                                               (64, 0, 16, 1, 1), (1024, 1024, 32, 1, 0), (1024, 512, 32, 0, 0)])
@pytest.mark.parametrize("H,W", [(8, 6), (16, 12)])     # <= 64 pixels: single-launch kernel; larger: statistics / finalize / apply
def test_groupnorm(lib, C0, C1, G, silu, down, H, W):
    B = 2
    g = torch.Generator().manual_seed(C0 + C1)
    x = torch.randn(B, C0 + C1, H, W, generator=g) * 2 + 0.5
    gamma = torch.randn(C0 + C1, generator=g)
    beta = torch.randn(C0 + C1, generator=g)
    ref = F.group_norm(x.double(), G, gamma.double(), beta.double(), eps=1e-6)
    if silu:
        ref = F.silu(ref)
    if down:
        ref = ref.reshape(B, C0 + C1, H // 2, 2, W // 2, 2).mean(dim=(3, 5))
    ref = ref.permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1)
    x0 = dev(xn[..., :C0])
    x1 = dev(xn[..., C0:]) if C1 else None
    for dt in (0, 1, 2):
        out = torch.empty(ref.shape, device="cuda", dtype=TDT[dt])
        check(lib, lib.t2p_op_groupnorm(P(x0), P(x1) if C1 else None, C0, C1, B, H, W, G, P(dev(gamma)), P(dev(beta)),
                                        1e-6, silu, down, P(out), dt, None))
        torch.cuda.synchronize()
        assert rel_l2(out.float().cpu(), ref) < (3e-6 if dt == 0 else (5e-3 if dt == 1 else 6e-4))


def test_groupnorm_large_mean(lib):
    """E[x^2]-E[x]^2 is combined in double across blocks: a large mean must not wreck the variance."""
    B, H, W, C, G = 1, 32, 32, 64, 16
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g) + 30.0
    ones, zeros = torch.ones(C), torch.zeros(C)
    ref = F.group_norm(x.double(), G, ones.double(), zeros.double(), eps=1e-6).permute(0, 2, 3, 1)
    out = torch.empty(ref.shape, device="cuda")
    check(lib, lib.t2p_op_groupnorm(P(dev(x.permute(0, 2, 3, 1))), None, C, 0, B, H, W, G, P(dev(ones)), P(dev(zeros)),
                                    1e-6, 0, 0, P(out), 0, None))
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref) < 2e-4


@pytest.mark.parametrize("rows,C", [(10, 64), (1000, 512), (3, 1024)])
def test_layernorm(lib, rows, C):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(rows, C, generator=g) * 3 + 1
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.layer_norm(x.double(), (C,), ga.double(), be.double(), eps=1e-5)
    out = torch.empty(rows, C, device="cuda")
    check(lib, lib.t2p_op_layernorm(P(dev(x)), P(dev(ga)), P(dev(be)), P(out), 0, rows, C, 1e-5, None))
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref) < 2e-6


@pytest.mark.parametrize("C", [128, 512, 1024])
def test_layernorm_16bit_rows(lib, C):
    """LayerNorm on rows stored in 16 bits (the transformer blocks in f16 / bf16 mode); 512 / 1024 channels take the kernel
    that keeps the row in registers (plan switch 20), which must agree with the three-pass kernel to output rounding."""
    g = torch.Generator().manual_seed(C)
    rows = 777
    for dt in (1, 2):
        td = TDT[dt]
        x = (torch.randn(rows, C, generator=g) * 3.0 + 0.7).to(td)
        ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
        ref = F.layer_norm(x.double(), (C,), ga.double(), be.double(), 1e-5)
        outs = []
        try:
            for sw in (1, 0):
                check(lib, lib.t2p_debug_set(20, sw))
                out = torch.full((rows, C), float("nan"), device="cuda", dtype=td)
                check(lib, lib.t2p_op_layernorm16(P(dev(x)), P(dev(ga)), P(dev(be)), P(out), dt, rows, C, 1e-5, None))
                torch.cuda.synchronize()
                outs.append(out.cpu())
        finally:
            lib.t2p_debug_set(20, 1)
        tol = 6e-3 if dt == 1 else 8e-4                    # one rounding of the output to bf16 / f16
        assert rel_l2(outs[0].double(), ref) < tol and rel_l2(outs[1].double(), ref) < tol
        assert rel_l2(outs[0].double(), outs[1].double()) < tol


@pytest.mark.parametrize("rows,n", [(9, 3), (64, 64), (100, 1000), (5, 1500)])
def test_softmax(lib, rows, n):
    g = torch.Generator().manual_seed(n)
    ld = (n + 7) // 8 * 8
    S = torch.randn(rows, ld, generator=g) * 4
    ref = torch.softmax(S[:, :n].double() * 0.3, dim=-1)
    out = torch.full((rows, ld), 7.0, device="cuda")
    check(lib, lib.t2p_op_softmax(P(dev(S)), ld, P(out), ld, 0, rows, n, 0.3, None))
    torch.cuda.synchronize()
    o = out.cpu()
    assert rel_l2(o[:, :n], ref) < 2e-6
    assert float(o[:, n:].abs().max()) == 0.0 if ld > n else True


def test_geglu(lib):
    g = torch.Generator().manual_seed(3)
    u = torch.randn(50, 512, generator=g) * 2
    a, gate = u.double().chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    out = torch.empty(50, 256, device="cuda")
    check(lib, lib.t2p_op_geglu(P(dev(u)), P(out), 0, 50, 256, None))
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref) < 2e-6


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("B,heads,nq,nk,d", [(2, 4, 64, 64, 16), (1, 8, 256, 77, 64), (2, 1, 64, 64, 256), (2, 2, 4, 3, 32),
                                             (2, 8, 1024, 1024, 64), (1, 4, 300, 512, 128), (3, 8, 256, 256, 32),
                                             (2, 8, 16, 16, 64), (1, 2, 130, 65, 64)])
def test_attention(lib, dt, B, heads, nq, nk, d):
    g = torch.Generator().manual_seed(nq + nk)
    C_ = heads * d
    q = torch.randn(B, nq, C_, generator=g)
    k = torch.randn(B, nk, C_, generator=g)
    v = torch.randn(B, nk, C_, generator=g)
    scale = d ** -0.5
    td = TDT[dt]
    qr, kr, vr = (t.to(td).double() for t in (q, k, v))    # reference on the rounded operands
    qh = qr.reshape(B, nq, heads, d).transpose(1, 2)
    kh = kr.reshape(B, nk, heads, d).transpose(1, 2)
    vh = vr.reshape(B, nk, heads, d).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B, nq, C_)
    nkp = (nk + 7) // 8 * 8
    vt = torch.zeros(B, C_, nkp)
    vt[:, :, :nk] = v.transpose(1, 2)
    vt[:, :, nk:] = float("nan") if nkp > nk else 0.0     # padding must never leak into the result
    ws = torch.empty(lib.t2p_op_attention_ws(dt, B, heads, nq, nk), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, nq, C_, device="cuda", dtype=td)
    check(lib, lib.t2p_op_attention(dt, P(dev(q.to(td))), C_, P(dev(k.to(td))), C_, P(dev(vt.to(td))), nkp, P(out), B, heads,
                                    nq, nk, d, scale, P(ws), None))
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref) < (3e-6 if dt == 0 else (8e-3 if dt == 1 else 1e-3))


def test_langevin_and_predictor(lib):
    from oracle import t2p_oracle as O
    g = torch.Generator().manual_seed(11)
    B, Cc, L = 3, 5, 16
    x = torch.randn(B, Cc, L, L, generator=g) * 50
    grad = torch.randn(B, Cc, L, L, generator=g) * 0.02
    noise = torch.randn(B, Cc, L, L, generator=g)
    mask = torch.rand(B, Cc, L, L, generator=g) > 0.3
    x0 = torch.randn(B, Cc, L, L, generator=g)
    rx, rxm = O.langevin_update(x, grad.double(), noise, 0.17)
    rx = torch.where(mask, rx, x0.double())
    ox, oxm = torch.empty_like(x, device="cuda"), torch.empty_like(x, device="cuda")
    sums = torch.zeros(2, device="cuda")
    check(lib, lib.t2p_op_langevin(P(dev(x)), P(dev(grad)), P(dev(noise)), P(dev(mask.to(torch.uint8))), P(dev(x0)), P(ox),
                                   P(oxm), B, Cc * L * L, 0.17, 1.0, P(sums), None))
    torch.cuda.synchronize()
    assert rel_l2(ox.cpu(), rx) < 1e-6 and rel_l2(oxm.cpu(), rxm) < 1e-6
    s = sums.cpu().double()
    assert abs(float(s[0]) - float(grad.reshape(B, -1).norm(dim=-1).sum())) < 1e-4
    assert abs(float(s[1]) / float(noise.reshape(B, -1).norm(dim=-1).sum()) - 1) < 1e-6
    G = torch.full((B,), 3.7)
    for pf in (0, 1):
        rx, rxm = O.reverse_diffusion_update(x, grad.double(), noise, G, bool(pf))
        check(lib, lib.t2p_op_predictor(P(dev(x)), P(dev(grad)), P(dev(noise)), None, None, P(ox), P(oxm), x.numel(), 3.7, pf, None))
        torch.cuda.synchronize()
        assert rel_l2(ox.cpu(), rx) < 1e-6 and rel_l2(oxm.cpu(), rxm) < 1e-6


def test_philox_normal_moments(lib):
    n = 1 << 20
    out = torch.empty(n, device="cuda")
    check(lib, lib.t2p_op_philox_normal(P(out), n, 1234, 1, None))
    out2 = torch.empty(n, device="cuda")
    check(lib, lib.t2p_op_philox_normal(P(out2), n, 1234, 2, None))
    torch.cuda.synchronize()
    z = out.cpu().double()
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1) < 5e-3
    assert abs(float((z ** 4).mean()) - 3) < 0.05
    assert abs(float((z * out2.cpu().double()).mean())) < 5e-3       # streams are independent
    out3 = torch.empty(n, device="cuda")
    check(lib, lib.t2p_op_philox_normal(P(out3), n, 1234, 1, None))
    torch.cuda.synchronize()
    assert torch.equal(out, out3)                                     # counter-based: reproducible


@pytest.mark.parametrize("geom,ring", [(1, 1), (2, 1), (3, 1), (3, 2), (0, 1), (0, 2)])
def test_dma_kernel_variants_agree(lib, geom, ring):
    """Every LDS-DMA geometry (256x128x3, 128x128x2, 256x256 with the 32x32x16 and with the 16x16x32
    MFMA shape) forced on the same operands: conv with halo + up-sampling, ragged M / N GEMM."""
    try:
        check(lib, lib.t2p_debug_set(2, geom))
        check(lib, lib.t2p_debug_set(8, ring))
        g = torch.Generator().manual_seed(geom * 10 + ring)
        for dt in (1, 2):
            td = TDT[dt]
            for (B, H, W, Cin, Cout, up) in [(2, 24, 20, 128, 256, 0), (1, 32, 32, 64, 512, 1), (3, 16, 16, 192, 256, 0)]:
                hs, ws = (H // 2, W // 2) if up else (H, W)
                x = torch.randn(B, Cin, hs, ws, generator=g).to(td)
                w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(td)
                b = torch.randn(Cout, generator=g)
                xin = x.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else x
                ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
                out = torch.full((B, H, W, Cout), float("nan"), device="cuda")
                check(lib, lib.t2p_op_conv3x3(dt, P(dev(x.permute(0, 2, 3, 1))), 0, P(dev(w.permute(0, 2, 3, 1))), P(dev(b)), P(out),
                                              B, H, W, Cin, Cout, up, None))
                torch.cuda.synchronize()
                assert rel_l2(out.cpu(), ref) < 3e-6, (geom, ring, dt, B, H, W, Cin, Cout, up)
            for (M, N, K) in [(1000, 256, 192), (700, 520, 64), (4096, 512, 1024)]:
                a = torch.randn(M, K, generator=g).to(td)
                w = (torch.randn(N, K, generator=g) / K ** 0.5).to(td)
                res = torch.randn(M, N, generator=g)
                ref = (a.double() @ w.double().T + res.double()) * 0.5
                out = torch.full((M, N), float("nan"), device="cuda")
                check(lib, lib.t2p_op_gemm(dt, P(dev(a)), 0, P(dev(w)), P(out), 1, M, N, K, K, K, N, None, P(dev(res)), 0.5, None))
                torch.cuda.synchronize()
                assert rel_l2(out.cpu(), ref) < 3e-6, (geom, ring, dt, M, N, K)
    finally:
        lib.t2p_debug_set(2, 0)
        lib.t2p_debug_set(8, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("geom,ring", [(3, 2), (3, 1), (1, 0), (6, 2)])
def test_dma_issue_order_does_not_change_results(lib, geom, ring):
    """The staggered DMA issue order of the two wave halves (default) against the same kernel with the
    stagger switched off (debug bits 128 / 256), and the register epilogue of the 16x16x32 kernels against the
    LDS-staged one (bit 4096): only the instruction order / the way the values reach memory differs, so bit-identical."""
    try:
        check(lib, lib.t2p_debug_set(2, geom))
        check(lib, lib.t2p_debug_set(8, ring))
        g = torch.Generator().manual_seed(77)
        B, H, W, Cin, Cout = 2, 40, 36, 256, 256
        x = dev(torch.randn(B, H, W, Cin, generator=g).half())
        w = dev((torch.randn(Cout, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).half())
        b = dev(torch.randn(Cout, generator=g))
        outs = []
        for mask in (0, 128, 256, 4096):
            check(lib, lib.t2p_debug_set(1, mask))
            out = torch.full((B, H, W, Cout), float("nan"), device="cuda")
            check(lib, lib.t2p_op_conv3x3(2, P(x), 0, P(w), P(b), P(out), B, H, W, Cin, Cout, 0, None))
            torch.cuda.synchronize()
            outs.append(out.cpu())
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3])
        # GEMM epilogue variants (16-bit output with a 16-bit residual, fp32 output with an fp32 residual, ragged N % 8 == 0)
        for (M, N, K, c_f32) in [(1000, 264, 192, 0), (3000, 512, 128, 1)]:
            a = dev(torch.randn(M, K, generator=g).half())
            wt = dev((torch.randn(N, K, generator=g) / K ** 0.5).half())
            bias = dev(torch.randn(N, generator=g))
            res = torch.randn(M, N, generator=g)
            res = dev(res if c_f32 else res.half())
            outs = []
            for mask in (0, 4096):
                check(lib, lib.t2p_debug_set(1, mask))
                out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if c_f32 else torch.float16)
                fn = lib.t2p_op_gemm if c_f32 else lib.t2p_op_gemm_r16
                check(lib, fn(2, P(a), 0, P(wt), P(out), c_f32, M, N, K, K, K, N, P(bias), P(res), 0.5, None))
                torch.cuda.synchronize()
                outs.append(out.cpu())
            assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1]), (M, N, K, c_f32)
    finally:
        lib.t2p_debug_set(1, 0)
        lib.t2p_debug_set(2, 0)
        lib.t2p_debug_set(8, 2)


def test_conv_midsize_splitk_plan(lib):
    """M = 8192, N = 512, K = 9 x 256 with a split-K workspace attached (development key 10): planned as
    64 tiles of 256x256 x 4 K-splits + the vectorised second pass (bias applied there); fp64 reference."""
    try:
        check(lib, lib.t2p_debug_set(10, 256))
        g = torch.Generator().manual_seed(123)
        B, H, W, Cin, Cout = 32, 16, 16, 256, 512
        for dt in (1, 2):
            td = TDT[dt]
            x = torch.randn(B, Cin, H, W, generator=g).to(td)
            w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(td)
            b = torch.randn(Cout, generator=g)
            ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
            outs = []
            for mid in (1, 0):
                check(lib, lib.t2p_debug_set(12, mid))
                out = torch.full((B, H, W, Cout), float("nan"), device="cuda")
                check(lib, lib.t2p_op_conv3x3(dt, P(dev(x.permute(0, 2, 3, 1))), 0, P(dev(w.permute(0, 2, 3, 1))), P(dev(b)), P(out),
                                              B, H, W, Cin, Cout, 0, None))
                torch.cuda.synchronize()
                assert rel_l2(out.cpu(), ref) < 3e-6, (dt, mid)
                outs.append(out.cpu())
            assert not torch.equal(outs[0], outs[1])     # the two plans sum in different orders
    finally:
        lib.t2p_debug_set(10, 0)
        lib.t2p_debug_set(12, 0)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("M,N,K,ws", [(4096, 512, 1024, 0),     # LDS-DMA kernel, lean epilogue
                                      (1000, 130, 192, 0),      # ragged: generic epilogue
                                      (300, 72, 40, 0),         # register-staged kernel
                                      (512, 256, 2304, 64)])    # split-K: vectorised second pass
def test_gemm_16bit_residual(lib, dt, M, N, K, ws):
    """t2p_op_gemm_r16: the residual operand stored in the compute dtype (the f16-mode residual stream)."""
    td = TDT[dt]
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(td)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(td)
    bias = torch.randn(N, generator=g)
    res = (torch.randn(M, N, generator=g) * 3).to(td)
    ref = (a.double() @ w.double().T + bias.double() + res.double()) * 0.5
    try:
        if ws:
            check(lib, lib.t2p_debug_set(10, ws))
        for c_f32 in (1, 0):
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if c_f32 else td)
            check(lib, lib.t2p_op_gemm_r16(dt, P(dev(a)), 0, P(dev(w)), P(out), c_f32, M, N, K, K, K, N, P(dev(bias)), P(dev(res)), 0.5, None))
            torch.cuda.synchronize()
            tol = 3e-6 if c_f32 else 2 * TOL[dt]
            assert rel_l2(out.float().cpu(), ref) < tol, (dt, M, N, K, c_f32)
    finally:
        lib.t2p_debug_set(10, 0)


@pytest.mark.parametrize("geom", [0, 1, 2, 3, 4])
def test_conv3x3_with_shortcut_segment(lib, geom):
    """The second convolution of a residual block with the block's 1x1 shortcut as an extra K segment of the same launch
    (centre tap of x0 | x1; layers.py:322-327: h + Conv_2(x), then / sqrt 2): every LDS-DMA geometry, one and two shortcut
    sources, 16-bit and fp32 output, with and without the split-K plan (whose splits may start inside the segment)."""
    try:
        check(lib, lib.t2p_debug_set(2, geom))
        g = torch.Generator().manual_seed(900 + geom)
        for dt in (1, 2):
            td = TDT[dt]
            for (B, H, W, C, CX0, CX1, Cout, ws) in [(2, 24, 20, 128, 64, 0, 256, 0), (1, 32, 32, 64, 128, 64, 128, 0),
                                                    (3, 16, 16, 256, 256, 128, 256, 64), (2, 8, 8, 128, 192, 64, 128, 64)]:
                check(lib, lib.t2p_debug_set(10, ws))
                a = torch.randn(B, C, H, W, generator=g).to(td)
                x = torch.randn(B, CX0 + CX1, H, W, generator=g).to(td)
                w1 = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(td)
                w2 = (torch.randn(Cout, CX0 + CX1, 1, 1, generator=g) / (CX0 + CX1) ** 0.5).to(td)
                b = torch.randn(Cout, generator=g)
                ref = (F.conv2d(a.double(), w1.double(), b.double(), padding=1) + F.conv2d(x.double(), w2.double())).permute(0, 2, 3, 1) * 0.5 ** 0.5
                wcat = torch.cat([w1.permute(0, 2, 3, 1).reshape(Cout, 9 * C), w2.reshape(Cout, CX0 + CX1)], 1).contiguous()
                xn = x.permute(0, 2, 3, 1)
                x0 = dev(xn[..., :CX0])
                x1 = dev(xn[..., CX0:]) if CX1 else None
                for c_f32 in (1, 0):
                    out = torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=torch.float32 if c_f32 else td)
                    check(lib, lib.t2p_op_conv3x3_shortcut(dt, P(dev(a.permute(0, 2, 3, 1))), P(dev(wcat)), P(dev(b)), P(x0), CX0,
                                                           P(x1) if CX1 else None, CX1, 0.5 ** 0.5, P(out), c_f32, B, H, W, C, Cout, None))
                    torch.cuda.synchronize()
                    tol = 3e-6 if c_f32 else (6e-4 if dt == 2 else 5e-3)
                    assert rel_l2(out.float().cpu(), ref) < tol, (geom, dt, B, H, W, C, CX0, CX1, Cout, ws, c_f32)
    finally:
        lib.t2p_debug_set(2, 0)
        lib.t2p_debug_set(10, 0)
    # shapes that are not on the LDS-DMA convolution are refused, not silently computed without the segment
    z = torch.zeros(1, 8, 8, 32, device="cuda", dtype=torch.float16)
    assert lib.t2p_op_conv3x3_shortcut(2, P(z), P(z), None, P(z), 32, None, 0, 1.0, P(z), 0, 1, 8, 8, 32, 32, None) != 0


@pytest.mark.parametrize("C,nf,H,W", [(5, 256, 8, 128), (5, 128, 8, 64), (8, 128, 4, 128), (5, 64, 8, 64), (5, 256, 6, 40), (8, 256, 18, 64)])
def test_input_conv_and_its_column_statistics(lib, C, nf, H, W):
    """pre_conv (ncsnpp.py:230) from the NCHW fp32 sample, fp32 arithmetic; where the shape allows it the kernel also emits
    the per-64-pixel column sums the first GroupNorm consumes: the same output bits with and without them, and the sums equal
    those of the fp32 results."""
    B = 2
    g = torch.Generator().manual_seed(C * 1000 + nf + W)
    x = torch.randn(B, C, H, W, generator=g) * 30.0
    w = torch.randn(nf, C, 3, 3, generator=g) / (9 * C) ** 0.5
    b = torch.randn(nf, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)       # NHWC
    w_tcn = dev(w.permute(2, 3, 1, 0).reshape(9, C, nf).contiguous())
    out32 = torch.full((B, H, W, nf), float("nan"), device="cuda")
    check(lib, lib.t2p_op_input_conv(P(dev(x)), P(w_tcn), P(dev(b)), P(out32), 0, B, C, H, W, nf, None, None))
    torch.cuda.synchronize()
    assert rel_l2(out32.cpu(), ref) < 1e-6
    # 16-bit output: on the 16-bit matrix pipe with every fp32 operand split into two f16 terms (plan switch 38; W % 64 == 0 and
    # an even H: the form the engine runs -- its column sums, taken before the rounding, are the evidence of fp32-class accuracy),
    # on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, plan switch 26; W % 64 == 0) and on the FMA kernel
    try:
        for split, sw in ((1, 1), (0, 1), (0, 0)):
            check(lib, lib.t2p_debug_set(38, split))
            check(lib, lib.t2p_debug_set(26, sw))
            out16 = torch.full((B, H, W, nf), float("nan"), device="cuda", dtype=torch.float16)
            check(lib, lib.t2p_op_input_conv(P(dev(x)), P(w_tcn), P(dev(b)), P(out16), 2, B, C, H, W, nf, None, None))
            torch.cuda.synchronize()
            assert rel_l2(out16.float().cpu(), ref) < 3e-4          # fp32 result rounded once to f16
            if W % 64 == 0:
                cs = torch.full((B * H * W // 64, nf, 2), float("nan"), device="cuda")
                o2 = torch.full_like(out16, float("nan"))
                check(lib, lib.t2p_op_input_conv(P(dev(x)), P(w_tcn), P(dev(b)), P(o2), 2, B, C, H, W, nf, P(cs), None))
                torch.cuda.synchronize()
                assert torch.equal(o2.cpu(), out16.cpu())
                chunks = out32.cpu().double().reshape(-1, 64, nf)
                assert rel_l2(cs[..., 0].cpu(), chunks.sum(1)) < 1e-6 and rel_l2(cs[..., 1].cpu(), (chunks ** 2).sum(1)) < 1e-6
            else:   # refused, not silently wrong
                cs = torch.zeros(B * H * W // 64 + 1, nf, 2, device="cuda")
                assert lib.t2p_op_input_conv(P(dev(x)), P(w_tcn), P(dev(b)), P(out16), 2, B, C, H, W, nf, P(cs), None) != 0
    finally:
        lib.t2p_debug_set(26, 1)
        lib.t2p_debug_set(38, 1)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("B,n,mode", [(3, 256, "entry"), (2, 64, "entry"), (5, 32, "normed"), (32, 256, "entry"), (1, 1024, "normed"),
                                      (3, 256, "mid"), (32, 256, "mid"), (4, 64, "mid"), (32, 16, "mid"), (32, 16, "normed"), (3, 256, "tail"), (32, 256, "tail"),
                                      (3, 256, "tail3"), (32, 256, "tail3"), (6, 64, "tail3")])
def test_spatial_transformer_row_chain(lib, dt, B, n, mode):
    """t2p_op_st_entry: the row-wise chains of a SpatialTransformer block in one launch over 32-row blocks, against the same chain
    in fp64 with the intermediate roundings of the separate launches (a, t, LayerNorm(t) stored in the compute dtype).
    entry: GroupNorm (from the producer's column sums) -> proj_in -> LayerNorm_1 -> q | k | v (model/attention.py:250-256, 208-213);
    normed: the same with an already normalised input; mid: t += to_out(o) + b -> LayerNorm_2 -> to_q, in place (:211-213, 186-193);
    tail: t += to_out(o) + b -> LayerNorm_3 -> ff.net.0 with the GEGLU epilogue (rows interleaved (value, gate), :37-64, 214);
    tail3: the same followed by y = [g | t] W_3^T + b_3 + x (ff.net.2 and proj_out as one matrix, :213-215, 259-263) and the per-64-row
    column sums of y (accumulated by pairs of workgroups)."""
    C, G = 256, 32                    # (the C = 512 instantiation was removed in round 4: equal to the separate launches, twice)
    td = TDT[dt]
    g = torch.Generator().manual_seed(17 * n + B + len(mode) + C)
    x = (torch.randn(B, n, C, generator=g) * 1.5 + 0.3 * torch.randn(B, 1, C, generator=g)).to(td)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    w_in = (torch.randn(C, C, generator=g) / C ** 0.5).to(td)
    b_in = 0.3 * torch.randn(C, generator=g)
    lg, lb = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    tail3 = mode == "tail3"
    if tail3:
        mode = "tail"
    n2 = {"mid": C, "tail": 8 * C}.get(mode, 3 * C)
    w_2 = (torch.randn(n2, C, generator=g) / C ** 0.5).to(td)
    b_2 = 0.3 * torch.randn(n2, generator=g) if mode == "tail" else None
    res = (torch.randn(B, n, C, generator=g) * 2).to(td) if mode in ("mid", "tail") else None
    xd = x.double()
    csp = None
    if mode == "entry":
        xg = xd.reshape(B, n, G, C // G)
        mean = xg.mean(dim=(1, 3), keepdim=True)
        var = xg.var(dim=(1, 3), unbiased=False, keepdim=True)
        a = (((xg - mean) / torch.sqrt(var + 1e-6)).reshape(B, n, C) * gamma.double() + beta.double()).to(td)
        chunks = x.float().reshape(B * n // 64, 64, C)
        cs = torch.stack([chunks.sum(1), (chunks ** 2).sum(1)], dim=-1).contiguous()          # [B n / 64][C][2]
        csp = P(dev(cs))
    else:
        a = x
    t_ref = (a.double() @ w_in.double().T + b_in.double() + (res.double() if res is not None else 0.0)).to(td)
    td_ = t_ref.double()
    ln = ((td_ - td_.mean(-1, keepdim=True)) / torch.sqrt(td_.var(-1, unbiased=False, keepdim=True) + 1e-5) * lg.double() + lb.double()).to(td)
    out2_ref = ln.double() @ w_2.double().T
    if mode == "tail":
        u = out2_ref + b_2.double()
        out2_ref = u[..., 0::2] * torch.nn.functional.gelu(u[..., 1::2])
    # mid: the residual stream is updated in place
    t = dev(res).clone() if res is not None else torch.full((B, n, C), float("nan"), device="cuda", dtype=td)
    out2 = torch.full((B, n, n2 // 2 if mode == "tail" else n2), float("nan"), device="cuda", dtype=td)
    w_3 = b_3 = res3 = y = ys = None
    if tail3:
        w_3 = (torch.randn(C, 5 * C, generator=g) / (5 * C) ** 0.5).to(td)
        b_3 = 0.3 * torch.randn(C, generator=g)
        res3 = (torch.randn(B, n, C, generator=g) * 2).to(td)
        y = torch.full((B, n, C), float("nan"), device="cuda", dtype=td)
        ys = torch.full((B * n // 64, C, 2), float("nan"), device="cuda")
    args = lambda CC, nn: (dt, P(dev(x)), csp, G, P(dev(gamma)), P(dev(beta)), 1e-6, P(dev(w_in)), P(dev(b_in)), P(t) if res is not None else None,
                           P(dev(lg)), P(dev(lb)), 1e-5, P(dev(w_2)), n2, P(dev(b_2)) if b_2 is not None else None, int(mode == "tail"),
                           P(t), None if tail3 else P(out2), P(dev(w_3)) if tail3 else None, P(dev(b_3)) if tail3 else None,
                           P(dev(res3)) if tail3 else None, P(y) if tail3 else None, P(ys) if tail3 else None, B, nn, CC, None)
    check(lib, lib.t2p_op_st_entry(*args(C, n)))
    torch.cuda.synchronize()
    tol = 1.5e-3 if dt == 2 else 1.2e-2
    assert rel_l2(t.float().cpu(), t_ref.double()) < tol
    if tail3:
        y_ref = torch.cat([out2_ref.to(td).double(), t_ref.double()], dim=-1) @ w_3.double().T + b_3.double() + res3.double()
        assert rel_l2(y.float().cpu(), y_ref) < 2 * tol
        ch = y_ref.reshape(B * n // 64, 64, C)
        assert rel_l2(ys[..., 0].cpu(), ch.sum(1)) < 2 * tol and rel_l2(ys[..., 1].cpu(), (ch ** 2).sum(1)) < 2 * tol
    else:
        assert rel_l2(out2.float().cpu(), out2_ref) < 2 * tol
    # refused, not silently computed by something else: other channel counts, ragged row blocks
    if C == 256:
        assert lib.t2p_op_st_entry(*args(128, n)) != 0
        if csp is not None or tail3:
            assert lib.t2p_op_st_entry(*args(C, 48)) != 0         # per-sample statistics need whole 32- / 64-row blocks per sample
        elif (B * 40) % 32 != 0:
            assert lib.t2p_op_st_entry(*args(C, 40)) != 0         # (without them any split of the rows will do: B n % 32 == 0)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("B,n,with_stats", [(3, 256, True), (32, 256, True), (2, 64, False), (5, 32, False)])
def test_attention_block_projections_in_one_launch(lib, dt, B, n, with_stats):
    """t2p_op_attn_proj: GroupNorm -> q | k (+ bias) and the value projection written transposed, one launch over 32-row blocks
    (AttnBlockpp, layers.py:160-167), against fp64 with the roundings of the separate launches."""
    Cc, G = 256, 32
    td = TDT[dt]
    g = torch.Generator().manual_seed(7 * n + B)
    x = (torch.randn(B, n, Cc, generator=g) * 1.5 + 0.3 * torch.randn(B, 1, Cc, generator=g)).to(td)
    gamma, beta = 1 + 0.2 * torch.randn(Cc, generator=g), 0.2 * torch.randn(Cc, generator=g)
    w_qk = (torch.randn(2 * Cc, Cc, generator=g) / Cc ** 0.5).to(td)
    b_qk = 0.3 * torch.randn(2 * Cc, generator=g)
    w_v = (torch.randn(Cc, Cc, generator=g) / Cc ** 0.5).to(td)
    csp = None
    if with_stats:
        xg = x.double().reshape(B, n, G, Cc // G)
        mean, var = xg.mean(dim=(1, 3), keepdim=True), xg.var(dim=(1, 3), unbiased=False, keepdim=True)
        a = (((xg - mean) / torch.sqrt(var + 1e-6)).reshape(B, n, Cc) * gamma.double() + beta.double()).to(td)
        chunks = x.float().reshape(B * n // 64, 64, Cc)
        csp = P(dev(torch.stack([chunks.sum(1), (chunks ** 2).sum(1)], dim=-1).contiguous()))
    else:
        a = x
    qk_ref = a.double() @ w_qk.double().T + b_qk.double()
    vt_ref = (a.double() @ w_v.double().T).transpose(1, 2)                      # [B][C][n]
    npad = n + 8
    qk = torch.full((B, n, 2 * Cc), float("nan"), device="cuda", dtype=td)
    vt = torch.full((B, Cc, npad), float("nan"), device="cuda", dtype=td)
    check(lib, lib.t2p_op_attn_proj(dt, P(dev(x)), csp, G, P(dev(gamma)), P(dev(beta)), 1e-6, P(dev(w_qk)), P(dev(b_qk)), P(dev(w_v)), P(qk), P(vt),
                                    None, npad, B, n, Cc, None))
    torch.cuda.synchronize()
    tol = 1.5e-3 if dt == 2 else 1.2e-2
    assert rel_l2(qk.float().cpu(), qk_ref) < tol
    assert rel_l2(vt[..., :n].float().cpu(), vt_ref) < tol
    assert torch.isnan(vt[..., n:].float()).all()                              # the padding columns are not touched
    # the same with K and V^T fragment-major (the order attn_strip_kernel<.., FM> streams): the same values in another place
    qk2 = torch.full((B, n, 2 * Cc), float("nan"), device="cuda", dtype=td)
    kf = torch.full((B, n * Cc), float("nan"), device="cuda", dtype=td)
    vf = torch.full((B, Cc * n), float("nan"), device="cuda", dtype=td)
    check(lib, lib.t2p_op_attn_proj(dt, P(dev(x)), csp, G, P(dev(gamma)), P(dev(beta)), 1e-6, P(dev(w_qk)), P(dev(b_qk)), P(dev(w_v)), P(qk2), P(vf),
                                    P(kf), n, B, n, Cc, None))
    torch.cuda.synchronize()
    assert torch.equal(qk2[..., :Cc].cpu(), qk[..., :Cc].cpu())
    assert torch.equal(kf.cpu(), _frag_major(qk[..., Cc:].cpu().contiguous()))
    assert torch.equal(vf.cpu(), _frag_major(vt[..., :n].cpu().contiguous()))


def _frag_major(x):
    """[..., R, Ccols] -> the fragment-major order of GemmParams::c_frag (flattened last two dimensions)."""
    R, Cc = x.shape[-2:]
    r = torch.arange(R)[:, None].expand(R, Cc)
    c = torch.arange(Cc)[None, :].expand(R, Cc)
    idx = ((r // 32) * (Cc // 32) + c // 32) * 1024 + ((c // 8) % 2) * 512 + ((c // 16) % 2) * 256 + (r % 32) * 8 + c % 8
    out = torch.empty(x.shape[:-2] + (R * Cc,), dtype=x.dtype)
    out[..., idx.reshape(-1)] = x.reshape(x.shape[:-2] + (R * Cc,))
    return out


@pytest.mark.parametrize("dt", [1, 2])
def test_gemm_fragment_major_output(lib, dt):
    """GemmParams::c_frag: the q | k projection of an AttnBlockpp writes its k columns fragment-major per sample, the transposed value
    projection (batched) all of them -- against the row-major product, element for element."""
    td = TDT[dt]
    g = torch.Generator().manual_seed(5)
    # q | k: 16 samples x 1024 tokens, C = 512
    Bn, n, C = 16, 1024, 512
    a = torch.randn(Bn * n, C, generator=g).to(td)
    w = (torch.randn(2 * C, C, generator=g) / C ** 0.5).to(td)
    ref = torch.full((Bn * n, 2 * C), float("nan"), device="cuda", dtype=td)          # the same product, row-major, by the same kernel family
    check(lib, lib.t2p_op_gemm(dt, P(dev(a)), 0, P(dev(w)), P(ref), 0, Bn * n, 2 * C, C, C, C, 2 * C, None, None, 1.0, None))
    out = torch.full((Bn * n, 2 * C), float("nan"), device="cuda", dtype=td)
    kf = torch.full((Bn, n * C), float("nan"), device="cuda", dtype=td)
    check(lib, lib.t2p_op_gemm_frag_major(dt, P(dev(a)), P(dev(w)), P(out), P(kf), Bn * n, 2 * C, C, C, n, 1, None))
    torch.cuda.synchronize()
    ref = ref.cpu()
    assert rel_l2(ref.float(), a.double() @ w.double().T) < (1e-3 if dt == 2 else 6e-3)
    assert torch.equal(out[:, :C].cpu(), ref[:, :C])
    assert torch.equal(kf.cpu(), _frag_major(ref[:, C:].reshape(Bn, n, C)))
    # V^T = W h^T per sample: 32 batch entries, [C][n]
    Bz = 32
    h = torch.randn(Bz, n, C, generator=g).to(td)
    wv = (torch.randn(C, C, generator=g) / C ** 0.5).to(td)
    refv = wv.double() @ h.double().transpose(1, 2)                            # [Bz][C][n]
    dummy = torch.full((Bz, C, n), float("nan"), device="cuda", dtype=td)
    vf = torch.full((Bz, C * n), float("nan"), device="cuda", dtype=td)
    check(lib, lib.t2p_op_gemm_frag_major(dt, P(dev(wv)), P(dev(h)), P(dummy), P(vf), C, n, C, 0, 1, Bz, None))
    torch.cuda.synchronize()
    got = vf.cpu().float()
    want = _frag_major(refv)
    assert rel_l2(got, want) < (1e-3 if dt == 2 else 6e-3)
    assert (got - want.float()).abs().max() < (0.02 if dt == 2 else 0.12)      # (a misplaced element would be off by ~1)
    # refused where the plan is another one (few tiles), not silently written row-major
    assert lib.t2p_op_gemm_frag_major(dt, P(dev(a)), P(dev(w)), P(out), P(kf), 4096, 2 * C, C, C, n, 1, None) != 0


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("B,n,d", [(2, 1024, 512), (3, 640, 512), (3, 256, 256), (2, 64, 256)])
def test_wide_head_attention_on_fragment_major_operands(lib, dt, B, n, d):
    """t2p_op_attention_wide_fm: the same kernel reading K and V^T fragment-major: bit-identical to the row-major form."""
    td = TDT[dt]
    g = torch.Generator().manual_seed(n + dt)
    qk = torch.randn(B, n, 2 * d, generator=g).to(td)
    vt = torch.randn(B, d, n, generator=g).to(td)
    bias = torch.randn(d, generator=g)
    res = torch.randn(B, n, d, generator=g).to(td)
    outs = []
    for fm in (0, 1):
        out = torch.full((B, n, d), float("nan"), device="cuda", dtype=td)
        cs = torch.full((B * n // 64, d, 2), float("nan"), device="cuda")
        qd = dev(qk)
        if fm:
            check(lib, lib.t2p_op_attention_wide_fm(dt, P(qd), 2 * d, P(dev(_frag_major(qk[..., d:].contiguous()))), P(dev(_frag_major(vt))), P(out), 0,
                                                    P(dev(bias)), P(dev(res)), 1, 0.7071, P(cs), B, n, d, d ** -0.5, None))
        else:
            check(lib, lib.t2p_op_attention_wide(dt, P(qd), 2 * d, C.c_void_p(qd.data_ptr() + 2 * d), 2 * d, P(dev(vt)), n, P(out), 0, P(dev(bias)),
                                                 P(dev(res)), 1, 0.7071, P(cs), B, n, d, d ** -0.5, None))
        torch.cuda.synchronize()
        outs.append((out.cpu(), cs.cpu()))
    assert torch.isfinite(outs[0][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_input_conv_split_operands_keep_fp32_accuracy(lib):
    """The split form of pre_conv (x = hi + lo / 2048 in f16, three partial products on v_mfma_f32_16x16x32_f16) at the
    magnitudes of a VE run: a prior sample (sigma_max = 100), a late sample (values in [-1, 1] + 0.01 noise), exact zeros of a
    masked region and tiny values below the f16 normal range.  Column sums (fp32, before the output rounding) against fp64."""
    B, C, H, W, nf = 2, 5, 4, 128, 256
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, C, H, W, generator=g)
    x[0] *= 100.0
    x[1, :, :2] = x[1, :, :2].clamp(-1, 1) + 0.01 * torch.randn(C, 2, W, generator=g)
    x[1, :, 2] = 0.0
    x[1, :, 3] *= 1e-6
    w = torch.randn(nf, C, 3, 3, generator=g) / (9 * C) ** 0.5
    b = torch.randn(nf, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, 64, nf)
    w_tcn = dev(w.permute(2, 3, 1, 0).reshape(9, C, nf).contiguous())
    out = torch.full((B, H, W, nf), float("nan"), device="cuda", dtype=torch.float16)
    cs = torch.full((B * H * W // 64, nf, 2), float("nan"), device="cuda")
    check(lib, lib.t2p_op_input_conv(P(dev(x)), P(w_tcn), P(dev(b)), P(out), 2, B, C, H, W, nf, P(cs), None))
    torch.cuda.synchronize()
    per_chunk = (cs[..., 0].cpu().double() - ref.sum(1)).abs().amax(1) / ref.abs().sum(1).amax(1)
    assert per_chunk.max() < 2e-6, per_chunk
    assert rel_l2(cs[..., 1].cpu(), (ref ** 2).sum(1)) < 2e-6
    assert rel_l2(out.float().cpu().reshape(-1, 64, nf), ref) < 3e-4


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("B,heads,n,d", [(2, 8, 1024, 64), (1, 4, 300, 128), (3, 8, 256, 32), (2, 8, 16, 64), (1, 2, 130, 64), (2, 8, 64, 64)])
def test_self_attention_on_stacked_qkv(lib, dt, B, heads, n, d):
    """t2p_op_attention_qkv: q | k | v as three column blocks of one projection output; V is consumed row-major through the
    transposing LDS read (no V^T projection).  Against the softmax(q k^T) v of the rounded operands, ragged key counts included,
    and equal to the V^T form of the same kernel within rounding of nothing (the same products in the same order)."""
    g = torch.Generator().manual_seed(n * 7 + d)
    C_ = heads * d
    td = TDT[dt]
    qkv = torch.randn(B, n, 3 * C_, generator=g).to(td)
    q, k, v = (qkv[..., i * C_:(i + 1) * C_] for i in range(3))
    scale = d ** -0.5
    qh, kh, vh = (t.double().reshape(B, n, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B, n, C_)
    out = torch.full((B, n, C_), float("nan"), device="cuda", dtype=td)
    check(lib, lib.t2p_op_attention_qkv(dt, P(dev(qkv)), 3 * C_, P(out), B, heads, n, d, scale, None))
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref) < (8e-3 if dt == 1 else 1e-3)
    npad = (n + 7) // 8 * 8
    vt = torch.zeros(B, C_, npad, dtype=td)
    vt[:, :, :n] = v.transpose(1, 2)
    ws = torch.empty(lib.t2p_op_attention_ws(dt, B, heads, n, n), dtype=torch.uint8, device="cuda")
    out2 = torch.empty(B, n, C_, device="cuda", dtype=td)
    check(lib, lib.t2p_op_attention(dt, P(dev(q.contiguous())), C_, P(dev(k.contiguous())), C_, P(dev(vt)), npad, P(out2), B, heads,
                                    n, n, d, scale, P(ws), None))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), out2.cpu())


def test_conv3x3_followed_by_groupnorm_in_the_second_pass(lib):
    """Low-resolution 3x3 convolutions run with the K loop split over workgroups; the pass that sums the partial tiles also applies
    the GroupNorm (+SiLU) that follows in ResnetBlockBigGANpp.forward (layers.py:304-321) -- t2p_op_conv3x3_groupnorm.  Against
    torch fp64 on the same 16-bit operands: the raw output (with time-embedding bias, residual, 1/sqrt 2), its normalised +
    activated form, and the per-64-row column sums a later GroupNorm over a concatenation reads."""
    try:
        check(lib, lib.t2p_debug_set(10, 256))
        g = torch.Generator().manual_seed(77)
        cases = [(4, 16, 16, 256, 256, 32, 1, True, 0), (3, 8, 8, 128, 128, 32, 1, True, 0), (8, 4, 4, 256, 256, 32, 0, False, 0),
                 (2, 16, 16, 512, 256, 32, 1, True, 0), (2, 8, 8, 256, 64, 16, 0, False, 0), (2, 16, 16, 128, 256, 32, 1, False, 1)]
        for dt in (2, 1):
            td = TDT[dt]
            for (B, H, W, Cin, Cout, G, silu, with_res, up) in cases:
                Hs, Ws = (H // 2, W // 2) if up else (H, W)
                a = torch.randn(B, Cin, Hs, Ws, generator=g).to(td)
                w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(td)
                bias, tb = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
                res = torch.randn(B, Cout, H, W, generator=g).to(td) if with_res else None
                gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
                alpha = 0.5 ** 0.5 if with_res else 1.0
                src = F.interpolate(a.double(), scale_factor=2, mode="nearest") if up else a.double()
                raw = F.conv2d(src, w.double(), bias.double(), padding=1) + tb.double()[:, :, None, None]
                if with_res:
                    raw = raw + res.double()
                raw = raw * alpha
                normed = F.group_norm(raw, G, gamma.double(), beta.double(), eps=1e-6)
                if silu:
                    normed = F.silu(normed)
                wk = dev(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin))
                for out_f32 in (1, 0, None):                       # None: the raw product is not wanted (GroupNorm_1's input)
                    out = None if out_f32 is None else torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else td)
                    nrm = torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=td)
                    want_cs = (H * W) % 64 == 0 and out_f32 is not None
                    cs = torch.full((B * H * W // 64, Cout, 2), float("nan"), device="cuda") if want_cs else None
                    check(lib, lib.t2p_op_conv3x3_groupnorm(dt, P(dev(a.permute(0, 2, 3, 1))), P(wk), P(dev(bias)), P(dev(tb)),
                                                            P(dev(res.permute(0, 2, 3, 1))) if with_res else None, alpha, up, G, P(dev(gamma)),
                                                            P(dev(beta)), 1e-6, silu, P(out) if out is not None else None, int(bool(out_f32)),
                                                            P(nrm), P(cs) if want_cs else None, B, H, W, Cin, Cout, None))
                    torch.cuda.synchronize()
                    key = (dt, B, H, W, Cin, Cout, G, silu, with_res, up, out_f32)
                    e_n = rel_l2(nrm.float().cpu(), normed.permute(0, 2, 3, 1))
                    assert e_n < (6e-4 if dt == 2 else 5e-3), (key, e_n)          # fp32 arithmetic, one rounding to 16 bits
                    if out is not None:
                        e_r = rel_l2(out.float().cpu(), raw.permute(0, 2, 3, 1))
                        assert e_r < (3e-6 if out_f32 else (6e-4 if dt == 2 else 5e-3)), (key, e_r)
                    if want_cs:
                        r64 = raw.permute(0, 2, 3, 1).reshape(B * H * W // 64, 64, Cout)
                        ref_cs = torch.stack([r64.sum(1), (r64 * r64).sum(1)], -1)
                        assert rel_l2(cs.cpu(), ref_cs) < 1e-5, key
        # a launch that does not take the split-K plan is refused (no workspace here), not computed without the norm
        check(lib, lib.t2p_debug_set(10, 0))
        z = torch.zeros(1, 8, 8, 64, device="cuda", dtype=torch.float16)
        zw = torch.zeros(64, 9 * 64, device="cuda", dtype=torch.float16)
        one = torch.ones(64, device="cuda")
        assert lib.t2p_op_conv3x3_groupnorm(2, P(z), P(zw), None, None, None, 1.0, 0, 16, P(one), P(one), 1e-6, 1, None, 0, P(z), None,
                                            1, 8, 8, 64, 64, None) != 0
    finally:
        lib.t2p_debug_set(10, 0)


@pytest.mark.parametrize("dt", [2, 1])
@pytest.mark.parametrize("B,n,d", [(3, 16, 256), (2, 64, 512), (2, 256, 256), (2, 256, 512), (1, 1024, 512), (2, 1024, 256), (1, 512, 1024),
                                   (2, 64, 1024), (2, 40, 512), (1, 200, 256), (1, 520, 512), (1, 8, 256)])
def test_wide_head_attention_in_one_launch(lib, dt, B, n, d):
    """AttnBlockpp's attention (layers.py:160-176: ONE head, d = C, all h w pixels as keys): attn_strip_kernel against torch fp64
    on the rounded operands, scores of realistic spread (|s| up to ~10 after scaling), NaN in the padding columns of V^T, and
    against the unfused GEMM -> softmax -> GEMM path (plan switch 29) on the same inputs."""
    g = torch.Generator().manual_seed(n * 7 + d)
    td = TDT[dt]
    q = torch.randn(B, n, d, generator=g) * 1.7
    k = torch.randn(B, n, d, generator=g) * 1.7
    v = torch.randn(B, n, d, generator=g)
    scale = d ** -0.5
    qr, kr, vr = (t.to(td).double() for t in (q, k, v))
    ref = torch.softmax(qr @ kr.transpose(-1, -2) * scale, dim=-1) @ vr
    npad = (n + 7) // 8 * 8 + 8                                   # a row stride larger than n: the columns beyond n must never be read
    vt = torch.full((B, d, npad), float("nan"))
    vt[:, :, :n] = v.transpose(1, 2)
    # q | k interleaved the way the engine holds them: one [rows][2 d] buffer
    qk = torch.cat([q, k], -1).to(td)
    dqk = dev(qk)
    ws = torch.empty(lib.t2p_op_attention_ws(dt, B, 1, n, n), dtype=torch.uint8, device="cuda")
    outs = {}
    try:
        for sw in (1, 0):
            check(lib, lib.t2p_debug_set(29, sw))
            out = torch.full((B, n, d), float("nan"), device="cuda", dtype=td)
            vt_in = vt.clone()
            if sw == 0:
                vt_in[:, :, n:] = 0.0                             # the unfused path multiplies the (zero) padding probabilities
            check(lib, lib.t2p_op_attention(dt, P(dqk), 2 * d, C.c_void_p(dqk.data_ptr() + d * dqk.element_size()), 2 * d,
                                            P(dev(vt_in.to(td))), npad, P(out), B, 1, n, n, d, scale, P(ws), None))
            torch.cuda.synchronize()
            outs[sw] = out.float().cpu()
    finally:
        lib.t2p_debug_set(29, 1)
    tol = 8e-3 if dt == 1 else 1e-3
    e1, e0 = rel_l2(outs[1], ref), rel_l2(outs[0], ref)
    assert torch.isfinite(outs[1]).all() and e1 < tol, (dt, B, n, d, e1, e0)
    assert e1 < 1.5 * e0 + 1e-4, (e1, e0)                          # not less accurate than the unfused path


@pytest.mark.parametrize("dt", [2, 1])
@pytest.mark.parametrize("B,n,d,r16,out_f32", [(2, 64, 256, 1, 0), (2, 256, 512, 1, 0), (1, 1024, 512, 1, 0), (2, 16, 512, 0, 1), (1, 256, 1024, 0, 1),
                                               (3, 64, 512, 1, 1)])
def test_wide_head_attention_with_the_block_tail(lib, dt, B, n, d, r16, out_f32):
    """t2p_op_attention_wide: attention of AttnBlockpp plus what is left of the block's tail once NIN_3 is folded into the value
    projection -- alpha (w v + bias + residual) -- and the per-64-query column statistics of the result, against torch fp64."""
    g = torch.Generator().manual_seed(n + d + r16)
    td = TDT[dt]
    q, k = torch.randn(B, n, d, generator=g) * 1.5, torch.randn(B, n, d, generator=g) * 1.5
    v = torch.randn(B, n, d, generator=g)
    bias = torch.randn(d, generator=g)
    res = torch.randn(B, n, d, generator=g) * 2
    res_in = res.to(td) if r16 else res
    scale, alpha = d ** -0.5, 0.5 ** 0.5
    qr, kr, vr = (t.to(td).double() for t in (q, k, v))
    ref = (torch.softmax(qr @ kr.transpose(-1, -2) * scale, dim=-1) @ vr + bias.double() + res_in.double()) * alpha
    npad = (n + 7) // 8 * 8
    vt = v.transpose(1, 2).contiguous()
    out = torch.full((B, n, d), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else td)
    want_cs = n % 64 == 0
    cs = torch.full((B * n // 64, d, 2), float("nan"), device="cuda") if want_cs else None
    check(lib, lib.t2p_op_attention_wide(dt, P(dev(q.to(td))), d, P(dev(k.to(td))), d, P(dev(vt.to(td))), npad, P(out), out_f32, P(dev(bias)),
                                         P(dev(res_in)), r16, alpha, P(cs) if want_cs else None, B, n, d, scale, None))
    torch.cuda.synchronize()
    e = rel_l2(out.float().cpu(), ref)
    assert e < (8e-3 if dt == 1 else 1e-3), e
    if want_cs:
        # the statistics are those of the fp32 values the kernel held (before rounding to 16 bits): compare with the reference's
        r64 = ref.reshape(B * n // 64, 64, d)
        ref_cs = torch.stack([r64.sum(1), (r64 * r64).sum(1)], -1)
        assert rel_l2(cs.cpu(), ref_cs) < (8e-3 if dt == 1 else 1e-3)
        if out_f32:                                   # and exactly consistent with the stored fp32 output
            o64 = out.cpu().double().reshape(B * n // 64, 64, d)
            assert rel_l2(cs.cpu(), torch.stack([o64.sum(1), (o64 * o64).sum(1)], -1)) < 1e-5


@pytest.mark.parametrize("dt", [2, 1])
@pytest.mark.parametrize("B,HWs,C,Cout,CX0,CX1,G,silu,with_res", [(8, 4, 256, 256, 0, 0, 32, 1, False), (4, 4, 512, 256, 256, 256, 32, 1, False),
                                                                   (3, 8, 256, 256, 0, 0, 32, 1, True), (2, 8, 512, 512, 512, 512, 32, 0, False),
                                                                   (4, 4, 64, 32, 0, 0, 4, 1, True), (1, 8, 1024, 512, 0, 0, 32, 1, False),
                                                                   (2, 8, 256, 256, 256, 0, 32, 0, False)])
def test_small_map_convolution_with_its_groupnorm(lib, dt, B, HWs, C, Cout, CX0, CX1, G, silu, with_res):
    """small_conv_gn_kernel (t2p_op_small_conv_groupnorm): a 3x3 convolution of a 4x4 / 8x8 map + 1x1 shortcut segment + biases +
    residual + the GroupNorm (+SiLU) that follows, one launch, against torch fp64 on the same 16-bit operands: the raw output in
    fp32 and in 16 bits, its column statistics, the normalised map, and the norm-only call (raw output not wanted)."""
    g = torch.Generator().manual_seed(B * 100 + C + CX0)
    td = TDT[dt]
    H = W = HWs
    CX = CX0 + CX1
    a = torch.randn(B, C, H, W, generator=g).to(td)
    w1 = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(td)
    x = torch.randn(B, max(CX, 1), H, W, generator=g).to(td)
    w2 = (torch.randn(Cout, max(CX, 1), 1, 1, generator=g) / max(CX, 1) ** 0.5).to(td)
    bias, tb = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g).to(td) if with_res else None
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    alpha = 0.5 ** 0.5
    raw = F.conv2d(a.double(), w1.double(), bias.double(), padding=1) + tb.double()[:, :, None, None]
    if CX:
        raw = raw + F.conv2d(x.double(), w2.double())
    if with_res:
        raw = raw + res.double()
    raw = raw * alpha
    normed = F.group_norm(raw, G, gamma.double(), beta.double(), eps=1e-6)
    if silu:
        normed = F.silu(normed)
    wk = w1.permute(0, 2, 3, 1).reshape(Cout, 9 * C)
    if CX:
        wk = torch.cat([wk, w2.reshape(Cout, CX)], 1)
    wk = dev(wk.contiguous())
    xn = x.permute(0, 2, 3, 1)
    x0 = dev(xn[..., :CX0]) if CX0 else None
    x1 = dev(xn[..., CX0:]) if CX1 else None
    for out_f32 in (1, 0, None):
        out = None if out_f32 is None else torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else td)
        nrm = torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=td)
        want_cs = H * W == 64 and out_f32 is not None
        cs = torch.full((B * H * W // 64, Cout, 2), float("nan"), device="cuda") if want_cs else None
        check(lib, lib.t2p_op_small_conv_groupnorm(dt, P(dev(a.permute(0, 2, 3, 1))), P(wk), wk.shape[1], P(x0) if CX0 else None, CX0,
                                                   P(x1) if CX1 else None, CX1, P(dev(bias)), P(dev(tb)),
                                                   P(dev(res.permute(0, 2, 3, 1))) if with_res else None, alpha, P(out) if out is not None else None,
                                                   int(bool(out_f32)), P(cs) if want_cs else None, P(nrm), G, P(dev(gamma)), P(dev(beta)), 1e-6, silu,
                                                   B, H, W, C, Cout, None))
        torch.cuda.synchronize()
        key = (dt, B, HWs, C, Cout, CX0, CX1, out_f32)
        e_n = rel_l2(nrm.float().cpu(), normed.permute(0, 2, 3, 1))
        assert e_n < (6e-4 if dt == 2 else 5e-3), (key, e_n)
        if out is not None:
            e_r = rel_l2(out.float().cpu(), raw.permute(0, 2, 3, 1))
            assert e_r < (3e-6 if out_f32 else (6e-4 if dt == 2 else 5e-3)), (key, e_r)
        if want_cs:
            r64 = raw.permute(0, 2, 3, 1).reshape(B * H * W // 64, 64, Cout)
            assert rel_l2(cs.cpu(), torch.stack([r64.sum(1), (r64 * r64).sum(1)], -1)) < 1e-5, key
    # raw output only (no norm follows): normed = null
    out = torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=td)
    check(lib, lib.t2p_op_small_conv_groupnorm(dt, P(dev(a.permute(0, 2, 3, 1))), P(wk), wk.shape[1], P(x0) if CX0 else None, CX0,
                                               P(x1) if CX1 else None, CX1, P(dev(bias)), P(dev(tb)),
                                               P(dev(res.permute(0, 2, 3, 1))) if with_res else None, alpha, P(out), 0, None, None, 0, None, None,
                                               1e-6, 0, B, H, W, C, Cout, None))
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), raw.permute(0, 2, 3, 1)) < (6e-4 if dt == 2 else 5e-3)
    # shapes it does not cover are refused
    assert lib.t2p_op_small_conv_groupnorm(dt, P(out), P(wk), wk.shape[1], None, 0, None, 0, None, None, None, 1.0, P(out), 0, None, None, 0, None,
                                           None, 1e-6, 0, 1, 16, 16, C, Cout, None) != 0


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("B,H,W,Cin,Cout,CX", [(4, 128, 128, 128, 256, 0), (8, 128, 128, 64, 128, 0), (32, 64, 64, 128, 128, 0),
                                               (16, 64, 64, 192, 256, 128), (2, 256, 256, 64, 128, 64), (4, 128, 128, 128, 128, 192),
                                               (20, 128, 128, 64, 128, 0), (44, 64, 64, 64, 128, 64)])
def test_dx_shared_stage_convolution_is_bit_identical(lib, dt, B, H, W, Cin, Cout, CX):
    """gemm_dxs_kernel (plan switch 47): the three horizontal taps of a window row read ONE LDS stage with shifted fragment rows, image-row
    edges zeroed in the fragments.  Same K order, same MFMA order, same epilogue as the implicit-GEMM kernel: bit-identical outputs, with and
    without the shortcut segment, on both tile geometries (N = 256: 256 x 256; N = 128: 512 x 128), W = 64 / 128 / 256 -- and both against
    torch in fp64 (every image-row edge of every tile is exercised: the inputs have no zero border).  The last two
    shapes are several rounds of tiles (640 and 352 tiles of 512 rows), their fp64 reference taken on a few samples."""
    td = TDT[dt]
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + CX)
    x = torch.randn(B, H, W, Cin, generator=g).to(td)
    w = (torch.randn(Cout, 9 * Cin + CX, generator=g) / (9 * Cin + CX) ** 0.5).to(td)
    bias = torch.randn(Cout, generator=g)
    xs = torch.randn(B, H, W, CX, generator=g).to(td) if CX else None
    dx_, dw, db = dev(x), dev(w), dev(bias)
    dxs = dev(xs) if CX else None
    outs = []
    try:
        for sw in (1, 1, 1, 0):          # (three runs of the new kernel: a missing wait in its DMA ring shows up as run-to-run differences)
            check(lib, lib.t2p_debug_set(47, sw))
            out = torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=td)
            check(lib, lib.t2p_op_conv3x3_shortcut(dt, P(dx_), P(dw), P(db), P(dxs) if CX else None, CX, None, 0, C.c_float(0.5), P(out), 0,
                                                   B, H, W, Cin, Cout, None))
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        lib.t2p_debug_set(47, 1)
    assert torch.isfinite(outs[0].float()).all() and all(torch.equal(outs[0], o) for o in outs[1:])
    sel = list(range(B)) if B <= 16 else [0, 1, B // 2, B - 2, B - 1]      # (the fp64 reference of the large shapes: a few samples)
    w9 = w[:, :9 * Cin].double().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(x[sel].double().permute(0, 3, 1, 2), w9, bias.double(), padding=1).permute(0, 2, 3, 1)
    if CX:
        ref = ref + xs[sel].double() @ w[:, 9 * Cin:].double().T
    assert rel_l2(outs[0][sel].double(), 0.5 * ref) < TOL[dt]
