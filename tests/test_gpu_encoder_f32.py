"""The reference-precision encoder (memory_2.py:43,738-739: DINOv2 runs f32): the in-tree split-operand MFMA GEMM
(bsc_enc_gemm_split, csrc/encoder_gemm.hip) against a plain PyTorch fp32 evaluation of the same op and against fp64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pieces_back(p, M, K, scale=1.0):
    """(M, 2K) fp16 pieces in the chunk-interleaved layout -> (M, K) float64 of h + l"""
    v = p.view(M, K // 32, 2, 32)
    return (v[:, :, 0].double() + v[:, :, 1].double()).reshape(M, K) / scale


@pytest.mark.parametrize("M,K,N", [(1000, 768, 2304), (777, 3072, 768), (257, 64, 96), (31, 32, 8), (50, 64, 30), (20741, 64, 768), (20741, 64, 2304),
                                   (197, 608, 1024), (197, 3072, 768), (1576, 768, 2304), (2088, 4096, 1024)])
@pytest.mark.parametrize("epilogue", [0, 1, 2, 3])
def test_split_gemm_matches_fp32_linear(M, K, N, epilogue):
    """out = epilogue(A W^T + b): within a few f32 ulps of the fp64 result — at least as close as torch's own f32 GEMM — for f32
    rows and for pre-split pieces, ragged M / N (tile edges), every epilogue.  M = 20741: more tiles than CUs — N = 768 takes the
    half-width tiles of the last round, N = 2304 gives every persistent workgroup several tiles (the next tile's first chunks
    requested under the last ones); N = 30: the element-by-element epilogue.  M <= 4 096 with few big tiles: the few-rows forms — 32 x 128
    tiles with the weight chunks two iterations ahead (M <= 512; K = 608: an odd chunk count), 128 x 128 tiles above, both with split-K
    into the workspace and the finishing kernel (K = 3072 / 4096: 12 / 4 slices), or without (N = 30: no workspace route)."""
    import torch
    import torch.nn.functional as F
    from bsc_nav_amd import encoder
    torch.manual_seed(M + K + N + epilogue)
    lin = torch.nn.Linear(K, N).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    A = torch.randn(M, K, device="cuda")
    R = torch.randn(M, N, device="cuda")
    sl = encoder.SplitLinear(lin)
    ref64 = A.double() @ lin.weight.double().t() + lin.bias.double()
    ref32 = A @ lin.weight.t() + lin.bias                      # the plain PyTorch fp32 reference of the op
    if epilogue == 1:
        ref64, ref32 = F.gelu(ref64, approximate="tanh"), F.gelu(ref32, approximate="tanh")
    if epilogue == 3:       # torch.nn.GELU() exactly — the reference's DINOv2 (memory_2.py:43,738): erf by A&S 7.1.26 in the epilogue
        ref64, ref32 = F.gelu(ref64), F.gelu(ref32)
    if epilogue == 2:
        ref64, ref32 = ref64 + R.double(), ref32 + R
    res = R.clone() if epilogue == 2 else None
    out = sl(A, epilogue, resid=res)
    outp = sl(encoder.split_rows(A, 4.0), epilogue, resid=res, a_scale=4.0, a_pieces=True)
    e32 = (ref32.double() - ref64).abs().max().item()
    tol = max(2.0 * e32, 2e-6)
    assert (out.double() - ref64).abs().max().item() <= tol
    assert (outp.double() - ref64).abs().max().item() <= tol
    # and within 1e-5 of the fp32 op itself (more where the fp32 op is itself further than that from fp64: K = 3072)
    assert (out - ref32).abs().max().item() <= max(1e-5, 2.0 * e32)
    if epilogue == 2:       # in place: C aliases the residual
        r2 = R.clone()
        sl(A, 2, resid=r2, out=r2)
        assert torch.equal(r2, out)
    if epilogue in (1, 3) and N % 32 == 0:      # the hidden tensor as pieces (what fc2 reads)
        cp = sl(A, epilogue, c_pieces_scale=4.0)
        assert cp.dtype == torch.float16 and cp.shape == (M, 2 * N)
        assert (_pieces_back(cp, M, N, 4.0) - ref64).abs().max().item() <= tol + 1e-6


def test_layernorm_split_and_split_rows():
    import torch
    import torch.nn.functional as F
    from bsc_nav_amd import encoder
    torch.manual_seed(5)
    for Wd in (256, 768, 1024):
        ln = torch.nn.LayerNorm(Wd, eps=1e-6).cuda()
        ln.weight.data.uniform_(0.5, 1.5)
        ln.bias.data.uniform_(-0.5, 0.5)
        x = torch.randn(1001, Wd, device="cuda") * 3 + 0.5
        ref = F.layer_norm(x.double(), (Wd,), ln.weight.double(), ln.bias.double(), 1e-6)
        got = _pieces_back(encoder.layernorm_split(x, ln, 2.0), 1001, Wd, 2.0)
        assert (got - ref).abs().max().item() < 4e-6
        back = _pieces_back(encoder.split_rows(x, 0.5), 1001, Wd, 0.5)
        assert (back - x.double()).abs().max().item() <= 2.0 ** -21 * x.abs().max().item()


@pytest.mark.parametrize("arch", ["vit_b16", "vit_tiny_test", "vit_l14"])
def test_f32_encoder_on_split_gemms_matches_pytorch_f32(arch):
    """Tokens of the f32 ViT with every dense layer on the fp16 matrix cores against the same module on PyTorch's f32 GEMMs
    (within 2e-5 of it at a token rms of 1) and against an fp64 evaluation (no further from it than PyTorch f32 is, x1.5)."""
    import torch
    from bsc_nav_amd import encoder
    vit = encoder.RandomViT(arch, image_size=224, seed=1, dtype=torch.float32).cuda()
    for ln in [m for m in vit.modules() if isinstance(m, torch.nn.LayerNorm)]:
        ln.weight.data = 1 + 0.1 * torch.randn_like(ln.weight)
        ln.bias.data = 0.1 * torch.randn_like(ln.bias)
    rgb = torch.randint(0, 255, (5, 96, 128, 4), dtype=torch.uint8, device="cuda")
    assert vit.split_gemm
    a = vit.patch_tokens(rgb)
    vit.split_gemm = False
    b = vit.patch_tokens(rgb)
    v64 = encoder.RandomViT(arch, image_size=224, seed=1, dtype=torch.float64).cuda()
    v64.load_state_dict({k: v.double() for k, v in vit.state_dict().items()})
    v64.fused = False
    c = v64.forward_features(v64.preprocess(rgb).double())["x_norm_patchtokens"].reshape(a.shape)
    assert a.shape == b.shape and torch.isfinite(a).all()
    assert (a - b).abs().max().item() < 2e-5
    e_split, e_torch = (a.double() - c).abs().max().item(), (b.double() - c).abs().max().item()
    assert e_split <= max(1.5 * e_torch, 5e-6), (e_split, e_torch)


@pytest.mark.parametrize("B,T,H", [(3, 197, 12), (2, 261, 16), (1, 17, 2), (2, 224, 1)])
def test_attention_split_matches_fp32_softmax(B, T, H):
    """softmax(Q K^T / 8) V on fp16 pieces against PyTorch's f32 attention and an fp64 evaluation: within 1e-5 of the former,
    no further from fp64 than it (x2), incl. ragged key tiles (T not a multiple of 16) and both LDS shapes (T <= 224, <= 288)."""
    import torch
    import torch.nn.functional as F
    from bsc_nav_amd import encoder
    torch.manual_seed(B * 1000 + T + H)
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda") * 0.7
    qp = encoder.split_rows(qkv, 1.0)
    out = encoder.attention_split(qp, B, T, H, out_scale=16.0)
    got = _pieces_back(out, B * T, H * 64, 16.0)
    v = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref64 = F.scaled_dot_product_attention(v[0].double(), v[1].double(), v[2].double()).transpose(1, 2).reshape(B * T, H * 64)
    ref32 = F.scaled_dot_product_attention(v[0], v[1], v[2]).transpose(1, 2).reshape(B * T, H * 64)
    e32 = (ref32.double() - ref64).abs().max().item()
    assert (got - ref64).abs().max().item() <= max(2.0 * e32, 2e-6)
    assert (got - ref32.double()).abs().max().item() <= 1e-5


@pytest.mark.parametrize("Wd,N,epilogue", [(768, 2304, 0), (768, 3072, 1), (1024, 3072, 0), (256, 512, 1)])
@pytest.mark.parametrize("M", [1000, 20741])
def test_layernorm_folded_into_the_gemm_operand_load(Wd, N, epilogue, M):
    """LN(x) W^T + b with the LayerNorm inside the GEMM's operand load (a_mode 2: gamma in the weight columns, W beta in the bias,
    (x - mean) rstd per row from the statistics records the embedding kernel / a residual epilogue wrote) against PyTorch fp32
    layer_norm -> linear and fp64; rows with a mean far from zero and mixed scales; the row means land in ln_mu."""
    import torch
    import torch.nn.functional as F
    from bsc_nav_amd import encoder
    torch.manual_seed(Wd + N + M)
    lin = torch.nn.Linear(Wd, N).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    ln = torch.nn.LayerNorm(Wd, eps=1e-6).cuda()
    ln.weight.data.uniform_(0.5, 1.5)
    ln.bias.data.uniform_(-0.5, 0.5)
    x = torch.randn(M, Wd, device="cuda") * (0.2 + 5 * torch.rand(M, 1, device="cuda")) + 3 * torch.randn(M, 1, device="cuda")
    # the records as a residual epilogue leaves them: shift = a previous mean (here: the true mean perturbed), sums per 128 columns
    s = x.mean(1) + 0.3 * torch.randn(M, device="cuda") * x.std(1)
    stats = torch.zeros(M, encoder.LN_REC, device="cuda")
    stats[:, 0] = s
    d = (x - s[:, None]).view(M, Wd // 128, 128)
    stats[:, 2:2 + 2 * (Wd // 128):2] = d.sum(2)
    stats[:, 3:3 + 2 * (Wd // 128):2] = (d * d).sum(2)
    mu = torch.full((M,), 7.0, device="cuda")
    sl = encoder.SplitLinear(lin, ln)
    cp = sl(x, epilogue, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=2.0)
    got = _pieces_back(cp, M, N, 2.0)
    ref64 = F.layer_norm(x.double(), (Wd,), ln.weight.double(), ln.bias.double(), 1e-6) @ lin.weight.double().t() + lin.bias.double()
    ref32 = F.layer_norm(x, (Wd,), ln.weight, ln.bias, 1e-6) @ lin.weight.t() + lin.bias
    if epilogue == 1:
        ref64, ref32 = F.gelu(ref64, approximate="tanh"), F.gelu(ref32, approximate="tanh")
    e32 = (ref32.double() - ref64).abs().max().item()
    assert (got - ref64).abs().max().item() <= max(2.0 * e32, 3e-6), ((got - ref64).abs().max().item(), e32)
    assert (mu.double() - x.double().mean(1)).abs().max().item() < 2e-6 * (1 + x.abs().max().item())


@pytest.mark.parametrize("Wd,K", [(768, 768), (768, 3072), (1024, 1024)])
@pytest.mark.parametrize("M", [999, 20741])
def test_residual_epilogue_leaves_the_rows_statistics(Wd, K, M):
    """u += a W^T + b through the residual epilogue with ln_stats: u as without it (bit for bit), and the records give the mean and
    variance of the NEW rows to f32 accuracy (shifted sums about the previous mean, one slot per 128 columns; whole and half-width
    tiles, ragged M)."""
    import torch
    from bsc_nav_amd import encoder
    torch.manual_seed(Wd + K + M)
    lin = torch.nn.Linear(K, Wd).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    a = torch.randn(M, K, device="cuda")
    u0 = torch.randn(M, Wd, device="cuda") * 4 + 2 * torch.randn(M, 1, device="cuda")
    sl = encoder.SplitLinear(lin)
    ap = encoder.split_rows(a, 4.0)
    plain = u0.clone()
    sl(ap, 2, resid=plain, out=plain, a_scale=4.0, a_pieces=True)
    u = u0.clone()
    stats = torch.full((M, encoder.LN_REC), float("nan"), device="cuda")
    stats[:, 2 + 2 * (Wd // 128):] = 0                                   # unused slots are zero (the embedding kernel writes them)
    stats[:, 1] = 0
    mu = u0.mean(1).contiguous()
    sl(ap, 2, resid=u, out=u, a_scale=4.0, a_pieces=True, ln_stats=stats, ln_mu=mu)
    if M > 4096:
        assert torch.equal(u, plain)
    else:       # few rows: the plain call takes the 32-row tiles with split-K (another summation order), the statistics form the big tile
        assert (u - plain).abs().max().item() < 2e-5
    assert torch.equal(stats[:, 0], mu)
    sa = stats[:, 2::2].double().sum(1)
    sb = stats[:, 3::2].double().sum(1)
    mean = stats[:, 0].double() + sa / Wd
    var = sb / Wd - (sa / Wd) ** 2
    u64 = u.double()
    assert (mean - u64.mean(1)).abs().max().item() < 1e-5
    assert ((var - u64.var(1, unbiased=False)).abs() / u64.var(1, unbiased=False)).max().item() < 2e-6
    # the same epilogue behind f32 operand rows (split in registers)
    plain32, u32 = u0.clone(), u0.clone()
    sl(a, 2, resid=plain32, out=plain32)
    st2 = torch.zeros((M, encoder.LN_REC), device="cuda")
    sl(a, 2, resid=u32, out=u32, ln_stats=st2, ln_mu=mu)
    assert torch.equal(u32, plain32) if M > 4096 else (u32 - plain32).abs().max().item() < 2e-5
    sa2, sb2 = st2[:, 2::2].double().sum(1), st2[:, 3::2].double().sum(1)
    assert ((st2[:, 0].double() + sa2 / Wd) - u32.double().mean(1)).abs().max().item() < 1e-5


def test_embed_and_final_layernorm_f32():
    """token assembly (cls + pos, registers, patches + pos) with the first LayerNorm as pieces and / or as statistics records, and
    the final LayerNorm over the patch rows, against the PyTorch ops"""
    import torch
    import torch.nn.functional as F
    from bsc_nav_amd import encoder
    for arch in ("vit_b16", "vit_l14"):
        vit = encoder.RandomViT(arch, image_size=224, seed=3, dtype=torch.float32).cuda()
        torch.manual_seed(11)
        vit.cls.data.normal_(); vit.pos.data.normal_()
        if vit.reg is not None:
            vit.reg.data.normal_()
        B, g, Wd, R = 3, vit.grid, vit.width, vit.registers
        T = 1 + R + g * g
        ln = vit.blocks[0].ln1
        ln.weight.data.uniform_(0.5, 1.5); ln.bias.data.uniform_(-0.5, 0.5)
        x = torch.randn(B, g * g, Wd, device="cuda") * 2 + 0.7
        t = torch.cat([vit.cls.expand(B, -1, -1), x], dim=1) + vit.pos
        if vit.reg is not None:
            t = torch.cat([t[:, :1], vit.reg.expand(B, -1, -1), t[:, 1:]], dim=1)
        u, pieces, st = encoder.embed_tokens_f32(vit, x.view(B * g * g, Wd), B, ln=ln, stats=True)
        assert torch.equal(u.view(B, T, Wd), t)
        ref = F.layer_norm(t.double(), (Wd,), ln.weight.double(), ln.bias.double(), ln.eps).view(B * T, Wd)
        assert (_pieces_back(pieces, B * T, Wd) - ref).abs().max().item() < 4e-6
        stats, mu = st
        assert (mu.double() - t.double().mean(2).view(-1)).abs().max().item() < 1e-6
        assert torch.equal(stats[:, 0], mu) and (stats[:, 2] == 0).all() and (stats[:, 4:] == 0).all()
        var = stats[:, 3].double() / Wd
        assert ((var - t.double().var(2, unbiased=False).view(-1)).abs() / var).max().item() < 2e-6
        u2, p2, st2 = encoder.embed_tokens_f32(vit, x.view(B * g * g, Wd), B, ln=None, stats=True)
        assert p2 is None and torch.equal(u2, u) and torch.equal(st2[0], stats)
        out = encoder.final_layernorm_f32(u, vit.norm, B, T, 1 + R)
        reff = F.layer_norm(t[:, 1 + R:], (Wd,), vit.norm.weight, vit.norm.bias, vit.norm.eps)
        assert out.shape == reff.shape and (out - reff).abs().max().item() < 2e-6


@pytest.mark.parametrize("arch", ["vit_b16", "vit_l14"])
def test_f32_encoder_fused_layernorm_equals_the_unfused_form(arch, monkeypatch):
    """the forward with LayerNorm folded into the GEMMs (default) against the one that runs LayerNorm as passes of its own"""
    import torch
    from bsc_nav_amd import encoder
    vit = encoder.RandomViT(arch, image_size=224, seed=2, dtype=torch.float32).cuda()
    for ln in [m for m in vit.modules() if isinstance(m, torch.nn.LayerNorm)]:
        ln.weight.data = 1 + 0.1 * torch.randn_like(ln.weight)
        ln.bias.data = 0.1 * torch.randn_like(ln.bias)
    rgb = torch.randint(0, 255, (3, 96, 128, 4), dtype=torch.uint8, device="cuda")
    a = vit.patch_tokens(rgb)
    monkeypatch.setenv("BSC_ENC_LN_FUSED", "0")
    b = vit.patch_tokens(rgb)
    assert a.shape == b.shape and torch.isfinite(a).all()
    assert (a - b).abs().max().item() < 1e-5


@pytest.mark.parametrize("arch", ["vit_l14", "vit_b16"])
def test_patch_matrix_as_padded_pieces(arch):
    """the fused preprocessing as operand pieces: patch 14 has 588 columns, padded with zeros to 608 (the GEMM contracts 32 at a
    time; the weight gets the same zero columns) — values equal the f32 output, pad columns are zero, and the patch embedding
    through the in-tree GEMM equals the f32 nn.Linear"""
    import torch
    from bsc_nav_amd import encoder
    vit = encoder.RandomViT(arch, image_size=224, seed=4, dtype=torch.float32).cuda()
    rgb = torch.randint(0, 255, (3, 480, 640, 4), dtype=torch.uint8, device="cuda")
    f = vit.preprocess_patches(rgb, mode=1)
    pz = vit.preprocess_patches(rgb, mode=2)
    K = 3 * vit.patch ** 2
    Kp = (K + 31) // 32 * 32
    M = 3 * vit.grid ** 2
    assert pz.shape == (3, vit.grid ** 2, 2 * Kp)
    back = _pieces_back(pz.view(M, 2 * Kp), M, Kp)
    assert (back[:, :K] - f.view(M, K).double()).abs().max().item() <= 2.0 ** -21 * f.abs().max().item()
    assert (back[:, K:] == 0).all()
    x = vit._split(vit.patch_embed, k_pad=Kp)(pz.view(M, 2 * Kp), a_pieces=True)
    ref = f.view(M, K).double() @ vit.patch_embed.weight.double().t() + vit.patch_embed.bias.double()
    assert (x.double() - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize("M,K,N,epilogue", [(197, 768, 2304, 0), (261, 4096, 1024, 2), (1576, 768, 3072, 1)])
def test_few_rows_gemm_without_split_k(M, K, N, epilogue, monkeypatch):
    """the few-rows tiles on their own (BSC_GEMM_NO_SPLITK: what a caller without a workspace gets) against the split-K route"""
    import torch
    from bsc_nav_amd import encoder
    torch.manual_seed(M + K)
    lin = torch.nn.Linear(K, N).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    A = torch.randn(M, K, device="cuda")
    R = torch.randn(M, N, device="cuda")
    sl = encoder.SplitLinear(lin)
    a = sl(A, epilogue, resid=R.clone() if epilogue == 2 else None)
    monkeypatch.setenv("BSC_GEMM_NO_SPLITK", "1")
    b = sl(A, epilogue, resid=R.clone() if epilogue == 2 else None)
    ref = A.double() @ lin.weight.double().t() + lin.bias.double()
    if epilogue == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if epilogue == 2:
        ref = ref + R.double()
    assert (a.double() - ref).abs().max().item() < 1e-5 and (b.double() - ref).abs().max().item() < 1e-5
