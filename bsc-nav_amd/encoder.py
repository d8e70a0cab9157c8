"""Patch-feature providers: the per-frame ViT forward of the memory path (memory_2.py:732-742).

The reference takes DINOv2 `x_norm_patchtokens` from torch.hub (third-party, not vendored, weights not
available offline), so encoder numerics are parity-unpinned.  What IS pinned is the
interface: rgb u8 (H,W,3) -> /255 -> resize (query_h, query_w) -> ImageNet normalise -> ViT ->
final-LayerNorm'd patch tokens, reshaped (g, g, D) and indexed [py, px].

`RandomViT` is a plain ViT (B/16: 12x768x12, L/14: 24x1024x16 with 4 register tokens) with
trunc_normal(0.02) weights.  In bf16 its dense GEMMs run on MFMA through PyTorch-ROCm (hipBLASLt); in f32 (the
reference's precision) they run in-tree on the fp16 matrix cores with split operands (`SplitLinear`,
csrc/encoder_gemm.hip: bias / GELU / residual in the epilogue).  Outputs are returned in fp32 for the voxel kernels.
"""
import ctypes as C
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

VIT_SHAPES = {
    "vit_b16": dict(patch=16, width=768, depth=12, heads=12, mlp=3072, registers=0),
    "vit_l14": dict(patch=14, width=1024, depth=24, heads=16, mlp=4096, registers=4),
    "vit_tiny_test": dict(patch=16, width=64, depth=2, heads=4, mlp=128, registers=0),
    # DINOv2 hub shapes (the reference loads args.dino_size = dinov2_vitl14_reg, args.py:50); ViT-g uses SwiGLU: not covered
    "vit_s14": dict(patch=14, width=384, depth=12, heads=6, mlp=1536, registers=0),
    "vit_b14": dict(patch=14, width=768, depth=12, heads=12, mlp=3072, registers=0),
    "vit_s14_reg": dict(patch=14, width=384, depth=12, heads=6, mlp=1536, registers=4),
    "vit_b14_reg": dict(patch=14, width=768, depth=12, heads=12, mlp=3072, registers=4),
    "vit_l14_noreg": dict(patch=14, width=1024, depth=24, heads=16, mlp=4096, registers=0),
}

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


_ATT_WORK = {}            # (device, stream) -> ticket / finished counters of the persistent attention kernels
_ATT_WORK_OVERRIDE = []   # a GraphedEncoder's own pair while it warms up and captures


def attention_work(device):
    """Ticket / finished counters of the attention kernels: launches on one stream are ordered and may share a pair; a
    captured graph owns its pair (GraphedEncoder allocates it before the capture), so two graphs replayed on different
    streams never meet in one."""
    if _ATT_WORK_OVERRIDE:
        return _ATT_WORK_OVERRIDE[-1]
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    if key not in _ATT_WORK:
        _ATT_WORK[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return _ATT_WORK[key]


def attention_split(qkv_pieces, B, T, heads, out_scale=1.0):
    """softmax(Q K^T / 8) V at f32 accuracy on the fp16 matrix cores (bsc_enc_attention_split): qkv_pieces = the pieces the qkv
    GEMM wrote, (B T, 2 * 3 * heads * 64) fp16 -> pieces of out_scale * attention, (B T, 2 * heads * 64) fp16."""
    from . import _lib
    out = torch.empty((B * T, 2 * heads * 64), dtype=torch.float16, device=qkv_pieces.device)
    _lib.check(_lib.load().bsc_enc_attention_split(
        C.c_void_p(qkv_pieces.data_ptr()), B, T, heads, 64, C.c_void_p(out.data_ptr()), float(out_scale),
        C.c_void_p(attention_work(qkv_pieces.device).data_ptr()),
        C.c_void_p(torch.cuda.current_stream(qkv_pieces.device).cuda_stream)))
    return out


class SplitLinear:
    """nn.Linear at f32 accuracy on the fp16 matrix cores (bsc_enc_gemm_split): the weight's two fp16 pieces are made once
    (scaled by the power of two that puts its largest element in (4, 8]), the activation rows are split in registers.
    epilogue: 0 bias, 1 bias + GELU(tanh), 2 bias + residual (in place when out is resid), 3 bias + GELU(erf: torch.nn.GELU(),
    what the reference's DINOv2 applies)."""

    BIAS, GELU, RESID, GELU_ERF = 0, 1, 2, 3
    A_F32, A_PIECES, A_LN = 0, 1, 2

    def __init__(self, lin, ln=None, k_pad=None):
        """ln: a LayerNorm applied to the input rows, folded into this layer — its gamma into the weight columns, its beta into
        the bias (LN(x) W^T + b = ((x - mean) rstd) (W gamma)^T + (W beta + b)); the GEMM then normalises the rows of the residual
        stream while it splits them (a_mode A_LN: mean / rstd from the row statistics the producing GEMM's epilogue left)."""
        from . import _lib
        w = lin.weight.detach()
        if k_pad is not None and k_pad != w.shape[1]:       # zero columns: the operand rows are padded the same way (patch matrix, K = 588)
            w = F.pad(w, (0, k_pad - w.shape[1]))
        assert w.is_cuda and w.dtype == torch.float32 and w.shape[1] % 32 == 0
        self.bias = None if lin.bias is None else lin.bias.detach().contiguous()
        self.ln_eps = 0.0
        if ln is not None:
            g, b = ln.weight.detach().float(), ln.bias.detach().float()
            # W beta in f64: a 768-term dot per output, once per matrix
            wb = (w.double() @ b.double()).float()
            self.bias = wb if self.bias is None else (self.bias.double() + wb.double()).float()
            w = w * g[None, :]
            self.ln_eps = float(ln.eps)
        self.N, self.K = w.shape
        amax = float(w.abs().max())
        self.scale = 2.0 ** (3 - int(torch.ceil(torch.log2(torch.tensor(max(amax, 1e-30))))))
        n_pad = (self.N + 255) // 256 * 256
        self.pieces = torch.empty((2, n_pad, self.K), dtype=torch.float16, device=w.device)
        lib = _lib.load()
        _lib.check(lib.bsc_enc_split_weights(C.c_void_p(w.contiguous().data_ptr()), self.N, self.K, self.scale,
                                             C.c_void_p(self.pieces.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)))

    def __call__(self, x2d, epilogue=0, resid=None, out=None, a_scale=1.0, a_pieces=False, c_pieces_scale=0.0, ln_stats=None,
                 ln_mu=None, a_ln=False):
        """x2d: (M,K) f32 rows, or (a_pieces) their (M,2K) fp16 pieces as `split_rows` / `layernorm_split` / a GELU epilogue
        with c_pieces_scale made them, already scaled by a_scale.  -> (M,N) f32, or (c_pieces_scale) (M,2N) fp16 pieces.
        a_ln: x2d = f32 rows of the residual stream, LayerNorm'd in the operand load from ln_stats (M, 20) (this object was built
        with ln=...; writes the row means to ln_mu).  ln_stats with the residual epilogue: the finished rows leave their statistics
        there (shifted by ln_mu) for the next a_ln GEMM."""
        from . import _lib
        M = x2d.shape[0]
        if a_pieces:
            assert x2d.dtype == torch.float16 and x2d.is_contiguous() and x2d.shape[1] == 2 * self.K
        else:
            assert x2d.dtype == torch.float32 and x2d.is_contiguous() and x2d.shape[1] == self.K
        if a_ln or ln_stats is not None:
            assert ln_stats.dtype == torch.float32 and ln_stats.shape == (M, LN_REC) and ln_mu.shape == (M,) and ln_stats.is_contiguous()
            assert not a_ln or self.ln_eps > 0.0, "a_ln needs a SplitLinear built with ln="
        if out is None:
            out = (torch.empty((M, 2 * self.N), dtype=torch.float16, device=x2d.device) if c_pieces_scale else
                   torch.empty((M, self.N), dtype=torch.float32, device=x2d.device))
        # few rows (a frame or a handful per call): workspace for the split-K partial results — from the caching allocator, so it is
        # ordered on the stream and lives in a capturing graph's pool
        ws = torch.empty(SPLITK_WS_BYTES, dtype=torch.uint8, device=x2d.device) if (M <= 8192 and not a_ln and ln_stats is None) else None
        _lib.check(_lib.load().bsc_enc_gemm_split_ws(
            C.c_void_p(x2d.data_ptr()), M, self.K, C.c_void_p(self.pieces.data_ptr()), self.N,
            None if self.bias is None else C.c_void_p(self.bias.data_ptr()),
            None if resid is None else C.c_void_p(resid.data_ptr()), C.c_void_p(out.data_ptr()), float(a_scale),
            1.0 / (float(a_scale) * self.scale), int(epilogue), self.A_LN if a_ln else self.A_PIECES if a_pieces else self.A_F32,
            float(c_pieces_scale), None if ln_stats is None else C.c_void_p(ln_stats.data_ptr()),
            None if ln_mu is None else C.c_void_p(ln_mu.data_ptr()), self.ln_eps,
            None if ws is None else C.c_void_p(ws.data_ptr()), 0 if ws is None else SPLITK_WS_BYTES,
            C.c_void_p(torch.cuda.current_stream(x2d.device).cuda_stream)))
        return out


SPLITK_WS_BYTES = 34 << 20      # split-K partial results of a few-rows GEMM: slices x tiles <= 2 x 256 CUs, 128 x 128 f32 each
LN_REC = 20       # floats per LayerNorm statistics record of a residual-stream row (csrc/encoder_gemm.hip GS_LN_REC)


def embed_tokens_f32(vit, patches2d, B, ln=None, stats=False):
    """Token assembly of the f32 forward in one pass (bsc_enc_embed_layernorm_f32): cls + pos[0], register tokens, patch + pos ->
    (u (B T, W) f32, pieces of ln(u) or None, (ln_stats, ln_mu) or None)."""
    from . import _lib
    Wd = vit.width
    T = 1 + vit.registers + vit.grid * vit.grid
    dev = patches2d.device
    u = torch.empty((B * T, Wd), dtype=torch.float32, device=dev)
    pieces = torch.empty((B * T, 2 * Wd), dtype=torch.float16, device=dev) if ln is not None else None
    st = (torch.empty((B * T, LN_REC), dtype=torch.float32, device=dev), torch.empty(B * T, dtype=torch.float32, device=dev)) if stats else None
    _lib.check(_lib.load().bsc_enc_embed_layernorm_f32(
        C.c_void_p(patches2d.data_ptr()), C.c_void_p(vit.cls.data_ptr()), None if vit.reg is None else C.c_void_p(vit.reg.data_ptr()),
        C.c_void_p(vit.pos.data_ptr()), None if ln is None else C.c_void_p(ln.weight.data_ptr()),
        None if ln is None else C.c_void_p(ln.bias.data_ptr()), B, T, vit.registers, Wd, float(ln.eps) if ln is not None else 0.0,
        C.c_void_p(u.data_ptr()), None if pieces is None else C.c_void_p(pieces.data_ptr()),
        None if st is None else C.c_void_p(st[0].data_ptr()), None if st is None else C.c_void_p(st[1].data_ptr()),
        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return u, pieces, st


def final_layernorm_f32(u2d, ln, B, T, skip):
    """the last LayerNorm over the patch rows only -> (B, T - skip, W) f32 (bsc_enc_final_layernorm_f32)"""
    from . import _lib
    Wd = u2d.shape[1]
    out = torch.empty((B, T - skip, Wd), dtype=torch.float32, device=u2d.device)
    _lib.check(_lib.load().bsc_enc_final_layernorm_f32(
        C.c_void_p(u2d.data_ptr()), C.c_void_p(ln.weight.data_ptr()), C.c_void_p(ln.bias.data_ptr()), B, T, skip, Wd, float(ln.eps),
        C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(u2d.device).cuda_stream)))
    return out


def layernorm_split(x2d, ln, a_scale=1.0):
    """LayerNorm of f32 rows written as the fp16 pieces the split GEMM reads (bsc_enc_layernorm_split)."""
    from . import _lib
    M, Wd = x2d.shape
    out = torch.empty((M, 2 * Wd), dtype=torch.float16, device=x2d.device)
    _lib.check(_lib.load().bsc_enc_layernorm_split(
        C.c_void_p(x2d.data_ptr()), C.c_void_p(ln.weight.data_ptr()), C.c_void_p(ln.bias.data_ptr()), M, Wd, float(ln.eps),
        float(a_scale), C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(x2d.device).cuda_stream)))
    return out


def split_rows(x2d, a_scale=1.0):
    """f32 rows -> fp16 pieces (bsc_enc_split_rows)."""
    from . import _lib
    M, K = x2d.shape
    out = torch.empty((M, 2 * K), dtype=torch.float16, device=x2d.device)
    _lib.check(_lib.load().bsc_enc_split_rows(C.c_void_p(x2d.data_ptr()), M, K, float(a_scale), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream(x2d.device).cuda_stream)))
    return out


class _Block(nn.Module):
    def __init__(self, width, heads, mlp):
        super().__init__()
        self.heads = heads
        self.ln1 = nn.LayerNorm(width, eps=1e-6)
        self.qkv = nn.Linear(width, 3 * width)
        self.proj = nn.Linear(width, width)
        self.ln2 = nn.LayerNorm(width, eps=1e-6)
        self.fc1 = nn.Linear(width, mlp)
        self.fc2 = nn.Linear(mlp, width)

    def attn_heads(self, y):
        """fused attention (bsc_enc_attention) of LayerNorm'd rows y: (B, T, width) before the output projection"""
        from . import _lib
        B, T, Wd = y.shape
        hd = Wd // self.heads
        qkv = self.qkv(y)
        a = torch.empty((B, T, Wd), dtype=torch.bfloat16, device=y.device)
        # ticket / finished counters: one pair per (device, stream) — launches on one stream are ordered and may share
        # them, two forwards of this module on different streams (a graph replay beside an eager query embedding) may not
        stream = torch.cuda.current_stream(y.device)
        _lib.check(_lib.load().bsc_enc_attention_dyn(C.c_void_p(qkv.data_ptr()), B, T, self.heads, hd,
                                                     C.c_void_p(a.data_ptr()), C.c_void_p(attention_work(y.device).data_ptr()),
                                                     C.c_void_p(stream.cuda_stream)))
        return a

    def can_fuse_attention(self, y):
        return y.shape[-1] // self.heads == 64 and y.shape[1] <= 288 and y.dtype == torch.bfloat16 and y.is_cuda

    def attn(self, y, fused=False):
        B, T, Wd = y.shape
        hd = Wd // self.heads
        if fused and self.can_fuse_attention(y):
            # (B,T,3,heads,64) straight out of the qkv GEMM -> bsc_enc_attention (K, V of a head resident in LDS)
            return self.proj(self.attn_heads(y))
        qkv = self.qkv(y).reshape(B, T, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(a.transpose(1, 2).reshape(B, T, Wd))

    gelu = "tanh"           # "erf": torch.nn.GELU() exactly (RandomViT sets it per model)

    def mlp(self, y, fuse_gelu):
        if fuse_gelu and self.gelu == "tanh":       # bias + GELU(tanh) in the GEMM epilogue (hipBLASLt)
            B, T, C = y.shape
            h = torch._addmm_activation(self.fc1.bias, y.reshape(B * T, C), self.fc1.weight.t(), use_gelu=True)
            return self.fc2(h).reshape(B, T, C)
        return self.fc2(F.gelu(self.fc1(y), approximate="tanh" if self.gelu == "tanh" else "none"))


class RandomViT(nn.Module):
    def __init__(self, arch="vit_b16", image_size=224, out_dim=None, seed=0, dtype=torch.bfloat16, fused=True,
                 init_weights=True, gelu=None):
        """gelu: the MLP activation — "erf" = torch.nn.GELU() exactly, what the reference's DINOv2 applies (memory_2.py:43,738), or
        "tanh" = its tanh approximation (differs by up to 5e-4 per activation, ~1e-2 on a token).  Default: "erf" for the f32 model
        (the reference's precision: in-tree erf epilogue of the split GEMM), "tanh" for the bf16 fast mode (the library GEMM's own
        GELU epilogue; the difference is below bf16 resolution)."""
        super().__init__()
        self.gelu = gelu or ("tanh" if dtype in (torch.bfloat16, torch.float16) else "erf")
        assert self.gelu in ("erf", "tanh")
        self.fused = fused
        self.fused_attention = fused and os.environ.get("BSC_ENC_ATTENTION", "1") == "1"   # 0: library SDPA
        # 0: residual adds in the LayerNorm kernel (the lagged form rides on the library GEMM's tanh-GELU epilogue)
        self.lagged = os.environ.get("BSC_ENC_LAGGED", "1") == "1" and self.gelu == "tanh"
        # f32 weights: dense layers through the in-tree split-operand MFMA GEMM (0: PyTorch-ROCm f32 GEMMs)
        self.split_gemm = dtype == torch.float32 and os.environ.get("BSC_ENC_SPLIT_GEMM", "1") == "1"
        s = VIT_SHAPES[arch]
        self.arch, self.image_size, self.patch = arch, image_size, s["patch"]
        self.grid = image_size // s["patch"]
        self.width = s["width"]
        self.registers = s["registers"]
        self.compute_dtype = dtype
        g = torch.Generator().manual_seed(seed)
        self.patch_embed = nn.Linear(3 * s["patch"] * s["patch"], s["width"])   # conv(p, stride p) as one GEMM
        self.cls = nn.Parameter(torch.zeros(1, 1, s["width"]))
        self.reg = nn.Parameter(torch.zeros(1, s["registers"], s["width"])) if s["registers"] else None
        self.pos = nn.Parameter(torch.zeros(1, 1 + self.grid * self.grid, s["width"]))
        self.blocks = nn.ModuleList([_Block(s["width"], s["heads"], s["mlp"]) for _ in range(s["depth"])])
        for blk in self.blocks:
            blk.gelu = self.gelu
        self.norm = nn.LayerNorm(s["width"], eps=1e-6)
        self.head = nn.Linear(s["width"], out_dim, bias=False) if out_dim and out_dim != s["width"] else None
        self.out_dim = out_dim or s["width"]
        # every random value comes from the seeded generator (matrices trunc_normal(0.02), Linear biases U(-0.02, 0.02)), so
        # two instances with the same seed hold the same weights whatever their dtype and the state of the global RNG
        for name, p in self.named_parameters() if init_weights else ():
            if p.dim() > 1:
                nn.init.trunc_normal_(p, std=0.02, generator=g)
            elif name.endswith(".bias") and isinstance(self.get_submodule(name.rsplit(".", 1)[0]), nn.Linear):
                with torch.no_grad():
                    p.uniform_(-0.02, 0.02, generator=g)
        self.register_buffer("mean", torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(IMAGENET_STD).view(1, 3, 1, 1))
        self.eval()
        # weights live in the compute dtype: no per-call casts (autocast would re-cast 50+ weights per forward)
        for name, prm in self.named_parameters():
            prm.data = prm.data.to(dtype)
            prm.requires_grad_(False)

    # ---- real weights -------------------------------------------------------------------------------------------------
    @classmethod
    def from_dinov2_state_dict(cls, sd, image_size=224, dtype=torch.bfloat16, fused=True, **kw):
        """A RandomViT of the right shape filled from a DINOv2 `state_dict()` (torch.hub facebookresearch/dinov2, the model
        the reference passes as `preload_dino`; memory_2.py:43, args.py:50)."""
        width, p = sd["patch_embed.proj.weight"].shape[0], sd["patch_embed.proj.weight"].shape[-1]
        depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks.") and k.split(".")[1].isdigit())
        regs = sd["register_tokens"].shape[1] if "register_tokens" in sd else 0
        mlp = sd["blocks.0.mlp.fc1.weight"].shape[0] if "blocks.0.mlp.fc1.weight" in sd else None
        if mlp is None:
            raise ValueError("SwiGLU feed-forward (DINOv2 ViT-g) is not covered")
        arch = next((a for a, s in VIT_SHAPES.items() if (s["patch"], s["width"], s["depth"], s["mlp"], s["registers"]) ==
                     (p, width, depth, mlp, regs)), None)
        if arch is None:
            raise ValueError(f"no ViT shape for patch {p}, width {width}, depth {depth}, mlp {mlp}, registers {regs}")
        vit = cls(arch, image_size=image_size, dtype=dtype, fused=fused, init_weights=False)
        vit.load_dinov2_state_dict(sd, **kw)
        return vit

    @torch.no_grad()
    def load_dinov2_state_dict(self, sd, interpolate_antialias=None, interpolate_offset=None):
        """Maps DINOv2's parameters onto this module: the patch convolution as the unfolded-patch GEMM, the position
        embedding resampled to this grid the way `interpolate_pos_encoding` does (bicubic; antialias and no offset for the
        register models, offset 0.1 otherwise — the hub defaults), LayerScale folded into the projection and fc2 weights
        and biases (x + g * (a W^T + b) = x + a (g W)^T + g b).  The MLP activation is `self.gelu`: the exact erf form for the f32
        model (DINOv2's nn.GELU), its tanh approximation in the bf16 mode (|tanh form - erf form| < 5e-4, below bf16 resolution)."""
        dev, dt = self.cls.device, self.compute_dtype
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32)
        put = lambda prm, t: prm.data.copy_(t.to(dt))
        D, g = self.width, self.grid
        put(self.patch_embed.weight, f(sd["patch_embed.proj.weight"]).reshape(D, -1))
        put(self.patch_embed.bias, f(sd["patch_embed.proj.bias"]))
        put(self.cls, f(sd["cls_token"]))
        if self.reg is not None:
            put(self.reg, f(sd["register_tokens"]))
        pos = f(sd["pos_embed"])
        n = pos.shape[1] - 1
        m = int(round(n ** 0.5))
        if m * m != n:
            raise ValueError("pos_embed is not a square grid")
        if m != g:
            aa = (self.registers > 0) if interpolate_antialias is None else interpolate_antialias
            off = (0.0 if self.registers > 0 else 0.1) if interpolate_offset is None else interpolate_offset
            grid = pos[:, 1:].reshape(1, m, m, D).permute(0, 3, 1, 2)
            if off:
                sc = float(g + off) / m
                grid = F.interpolate(grid, scale_factor=(sc, sc), mode="bicubic", antialias=aa)
            else:
                grid = F.interpolate(grid, size=(g, g), mode="bicubic", antialias=aa)
            assert grid.shape[-2:] == (g, g)
            pos = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, g * g, D)], dim=1)
        put(self.pos, pos)
        for i, blk in enumerate(self.blocks):
            k = f"blocks.{i}."
            g1 = f(sd[k + "ls1.gamma"]) if k + "ls1.gamma" in sd else torch.ones(D, device=dev)
            g2 = f(sd[k + "ls2.gamma"]) if k + "ls2.gamma" in sd else torch.ones(D, device=dev)
            put(blk.ln1.weight, f(sd[k + "norm1.weight"])); put(blk.ln1.bias, f(sd[k + "norm1.bias"]))
            put(blk.ln2.weight, f(sd[k + "norm2.weight"])); put(blk.ln2.bias, f(sd[k + "norm2.bias"]))
            put(blk.qkv.weight, f(sd[k + "attn.qkv.weight"])); put(blk.qkv.bias, f(sd[k + "attn.qkv.bias"]))
            put(blk.proj.weight, g1[:, None] * f(sd[k + "attn.proj.weight"])); put(blk.proj.bias, g1 * f(sd[k + "attn.proj.bias"]))
            put(blk.fc1.weight, f(sd[k + "mlp.fc1.weight"])); put(blk.fc1.bias, f(sd[k + "mlp.fc1.bias"]))
            put(blk.fc2.weight, g2[:, None] * f(sd[k + "mlp.fc2.weight"])); put(blk.fc2.bias, g2 * f(sd[k + "mlp.fc2.bias"]))
        put(self.norm.weight, f(sd["norm.weight"])); put(self.norm.bias, f(sd["norm.bias"]))
        self._bsum = None                       # bias sums of the lagged stream follow the new weights
        return self

    @torch.no_grad()
    def preprocess(self, rgb):
        """rgb (B,H,W,>=3) u8 -> (B,3,S,S) normalised float (memory_2.py:733-736, transform_ :71-74)."""
        x = rgb[..., :3].permute(0, 3, 1, 2).float() / 255
        if x.shape[-2:] != (self.image_size, self.image_size):
            x = F.interpolate(x, size=(self.image_size, self.image_size), mode="bilinear", antialias=True,
                              align_corners=False)
        return (x - self.mean) / self.std

    @torch.no_grad()
    def forward_features(self, x):
        """(B,3,S,S) -> {'x_norm_patchtokens': (B, g*g, D)}  (the key the reference reads, memory_2.py:739)."""
        B, g, p = x.shape[0], self.grid, self.patch
        x = x.to(self.compute_dtype)
        t = x.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
        return self._forward_patches(t)

    @torch.no_grad()
    def _forward_patches(self, t, keep_dtype=False):
        """(B, g*g, 3*p*p) unfolded, normalised patches -> {'x_norm_patchtokens': (B, g*g, D)} (f32 unless keep_dtype)."""
        B = t.shape[0]
        if self.split_gemm and t.is_cuda and t.dtype == torch.float32 and self.head is None:
            return {"x_norm_patchtokens": self._forward_f32_split(t)}
        t = self.patch_embed(t)
        fuse = self.fused and t.is_cuda and t.dtype == torch.bfloat16 and self.width % 256 == 0
        T = 1 + self.registers + self.grid * self.grid
        if fuse:
            # token assembly (cls, registers, + pos) and the first LayerNorm in one pass
            from . import _lib
            lib = _lib.load()
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            ln0 = self.blocks[0].ln1
            x = torch.empty((B, T, self.width), dtype=torch.bfloat16, device=t.device)
            y = torch.empty_like(x)
            t = t.contiguous()
            _lib.check(lib.bsc_enc_embed_layernorm(
                C.c_void_p(t.data_ptr()), C.c_void_p(self.cls.data_ptr()),
                None if self.reg is None else C.c_void_p(self.reg.data_ptr()), C.c_void_p(self.pos.data_ptr()),
                C.c_void_p(ln0.weight.data_ptr()), C.c_void_p(ln0.bias.data_ptr()), C.c_void_p(x.data_ptr()),
                C.c_void_p(y.data_ptr()), B, T, self.registers, self.width, float(ln0.eps), stream))
            t = x
        else:
            t = torch.cat([self.cls.expand(B, -1, -1), t], dim=1) + self.pos
            if self.reg is not None:
                t = torch.cat([t[:, :1], self.reg.expand(B, -1, -1), t[:, 1:]], dim=1)
            t = t.contiguous()
            t, y = self._add_ln(t, None, self.blocks[0].ln1, fuse)
        if fuse and self.lagged and self.head is None and self.fused_attention and self.blocks[0].can_fuse_attention(y):
            return {"x_norm_patchtokens": self._forward_lagged(t, y, keep_dtype)}
        # pre-LN transformer with every residual add fused into the LayerNorm that follows it
        last = len(self.blocks) - 1
        for i, blk in enumerate(self.blocks):
            t, y = self._add_ln(t, blk.attn(y, fuse and self.fused_attention), blk.ln2, fuse)
            if i == last and fuse and self.head is None:
                # last residual add + final LayerNorm of the patch rows, written as the token tensor itself
                delta = blk.mlp(y, fuse).contiguous()
                skip = 1 + self.registers
                out = torch.empty((B, T - skip, self.width), dtype=torch.bfloat16 if keep_dtype else torch.float32,
                                  device=t.device)
                _lib.check(lib.bsc_enc_final_layernorm(
                    C.c_void_p(t.data_ptr()), C.c_void_p(delta.data_ptr()), C.c_void_p(self.norm.weight.data_ptr()),
                    C.c_void_p(self.norm.bias.data_ptr()), C.c_void_p(out.data_ptr()), 0 if keep_dtype else 1, B, T, skip,
                    self.width, float(self.norm.eps), stream))
                return {"x_norm_patchtokens": out}
            nxt = self.blocks[i + 1].ln1 if i < last else self.norm
            t, y = self._add_ln(t, blk.mlp(y, fuse), nxt, fuse)
        t = y[:, 1 + self.registers:]
        if self.head is not None:
            t = self.head(t)
        return {"x_norm_patchtokens": t if keep_dtype else t.float()}

    def _split(self, lin, ln=None, k_pad=None):
        """the fp16 pieces of a Linear's weight, made on first use (f32 weights on the device); ln: a LayerNorm folded in"""
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (id(lin), None if ln is None else id(ln), k_pad)
        # in-place updates (load_state_dict, copy_) keep the pointer and bump the version counter
        tag = (lin.weight.data_ptr(), lin.weight._version, None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version),
               None if ln is None else (ln.weight.data_ptr(), ln.weight._version, ln.bias.data_ptr(), ln.bias._version))
        if key not in cache or cache[key][0] != tag:
            cache[key] = (tag, SplitLinear(lin, ln, k_pad))
        return cache[key][1]

    def invalidate_split_weights(self):
        """drop the cached fp16 weight pieces (after replacing weights by a route the version counters do not see)"""
        self.__dict__.pop("_split_cache", None)

    def _forward_f32_split(self, t, t_pieces=False):
        """The f32 forward (the reference's precision) with every dense layer on the fp16 matrix cores at f32 accuracy:
        bias, GELU and the residual adds ride in the GEMM epilogues, LayerNorm and the MLP's hidden tensor leave as operand
        pieces, attention runs on pieces (bsc_enc_attention_split).  t: the unfolded patch matrix, f32 or (t_pieces) pieces."""
        B, n_patch = t.shape[0], t.shape[1]
        kin = t.shape[2] // 2 if t_pieces else t.shape[2]
        Wd, heads = self.width, self.blocks[0].heads
        hd = Wd // heads
        SL = SplitLinear
        act = SL.GELU_ERF if self.gelu == "erf" else SL.GELU
        if kin % 32 == 0:       # (piece rows come zero-padded to a multiple of 32: ViT-L/14's 588 columns as 608)
            x = self._split(self.patch_embed, k_pad=kin)(t.reshape(B * n_patch, t.shape[2]).contiguous(), a_pieces=t_pieces).view(B, n_patch, Wd)
        else:
            x = self.patch_embed(t)
        T = 1 + self.registers + n_patch
        ln_ok = Wd in (256, 512, 768, 1024)
        own_attention = hd == 64 and T <= 288 and os.environ.get("BSC_ENC_SPLIT_ATTENTION", "1") == "1"
        # LayerNorm folded into the qkv / fc1 operand loads, its statistics from the proj / fc2 epilogues: no LayerNorm pass at all
        # (BSC_ENC_LN_FUSED=0: LayerNorm -> pieces as a pass of its own, the round-4 form)
        # (a frame or a handful per call — at most 6 144 rows — takes the LayerNorm pass: the GEMM then runs on its 32-row tiles,
        #  which fill the chip where the 256-row tiles of the fused forms would leave it to 3 .. 12 workgroups: bsc_enc_gemm_split_ln)
        ln_fused = ln_ok and own_attention and os.environ.get("BSC_ENC_LN_FUSED", "1") == "1" and B * T > 6144
        if ln_ok:
            x2 = x.reshape(B * n_patch, Wd).contiguous()
            u, y0, st = embed_tokens_f32(self, x2, B, ln=None if ln_fused else self.blocks[0].ln1, stats=ln_fused)
        else:
            x = torch.cat([self.cls.expand(B, -1, -1), x], dim=1) + self.pos
            if self.reg is not None:
                x = torch.cat([x[:, :1], self.reg.expand(B, -1, -1), x[:, 1:]], dim=1)
            u, y0, st = x.contiguous().view(B * T, Wd), None, None

        def ln_in(ln):          # LayerNorm output as the next GEMM's operand: pieces straight from the LayerNorm kernel
            if ln_ok:
                return layernorm_split(u, ln), True
            return F.layer_norm(u, (Wd,), ln.weight, ln.bias, ln.eps), False

        if ln_fused:
            stats, mu = st
            for blk in self.blocks:
                qkv = self._split(blk.qkv, blk.ln1)(u, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=1.0)
                a = attention_split(qkv, B, T, heads, out_scale=16.0)
                self._split(blk.proj)(a, SL.RESID, resid=u, out=u, a_scale=16.0, a_pieces=True, ln_stats=stats, ln_mu=mu)
                h = self._split(blk.fc1, blk.ln2)(u, act, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=4.0)
                self._split(blk.fc2)(h, SL.RESID, resid=u, out=u, a_scale=4.0, a_pieces=True, ln_stats=stats, ln_mu=mu)
            return final_layernorm_f32(u, self.norm, B, T, 1 + self.registers)
        for bi, blk in enumerate(self.blocks):
            y, yp = (y0, True) if (bi == 0 and y0 is not None) else ln_in(blk.ln1)
            if own_attention:
                # q, k, v leave the GEMM as pieces, the attention kernel reads and writes pieces: no f32 copy, no transpose
                qkv = self._split(blk.qkv)(y, a_pieces=yp, c_pieces_scale=1.0)
                a = attention_split(qkv, B, T, heads, out_scale=16.0)
                self._split(blk.proj)(a, SL.RESID, resid=u, out=u, a_scale=16.0, a_pieces=True)
            else:
                qkv = self._split(blk.qkv)(y, a_pieces=yp).view(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
                a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
                a = a.transpose(1, 2).reshape(B * T, Wd)
                self._split(blk.proj)(a, SL.RESID, resid=u, out=u, a_scale=16.0)
            y, yp = ln_in(blk.ln2)
            # the hidden tensor exists only as pieces (scaled by 4): written by fc1's GELU epilogue, read by fc2
            h = self._split(blk.fc1)(y, act, a_pieces=yp, c_pieces_scale=4.0)
            self._split(blk.fc2)(h, SL.RESID, resid=u, out=u, a_scale=4.0, a_pieces=True)
        if ln_ok:
            return final_layernorm_f32(u, self.norm, B, T, 1 + self.registers)
        y = F.layer_norm(u, (Wd,), self.norm.weight, self.norm.bias, self.norm.eps)
        return y.view(B, T, Wd)[:, 1 + self.registers:].contiguous()

    def _bias_sums(self, device):
        """f32 running sums of the biases of the residual updates (proj, fc2 of every block): row k = what the stream lacks
        after k updates"""
        if getattr(self, "_bsum", None) is None or self._bsum.device != device:
            acc, rows = torch.zeros(self.width, dtype=torch.float32, device=device), []
            for blk in self.blocks:
                for lin in (blk.proj, blk.fc2):
                    acc = acc + lin.bias.float()
                    rows.append(acc)
            self._bsum = torch.stack(rows).contiguous()
        return self._bsum

    def _forward_lagged(self, u, y, keep_dtype):
        """The transformer stack on a bias-lagged residual stream: the projection and fc2 GEMMs accumulate into the stream
        (D = A W^T + C, beta = 1, no bias — the add happens in the GEMM's f32 accumulator), LayerNorm adds the running bias
        sum on the fly (bsc_enc_bias_layernorm): one read + one write per LayerNorm instead of two + two."""
        from . import _lib
        lib = _lib.load()
        B, T, Wd = u.shape
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        bsum = self._bias_sums(u.device)
        u2 = u.view(B * T, Wd)
        last = len(self.blocks) - 1
        skip = 1 + self.registers

        def bias_ln(k, ln, final):
            if final:
                out = torch.empty((B, T - skip, Wd), dtype=torch.bfloat16 if keep_dtype else torch.float32, device=u.device)
            else:
                out = torch.empty_like(u)
            _lib.check(lib.bsc_enc_bias_layernorm(
                C.c_void_p(u.data_ptr()), C.c_void_p(bsum[k].data_ptr()), C.c_void_p(ln.weight.data_ptr()),
                C.c_void_p(ln.bias.data_ptr()), C.c_void_p(out.data_ptr()), 0 if (keep_dtype or not final) else 1, B, T,
                skip if final else 0, Wd, float(ln.eps), stream))
            return out

        for i, blk in enumerate(self.blocks):
            a = blk.attn_heads(y)
            torch.addmm(u2, a.view(B * T, Wd), blk.proj.weight.t(), out=u2)
            y = bias_ln(2 * i, blk.ln2, False)
            h = torch._addmm_activation(blk.fc1.bias, y.view(B * T, Wd), blk.fc1.weight.t(), use_gelu=True)
            torch.addmm(u2, h, blk.fc2.weight.t(), out=u2)
            if i == last:
                return bias_ln(2 * i + 1, self.norm, True)
            y = bias_ln(2 * i + 1, self.blocks[i + 1].ln1, False)

    def _add_ln(self, x, delta, ln, fuse):
        """(x + delta, LayerNorm(x + delta)); one HIP kernel (bsc_enc_add_layernorm) when `fuse`."""
        if not fuse:
            if delta is not None:
                x = x + delta
            return x, ln(x)
        from . import _lib
        lib = _lib.load()
        B, T, Cw = x.shape
        y = torch.empty_like(x)
        xout = torch.empty_like(x) if delta is not None else x
        if delta is not None:
            delta = delta.contiguous()
        _lib.check(lib.bsc_enc_add_layernorm(
            C.c_void_p(x.data_ptr()), None if delta is None else C.c_void_p(delta.data_ptr()),
            C.c_void_p(ln.weight.data_ptr()), C.c_void_p(ln.bias.data_ptr()),
            None if delta is None else C.c_void_p(xout.data_ptr()), C.c_void_p(y.data_ptr()), B * T, Cw,
            float(ln.eps), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return xout, y

    def can_fuse_preprocess(self, rgb):
        return self.fused and rgb.is_cuda and self.compute_dtype == torch.bfloat16 and rgb.is_contiguous()

    def can_fuse_preprocess_f32(self, rgb):
        """the f32 encoder on split GEMMs takes its patch matrix from the same fused kernel: as fp16 pieces when the patch
        embedding can read them (3 p^2 % 32 == 0), otherwise as f32"""
        return self.fused and self.split_gemm and rgb.is_cuda and rgb.is_contiguous() and self.head is None

    @torch.no_grad()
    def preprocess_patches(self, rgb, out=None, mode=0):
        """u8 frames (B,H,W,C) -> normalised, unfolded patch matrix (B, g*g, 3*p*p) in one pass (bsc_enc_preprocess_patches:
        /255, antialiased bilinear resize, ImageNet normalise, unfold): mode 0 bf16, 1 f32, 2 fp16 pieces (B, g*g, 2*3*p*p)."""
        from . import _lib
        B, H, W, Cc = rgb.shape
        g, p = self.grid, self.patch
        kin = 3 * p * p
        if mode == 0:
            shape, dt = (B, g * g, kin), torch.bfloat16
        elif mode == 1:
            shape, dt = (B, g * g, kin), torch.float32
        else:
            shape, dt = (B, g * g, 2 * ((kin + 31) // 32 * 32)), torch.float16
        patches = out if out is not None else torch.empty(shape, dtype=dt, device=rgb.device)
        mean = (C.c_float * 3)(*IMAGENET_MEAN)
        std = (C.c_float * 3)(*IMAGENET_STD)
        _lib.check(_lib.load().bsc_enc_preprocess_patches_typed(
            C.c_void_p(rgb.data_ptr()), B, H, W, Cc, self.image_size, p, C.c_void_p(patches.data_ptr()), mode, mean, std,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return patches

    @torch.no_grad()
    def patch_tokens(self, rgb, keep_dtype=False):
        """rgb (B,H,W,C) u8 on the device -> (B, g, g, D) contiguous: fp32 like the reference's _get_patch_token, or
        (keep_dtype) the encoder's own bf16, which bsc_ingest_typed widens exactly on load."""
        if self.can_fuse_preprocess(rgb):
            t = self._forward_patches(self.preprocess_patches(rgb), keep_dtype)["x_norm_patchtokens"]
        elif self.can_fuse_preprocess_f32(rgb):
            t = self._forward_f32_split(self.preprocess_patches(rgb, mode=2), True)
        else:
            t = self.forward_features(self.preprocess(rgb))["x_norm_patchtokens"]
        return t.reshape(rgb.shape[0], self.grid, self.grid, -1).contiguous()

    def flops_per_frame(self):
        s = VIT_SHAPES[self.arch]
        T = 1 + self.registers + self.grid * self.grid
        w, m = s["width"], s["mlp"]
        per_layer = 2 * T * w * 3 * w + 2 * T * w * w + 2 * 2 * T * w * m + 2 * 2 * T * T * w
        return s["depth"] * per_layer + 2 * self.grid * self.grid * 3 * self.patch * self.patch * w


class GraphedEncoder:
    """Replays the encoder for a fixed batch shape from a captured HIP graph (launch-bound at small batch).  The graph starts
    at the patch matrix: the fused preprocessing kernel runs eagerly on the caller's frames and writes the graph's input, so
    the frames themselves are never copied.  Both precisions: the bf16 encoder (bf16 patch matrix) and the f32 encoder on
    split-operand GEMMs (patch matrix as fp16 pieces, or f32 rows when 3 p^2 is not a multiple of 32)."""

    def __init__(self, vit, batch, H, W, channels=4, keep_dtype=False):
        self.vit, self.keep_dtype = vit, keep_dtype
        probe = torch.zeros((batch, H, W, channels), dtype=torch.uint8, device="cuda")
        self.f32 = (not vit.can_fuse_preprocess(probe)) and vit.can_fuse_preprocess_f32(probe)
        self.pp_mode = 0
        if self.f32:
            self.pp_mode = 2
        self.from_patches = (vit.can_fuse_preprocess(probe) or self.f32) and os.environ.get("BSC_GRAPH_COPY") is None   # A/B switch
        s = torch.cuda.Stream()
        # this graph's attention counters: allocated before the capture (outside the graph's private pool), used by every
        # attention launch of the warm-up and of the captured forward
        self.att_work = torch.zeros(2, dtype=torch.int32, device="cuda")
        _ATT_WORK_OVERRIDE.append(self.att_work)
        try:
            self._capture(vit, batch, probe, keep_dtype, s)
        finally:
            _ATT_WORK_OVERRIDE.pop()

    def _capture(self, vit, batch, probe, keep_dtype, s):
        if self.from_patches and self.f32:
            self.static_in = vit.preprocess_patches(probe, mode=self.pp_mode)
            run = lambda: vit._forward_f32_split(self.static_in, self.pp_mode == 2).reshape(batch, vit.grid, vit.grid, -1)
        elif self.from_patches:
            self.static_in = vit.preprocess_patches(probe)
            run = lambda: vit._forward_patches(self.static_in, keep_dtype)["x_norm_patchtokens"].reshape(
                batch, vit.grid, vit.grid, -1)
        else:
            self.static_in = probe
            run = lambda: vit.patch_tokens(self.static_in, keep_dtype)
        s.wait_stream(torch.cuda.current_stream())          # after the kernel that writes the graph's input buffer
        with torch.cuda.stream(s):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = run()

    def __call__(self, rgb):
        if self.from_patches:
            self.vit.preprocess_patches(rgb if rgb.is_contiguous() else rgb.contiguous(), out=self.static_in, mode=self.pp_mode)
        else:
            self.static_in.copy_(rgb)
        self.graph.replay()
        return self.static_out
