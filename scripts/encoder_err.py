"""Error of the bf16 encoder paths against the same ViT evaluated in f32 with plain PyTorch ops."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bsc_nav_amd import encoder
for arch in ("vit_b16", "vit_l14"):
    torch.manual_seed(1)
    vit = encoder.RandomViT(arch, seed=5).cuda()
    for prm in (vit.cls, vit.pos) + ((vit.reg,) if vit.reg is not None else ()):
        prm.data = (0.5 * torch.randn_like(prm.float())).to(prm.dtype)
    for blk in vit.blocks:                      # biases that matter
        for lin in (blk.proj, blk.fc2, blk.fc1, blk.qkv):
            lin.bias.data = (0.1 * torch.randn_like(lin.bias.float())).to(lin.bias.dtype)
    rgb = torch.randint(0, 255, (2, 60, 80, 4), dtype=torch.uint8, device="cuda")
    ref = copy.deepcopy(vit).float()
    ref.compute_dtype = torch.float32
    ref.fused = False
    r = ref.patch_tokens(rgb)
    res = {}
    for name, fused, lag in (("lagged", True, True), ("add_ln", True, False), ("unfused", False, False)):
        vit.fused, vit.lagged = fused, lag
        a = vit.patch_tokens(rgb)
        res[name] = a
        e = (a - r).abs()
        print(f"{arch} {name:8s} vs f32: max {e.max().item():.4f} mean {e.mean().item():.5f}")
    d = (res["lagged"] - res["unfused"]).abs()
    print(f"{arch} lagged vs unfused: max {d.max().item():.4f} mean {d.mean().item():.5f}")
