"""The f32 encoder with the in-tree split-operand GEMMs against the same module on PyTorch f32 GEMMs and against an fp64 evaluation.
usage: encoder_f32_check.py [arch] [frames]"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
vit = encoder.RandomViT(arch, image_size=224, seed=0, dtype=torch.float32).cuda()
rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
vit.split_gemm = True
a = vit.patch_tokens(rgb)
vit.split_gemm = False
b = vit.patch_tokens(rgb)
v64 = encoder.RandomViT(arch, image_size=224, seed=0, dtype=torch.float64).cuda()
v64.fused = False
n = min(B, 8)
c = v64.forward_features(v64.preprocess(rgb[:n]).double())["x_norm_patchtokens"].reshape(n, vit.grid, vit.grid, -1)
print(f"tokens rms {c.pow(2).mean().sqrt().item():.3f}; split vs torch-f32: max {(a - b).abs().max().item():.2e} mean {(a - b).abs().mean().item():.2e}; "
      f"split vs fp64: max {(a[:n].double() - c).abs().max().item():.2e}; torch-f32 vs fp64: max {(b[:n].double() - c).abs().max().item():.2e}")
for flag in (True, False):
    vit.split_gemm = flag
    for _ in range(2): vit.patch_tokens(rgb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): vit.patch_tokens(rgb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"split_gemm={flag}: {dt * 1e3:.2f} ms per {B} frames, {vit.flops_per_frame() * B / dt / 1e12:.0f} TFLOP/s f32-equivalent")
