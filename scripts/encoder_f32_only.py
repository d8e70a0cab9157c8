"""f32 encoder loop (the reference-precision leg) for rocprofv3: isolated kernel durations of one forward.
usage: encoder_f32_only.py [arch] [frames] [split: 0 = PyTorch f32 GEMMs, 1 = in-tree split-operand MFMA GEMMs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bsc_nav_amd import encoder
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 384
if len(sys.argv) > 3:
    os.environ["BSC_ENC_SPLIT_GEMM"] = sys.argv[3]
vit = encoder.RandomViT(arch, image_size=224, seed=0, dtype=torch.float32).cuda()
rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
for _ in range(2):
    vit.patch_tokens(rgb)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 4
for _ in range(n):
    vit.patch_tokens(rgb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{arch} f32 B={B}: {dt * 1e3:.2f} ms per forward, {vit.flops_per_frame() * B / dt / 1e12:.0f} TFLOP/s")
