"""Encoder-only loop (ViT-B/16, 128 frames per call) for rocprofv3: isolated kernel durations of one forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shutil, tempfile
_src = os.environ.get("BSC_TUNABLEOP_FILE") or os.path.join(ROOT, "bsc-nav_amd", "tunableop_gfx950.csv")
_dst = os.path.join(tempfile.gettempdir(), f"bsc_tunableop_{os.getpid()}_.csv")
if not os.environ.get("BSC_TUNE_FRESH"):
    shutil.copy(_src, _dst[:-4] + "0.csv")
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING=os.environ.get("BSC_TUNING", "1"), PYTORCH_TUNABLEOP_FILENAME=_dst, PYTORCH_TUNABLEOP_VERBOSE="0")
import torch
if os.environ.get("BSC_FA"):
    print("fa library ->", os.environ["BSC_FA"], torch.backends.cuda.preferred_rocm_fa_library(os.environ["BSC_FA"]))
from bsc_nav_amd import encoder
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
vit = encoder.RandomViT(arch, image_size=224, seed=0).cuda()
rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
for _ in range(3):
    vit.patch_tokens(rgb)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for _ in range(n):
    vit.patch_tokens(rgb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("tunableop file:", _dst[:-4] + "0.csv")
print(f"{arch} B={B}: {dt * 1e3:.2f} ms per forward, {vit.flops_per_frame() * B / dt / 1e12:.0f} TFLOP/s")
