"""Does a concurrent pinned-host -> device copy slow the f32 encoder down?  The captured forward (768 frames) timed alone, with
one 1.9 GB H2D copy per forward on a second stream, and with the same bytes as a device -> device copy."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from bsc_nav_amd import encoder
B = 768
torch.cuda.set_stream(torch.cuda.Stream())
vit = encoder.RandomViT("vit_b16", image_size=224, seed=0, dtype=torch.float32).cuda()
rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
g = encoder.GraphedEncoder(vit, B, 480, 640, 4, False)
h = torch.empty((B, 480, 640, 8), dtype=torch.uint8, pin_memory=True)
d = torch.empty((B, 480, 640, 8), dtype=torch.uint8, device="cuda")
d2 = torch.empty_like(d)
cs = torch.cuda.Stream()
def run(mode, n=12):
    for _ in range(2): g(rgb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if mode == "h2d":
            with torch.cuda.stream(cs): d.copy_(h, non_blocking=True)
        elif mode == "d2d":
            with torch.cuda.stream(cs): d2.copy_(d, non_blocking=True)
        elif mode == "h2d_slow":
            with torch.cuda.stream(cs):
                for lo in range(0, B, 96):
                    d[lo:lo + 96].copy_(h[lo:lo + 96], non_blocking=True)
        g(rgb)
        torch.cuda.current_stream().wait_stream(cs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for mode in ("alone", "h2d", "alone", "d2d", "h2d_slow", "alone"):
    print(f"{mode:9s}: {run(mode):7.2f} ms per forward")
