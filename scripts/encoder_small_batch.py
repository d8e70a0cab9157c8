import sys, time, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder as E
torch.cuda.set_stream(torch.cuda.Stream())
for arch in ("vit_b16", "vit_l14"):
    for dt in (torch.float32, torch.bfloat16):
        vit = E.RandomViT(arch, image_size=224, seed=0, dtype=dt).cuda()
        for B in (1, 8):
            rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
            for _ in range(3): vit.patch_tokens(rgb)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): vit.patch_tokens(rgb)
            torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20 * 1e3
            g = E.GraphedEncoder(vit, B, 480, 640, 4, False)
            for _ in range(3): g(rgb)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): g(rgb)
            torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20 * 1e3
            print(f"{arch} {str(dt)[6:]:9s} B={B}: eager {te:.3f} ms  graph {tg:.3f} ms")
