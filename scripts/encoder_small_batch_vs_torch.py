import sys, time, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder as E
torch.cuda.set_stream(torch.cuda.Stream())
for arch in ("vit_b16", "vit_l14"):
    vit = E.RandomViT(arch, image_size=224, seed=0, dtype=torch.float32).cuda()
    for B in [int(v) for v in sys.argv[1:]] or (1, 8, 32, 64):
        rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
        out = {}
        for split in (True, False):
            vit.split_gemm = split
            for _ in range(3): vit.patch_tokens(rgb)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): vit.patch_tokens(rgb)
            torch.cuda.synchronize(); out[split] = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{arch} f32 B={B}: in-tree split GEMMs {out[True]:.3f} ms   PyTorch f32 GEMMs + SDPA {out[False]:.3f} ms")
