"""A/B: the split GEMM reading its activation operand as pre-split fp16 pieces against f32 rows split in registers (what a
LayerNorm folded into the A load would run on).  usage: gemm_a_mode_ab.py [frames]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 768
M = frames * 197
for name, (K, N, epi, cps) in {"qkv": (768, 2304, 0, 1.0), "fc1": (768, 3072, 1, 4.0)}.items():
    lin = torch.nn.Linear(K, N).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    A = torch.randn(M, K, device="cuda")
    sl = encoder.SplitLinear(lin)
    Ap = encoder.split_rows(A, 1.0)
    for label, a, ap in (("pieces", Ap, True), ("f32 rows", A, False)):
        out = sl(a, epi, a_pieces=ap, c_pieces_scale=cps)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(10):
            sl(a, epi, a_pieces=ap, c_pieces_scale=cps, out=out)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 10
        print(f"{name:4s} M={M} A as {label:9s}: {1e3 * ms:8.1f} us  {3 * 2.0 * M * K * N / ms / 1e9:7.1f} TF fp16 MFMA")
