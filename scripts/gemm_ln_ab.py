"""A/B of the LayerNorm-fused forms of the split GEMM at the bench's shapes: qkv / fc1 reading pieces against LayerNorm in the operand
load; proj / fc2 residual epilogue without and with the statistics records.  usage: gemm_ln_ab.py [frames]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 768
M, Wd = frames * 197, 768
SL = encoder.SplitLinear

def timeit(fn, reps=10):
    fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3

u = torch.randn(M, Wd, device="cuda")
ln = torch.nn.LayerNorm(Wd, eps=1e-6).cuda()
vit = encoder.RandomViT("vit_b16", dtype=torch.float32).cuda()
x = torch.randn(frames * 196, Wd, device="cuda")
_, _, (stats, mu) = encoder.embed_tokens_f32(vit, x, frames, ln=None, stats=True)
y = encoder.layernorm_split(u, ln)
for name, N, epi, cps in (("qkv", 2304, 0, 1.0), ("fc1", 3072, 1, 4.0)):
    lin = torch.nn.Linear(Wd, N).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    a, b = SL(lin), SL(lin, ln)
    out = a(y, epi, a_pieces=True, c_pieces_scale=cps)
    t0 = timeit(lambda: a(y, epi, a_pieces=True, c_pieces_scale=cps, out=out))
    t1 = timeit(lambda: a(u, epi, c_pieces_scale=cps, out=out))
    t2 = timeit(lambda: b(u, epi, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=cps, out=out))
    print(f"{name}: pieces {t0:7.1f} us | f32 rows {t1:7.1f} | LayerNorm in the load {t2:7.1f}")
for name, K, sc in (("proj", 768, 16.0), ("fc2", 3072, 4.0)):
    lin = torch.nn.Linear(K, Wd).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    a = SL(lin)
    ap = encoder.split_rows(torch.randn(M, K, device="cuda"), sc)
    af = torch.randn(M, K, device="cuda")
    t0 = timeit(lambda: a(ap, 2, resid=u, out=u, a_scale=sc, a_pieces=True))
    t1 = timeit(lambda: a(ap, 2, resid=u, out=u, a_scale=sc, a_pieces=True, ln_stats=stats, ln_mu=mu))
    t2 = timeit(lambda: a(af, 2, resid=u, out=u, a_scale=sc, ln_stats=stats, ln_mu=mu))
    print(f"{name}: residual epilogue {t0:7.1f} us | + row statistics {t1:7.1f} | the same from f32 rows {t2:7.1f}")
print(f"layernorm pass: {timeit(lambda: encoder.layernorm_split(u, ln)):7.1f} us")
