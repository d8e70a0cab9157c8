"""One encoder GEMM shape through bsc_enc_gemm_split, in a loop, for rocprofv3 counters.  usage: gemm_split_prof.py [qkv|proj|fc1|fc2] [reps] [frames]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder
name = sys.argv[1] if len(sys.argv) > 1 else "qkv"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K, N, epi = {"qkv": (768, 2304, 0), "proj": (768, 768, 2), "fc1": (768, 3072, 1), "fc2": (3072, 768, 2)}[name]
M = (int(sys.argv[3]) if len(sys.argv) > 3 else 384) * 197
lin = torch.nn.Linear(K, N).cuda().float()
torch.nn.init.trunc_normal_(lin.weight, std=0.02)
A = torch.randn(M, K, device="cuda")
R = torch.randn(M, N, device="cuda")
sl = encoder.SplitLinear(lin)
Ap = encoder.split_rows(A, 1.0)
for _ in range(reps):
    sl(Ap, epi, resid=R if epi == 2 else None, a_pieces=True)
torch.cuda.synchronize()
