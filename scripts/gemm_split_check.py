"""Accuracy and speed of the split-operand GEMM (bsc_enc_gemm_split) against fp64 and against PyTorch's f32 GEMM, on the encoder's
shapes, f32-row and piece operands.  usage: gemm_split_check.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder
torch.manual_seed(0)
M = 384 * 197
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (K, N, epi, name) in ((768, 2304, 0, "qkv"), (768, 768, 2, "proj"), (768, 3072, 1, "fc1"), (3072, 768, 2, "fc2")):
    lin = torch.nn.Linear(K, N).cuda().float()
    torch.nn.init.trunc_normal_(lin.weight, std=0.02)
    A = torch.randn(M, K, device="cuda") * (0.3 if name in ("proj", "fc2") else 1.0)
    R = torch.randn(M, N, device="cuda")
    sl = encoder.SplitLinear(lin)
    asc = {"proj": 16.0, "fc2": 4.0}.get(name, 1.0)
    Ap = encoder.split_rows(A, asc)
    res = R if epi == 2 else None
    out = sl(A, epi, resid=res, a_scale=asc)
    outp = sl(Ap, epi, resid=res, a_scale=asc, a_pieces=True)
    rows = torch.randint(0, M, (2048,), device="cuda")
    ref64 = A[rows].double() @ lin.weight.double().t() + lin.bias.double()
    t32 = A[rows] @ lin.weight.t() + lin.bias
    if epi == 1:
        ref64 = torch.nn.functional.gelu(ref64, approximate="tanh"); t32 = torch.nn.functional.gelu(t32, approximate="tanh")
    if epi == 2:
        ref64 = ref64 + R[rows].double(); t32 = t32 + R[rows]
    errs = {"split(f32 rows)": (out[rows].double() - ref64).abs().max().item(), "split(pieces)": (outp[rows].double() - ref64).abs().max().item(),
            "torch f32": (t32.double() - ref64).abs().max().item()}
    if epi == 1:        # piece output: h + l of 4 * gelu
        cp = sl(Ap, epi, a_scale=asc, a_pieces=True, c_pieces_scale=4.0).view(M, N // 32, 2, 32)
        back = (cp[:, :, 0].float() + cp[:, :, 1].float()).reshape(M, N) / 4.0
        errs["split(pieces->pieces)"] = (back[rows].double() - ref64).abs().max().item()
    for fn, tag in ((lambda: sl(A, epi, resid=res, a_scale=asc), "split f32 rows"), (lambda: sl(Ap, epi, resid=res, a_scale=asc, a_pieces=True), "split pieces"),
                    (lambda: torch.addmm(lin.bias, A, lin.weight.t()), "torch f32")):
        dt = timeit(fn)
        print(f"{name:5s} {tag:15s} {dt * 1e3:7.3f} ms  {2 * M * N * K / dt / 1e12:6.1f} TFLOP/s (f32-equivalent)")
    print(f"{name:5s} max |err| vs fp64: " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()) + f"   (|out| rms {ref64.pow(2).mean().sqrt().item():.3f})")
ln = torch.nn.LayerNorm(768, eps=1e-6).cuda()
x = torch.randn(M, 768, device="cuda") * 3 + 0.5
p = encoder.layernorm_split(x, ln).view(M, 24, 2, 32)
back = (p[:, :, 0].float() + p[:, :, 1].float()).reshape(M, 768)
ref = torch.nn.functional.layer_norm(x.double(), (768,), ln.weight.double(), ln.bias.double(), 1e-6)
print(f"layernorm_split: max |err| vs fp64 {(back.double() - ref).abs().max().item():.2e} (torch f32 {(torch.nn.functional.layer_norm(x, (768,), ln.weight, ln.bias, 1e-6).double() - ref).abs().max().item():.2e}), {timeit(lambda: encoder.layernorm_split(x, ln)) * 1e3:.3f} ms")
