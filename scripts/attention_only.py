"""Attention-only loop (bsc_enc_attention_dyn on random qkv): kernel time and PMC counters of k_attention alone.
usage: attention_only.py [B] [T] [heads] [reps]"""
import ctypes as C, sys, time, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 197
H = int(sys.argv[3]) if len(sys.argv) > 3 else 12
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 24
lib = _lib.load()
qkv = (torch.randn((B, T, 3, H, 64), device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty((B, T, H * 64), dtype=torch.bfloat16, device="cuda")
work = torch.zeros(2, dtype=torch.int32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
run = lambda: _lib.check(lib.bsc_enc_attention_dyn(C.c_void_p(qkv.data_ptr()), B, T, H, 64, C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(work.data_ptr()), st))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
# reference: f32 softmax attention of the same bf16 inputs
q, k, v = (qkv[:8, :, i].float().permute(0, 2, 1, 3) for i in range(3))
ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(8, T, H * 64)
err = (out[:8].float() - ref).abs().max().item()
fl = 4.0 * T * T * 64 * B * H
print(f"k_attention B={B} T={T} H={H}: {ms * 1e3:.1f} us per launch, {fl / ms / 1e9:.1f} TFLOP/s, {(B * T * H * 64 * 2 * 4) / ms / 1e6:.0f} GB/s of qkv+out, max abs err vs f32 SDPA {err:.4f}")
