"""k_attention_split forms at the bench's shape, sustained (1.5 s each): BSC_ATT_SPLIT_MODE unset = 32x32x16 strips, 2 = 16x16x32.  usage: attention_split_ab.py [frames]"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bsc_nav_amd import encoder as E
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 768
T, H = 197, 12
qkv = E.split_rows(torch.randn(frames * T, 3 * H * 64, device="cuda") * 0.7, 1.0)
E.attention_split(qkv, frames, T, H, out_scale=16.0); torch.cuda.synchronize()
n, t0 = 0, time.perf_counter()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
while time.perf_counter() - t0 < 1.5:
    for _ in range(20): E.attention_split(qkv, frames, T, H, out_scale=16.0)
    n += 20; torch.cuda.synchronize()
ev[1].record(); torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) / n * 1e3
print(f"attention_split {frames} frames: {us:.1f} us  ({3 * 4.0 * frames * H * T * T * 64 / us / 1e6:.0f} TF fp16 MFMA)")
