"""Socket power and engine clock (rocm-smi) while ONE kernel of the f32 forward runs back to back for ~1.5 s each (768 frames)."""
import subprocess, sys, threading, time
sys.path.insert(0, "/root/repo")
import torch
from bsc_nav_amd import encoder as E
frames = 768
vit = E.RandomViT("vit_b16", image_size=224, seed=0, dtype=torch.float32).cuda()
blk = vit.blocks[0]
T, Wd, heads = 197, 768, 12
M = frames * T
x = torch.randn((frames * 196, Wd), device="cuda")
u, _, (stats, mu) = E.embed_tokens_f32(vit, x, frames, ln=None, stats=True)
SL = E.SplitLinear
qkv_l, fc1_l = vit._split(blk.qkv, blk.ln1), vit._split(blk.fc1, blk.ln2)
y = E.layernorm_split(u, blk.ln1)
qkv = qkv_l(u, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=1.0)
att = E.attention_split(qkv, frames, T, heads, out_scale=16.0)
h = fc1_l(u, SL.GELU, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=4.0)
jobs = {
    "qkv (LN in load)": (lambda: qkv_l(u, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=1.0, out=qkv), 2.0 * M * Wd * 3 * Wd),
    "qkv (pieces)": (lambda: vit._split(blk.qkv)(y, a_pieces=True, c_pieces_scale=1.0, out=qkv), 2.0 * M * Wd * 3 * Wd),
    "fc1 (LN in load)": (lambda: fc1_l(u, SL.GELU, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=4.0, out=h), 2.0 * M * Wd * 3072),
    "fc2 (resid+stats)": (lambda: vit._split(blk.fc2)(h, SL.RESID, resid=u, out=u, a_scale=4.0, a_pieces=True, ln_stats=stats, ln_mu=mu), 2.0 * M * Wd * 3072),
    "attention": (lambda: E.attention_split(qkv, frames, T, heads, out_scale=16.0), 4.0 * frames * heads * T * T * 64),
    "layernorm pass": (lambda: E.layernorm_split(u, blk.ln1), 0.0),
}
samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            f = o.strip().splitlines()[-1].split(",")
            samples.append((time.perf_counter(), f[5], f[-1]))
        except Exception as e:
            pass
        time.sleep(0.03)
threading.Thread(target=sampler, daemon=True).start()
for name, (fn, flops) in jobs.items():
    fn(); torch.cuda.synchronize()
    n = 0
    t0 = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(20): fn()
        n += 20
        torch.cuda.synchronize()
    ev[1].record(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    us = ev[0].elapsed_time(ev[1]) / n * 1e3
    sm = [(c, p) for t, c, p in samples if t0 + 0.5 <= t <= t1]
    clk = sorted(int(c.strip("()Mhz")) for c, p in sm)
    pw = sorted(float(p) for c, p in sm)
    print(f"{name:20s} {us:8.1f} us  {3 * flops / us / 1e6:7.1f} TF fp16 MFMA   clock median {clk[len(clk)//2] if clk else 0} MHz  power median {pw[len(pw)//2] if pw else 0:.0f} W  ({len(sm)} samples)")
    time.sleep(0.5)
stop = True
