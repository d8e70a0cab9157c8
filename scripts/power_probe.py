"""Is the f32 encoder's sustained rate set by the chip's power / clock management?  The same captured forward (768 frames) timed
back to back and with idle gaps between the replays; the engine clock and socket power sampled by rocm-smi in a side thread."""
import os, subprocess, sys, threading, time
sys.path.insert(0, "/root/repo")
import torch
from bsc_nav_amd import encoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
torch.cuda.set_stream(torch.cuda.Stream())
vit = encoder.RandomViT("vit_b16", image_size=224, seed=0, dtype=torch.float32).cuda()
rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
g = encoder.GraphedEncoder(vit, B, 480, 640, 4, False)
samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), o.strip().splitlines()[-1]))
        except Exception as e:
            samples.append((time.perf_counter(), repr(e)))
        time.sleep(0.05)
th = threading.Thread(target=sampler, daemon=True); th.start()
def one():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g(rgb); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for _ in range(3): one()
print("hdr", subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[0])
for gap in (0.0, 0.1, 0.5, 0.0):
    ts = []
    t_start = time.perf_counter()
    for i in range(16):
        ts.append(one())
        if gap: time.sleep(gap)
    t_end = time.perf_counter()
    sm = [s for t, s in samples if t_start <= t <= t_end]
    print(f"gap {gap:.1f} s: forward ms first {ts[0]:.1f} median {sorted(ts)[8]:.1f} last {ts[-1]:.1f} min {min(ts):.1f}")
    for s in sm[:: max(1, len(sm) // 3)][:3]:
        print("    ", s)
stop = True
