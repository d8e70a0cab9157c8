"""How do the encoder's library GEMMs (hipBLASLt through PyTorch) react when a few CUs are held by another stream?
usage: gemm_vs_hog.py   (prints ms per GEMM alone / beside N single-wave spinning workgroups)"""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hog = C.CDLL(os.path.join(ROOT, "scripts", "microbench", "libhog.so"))
M = 75648
shapes = {"qkv": (768, 2304), "proj": (768, 768), "fc1": (768, 3072), "fc2": (3072, 768)}
sink = torch.zeros(4, device="cuda")
side = torch.cuda.Stream()
for name, (K, N) in shapes.items():
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    lin = torch.nn.Linear(K, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        lin(x)
    res = []
    for n_hog, thr in ((0, 64), (1, 64), (8, 64), (32, 64), (8, 1024)):
        torch.cuda.synchronize()
        if n_hog:
            hog.hog_launch(C.c_void_p(side.cuda_stream), n_hog, thr, C.c_double(20000.0), C.c_void_p(sink.data_ptr()))
            time.sleep(0.002)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lin(x)
        e1.record()
        e1.synchronize()
        res.append((n_hog, thr, e0.elapsed_time(e1) / 20))
        torch.cuda.synchronize()
    print(name, " ".join(f"hog{n}x{t}={ms:.3f}ms" for n, t, ms in res))
