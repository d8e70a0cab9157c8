import torch, time
S = torch.randn(256, 1048576 + 2112, device="cuda")
for name, fn in [("max_dim1", lambda: S.max(dim=1)), ("gt_sum", lambda: (S > 3.0).sum()), ("copy", lambda: S.clone())]:
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(name, f"{dt*1e3:.3f} ms", f"{S.numel()*4/dt/1e12:.2f} TB/s read")
