"""Known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM: FETCH_SIZE
reports half of a wide coalesced read; other access patterns are uncalibrated).  Run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
and compare the per-kernel counter (KiB) with the bytes printed here (scripts/pmc_calibrate.sh does both and divides).
Working sets are 4 GiB (far beyond the 256 MiB Infinity Cache); gather indices are uniform random."""
import torch
torch.manual_seed(0)
dev = "cuda"
N = 1 << 30                                   # 4 GiB of f32
x = torch.ones(N, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
n_idx = 1 << 26
idx = torch.randint(0, N // 4, (n_idx,), device=dev, dtype=torch.int64)
rows = x.view(N // 4, 4)                      # 16-byte rows (the rgb chain gathers 12-byte records)
tok = torch.ones((1 << 21, 768), dtype=torch.bfloat16, device=dev)    # 3 GiB of 1536-byte rows (the reduce gathers token rows)
ridx = torch.randint(0, tok.shape[0], (1 << 22,), device=dev, dtype=torch.int64)
torch.cuda.synchronize()
cases = {}
for rep in range(2):
    s = x.sum()                                                # reduce_kernel: coalesced 16 B / lane stream read
    y.copy_(x)                                                 # copy: stream read + stream write
    y.fill_(2.0)                                               # fill: stream write
    g4 = x.view(-1)[idx]                                       # index kernel: 4-byte gathers
    g16 = rows[idx[:1 << 24]]                                  # 16-byte row gathers
    gt = tok.index_select(0, ridx)                             # 1536-byte row gathers
    torch.cuda.synchronize()
print("expected bytes per launch:")
print(f"  reduce_kernel (sum)        read {N * 4}")
print(f"  copy (elementwise copy)    read {N * 4} write {N * 4}")
print(f"  fill                       write {N * 4}")
print(f"  index (4 B gather)         read idx {n_idx * 8} + gather {n_idx * 4} useful / {n_idx * 64} in 64 B sectors; write {n_idx * 4}")
print(f"  index (16 B row gather)    read idx {(1 << 24) * 8} + gather {(1 << 24) * 16} useful / {(1 << 24) * 64} in 64 B sectors; write {(1 << 24) * 16}")
print(f"  index_select 1536 B rows   read idx {ridx.numel() * 8} + gather {ridx.numel() * 1536}; write {ridx.numel() * 1536}")
