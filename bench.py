#!/usr/bin/env python3
"""bench.py — RGB-D frames/s into the voxel feature memory (BASELINE.json metric), one rank per GPU.

A step = one batch of synthetic 640x480 RGB-D frames through the hot path, inputs resident in HBM:
    ViT-B/16 patch features (random weights, bf16 MFMA via PyTorch-ROCm)  ->  libbscnav bsc_ingest
    (fp64 unprojection, first-touch voxel ids, rgb chain, top-down map, dense per-voxel feature reduce).
Workload: BASELINE.json configs[1] — 640x480 frames, 768-D tokens (14x14 patch grid), 256^3 grid of 0.1 m
cells, every pixel ingested (depth_sample_rate 1), "room" depth (camera random-walking inside an 8x3x6 m box).
With N>1 ranks each rank ingests its own frame shard (weak scaling) and the per-rank maps are merged by one
RCCL reduce-scatter at the end of the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant hand-written kernel (k_dense_reduce, HBM
bound): algorithmic bytes / HIP-event time measured live; `cpu_baseline` is the plain-C oracle (port of
the reference loop) timed on this box's host cores over a bounded sample of the same frames.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Library-GEMM selection for the encoder: PyTorch TunableOp picks the fastest hipBLASLt / rocBLAS solution per GEMM
# shape.  The choices for the default shapes are committed (bsc-nav_amd/tunableop_gfx950.csv); shapes not in the file
# are tuned during the warm-up steps, before the clock starts.  Must be configured before torch is imported.
if os.environ.get("BSC_TUNABLEOP", "1") == "1" and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ:
    import shutil
    import tempfile
    _src = os.path.join(ROOT, "bsc-nav_amd", "tunableop_gfx950.csv")
    _ord = os.environ.get("LOCAL_RANK", "0")
    _dst = os.path.join(tempfile.gettempdir(), f"bsc_tunableop_{os.getpid()}_.csv")
    if os.path.exists(_src):
        shutil.copy(_src, _dst[:-4] + _ord + ".csv")      # TunableOp reads/writes <name><device ordinal>.csv
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=_dst,
                      PYTORCH_TUNABLEOP_VERBOSE="0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def pmc_traffic():
    """HBM bytes per k_dense_reduce launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_ingest_kernels.json): (2 x FETCH_SIZE + WRITE_SIZE) KiB -- gfx950 FETCH_SIZE counts wide
    coalesced reads at half (MI355X_MICROARCH.md, HBM).  None when the file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_ingest_kernels.json")) as f:
            ks = json.load(f)["kernels"]
        k = next(v for name, v in ks.items() if name.startswith("void k_dense_reduce<"))
        return (2.0 * k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024.0
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=384, help="frames per step (per rank); 8 steps x 384 = 3072 frames")
    ap.add_argument("--tokens", choices=["bf16", "f32"], default="f32",
                    help="dtype the encoder hands to bsc_ingest: f32 like the reference's _get_patch_token, or its "
                         "native bf16 (bsc_ingest_typed widens it exactly on load)")
    ap.add_argument("--kind", default="room", choices=["room", "iid"])
    ap.add_argument("--mode", default="mean", choices=["mean", "max"])
    ap.add_argument("--arch", default="vit_b16")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run the encoder eagerly instead of a HIP graph")
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap encoder(s+1) with ingest(s)")
    ap.add_argument("--prefetch", type=int, default=1, help="batches the encoder runs ahead of the ingest")
    ap.add_argument("--priority", action="store_true", help="ingest on a high-priority stream (pair with --prefetch 2)")
    ap.add_argument("--no-localize", action="store_true", help="skip the localize top-K latency measurement")
    ap.add_argument("--no-iid", action="store_true", help="skip the iid-depth (HBM-bound regime) measurement of k_dense_reduce")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU")
    backend = os.environ.get("BSC_BENCH_BACKEND", "nccl")      # "gloo" only to exercise the N>1 path on a 1-GPU box
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic, encoder, dist as bdist

    H, W, gs, cs = a.height, a.width, a.grid, 0.1
    vit = encoder.RandomViT(a.arch, image_size=224, seed=0).cuda()
    g, D = vit.grid, vit.out_dim
    half = gs * cs / 2.0
    N = H * W
    n_steps = a.steps + a.warmup
    n_frames = n_steps * a.batch
    vcap = 3_000_000 if a.kind == "room" else min(gs ** 3, max(3_000_000, n_frames * N))
    # the ingest is the latency-critical stage of the pipeline: its stream (and the library's side stream) are
    # high priority, the MFMA-bound encoder fills the rest of the machine from a normal-priority stream
    ing_stream = torch.cuda.Stream(priority=-1 if a.priority else 0)
    torch.cuda.set_stream(ing_stream)
    eng = B.VoxelEngine(H, W, gs, cs, -half, half, g, D, mode=a.mode, voxel_capacity=vcap, max_points=a.batch * N,
                        device=local_rank)
    # ---- synthetic frames of this rank's shard, resident in HBM before the clock starts ----
    poses = synthetic.random_walk_poses(1000 + rank, n_frames)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    rgbs, depths = [], []
    for s in range(n_steps):
        r, d, _ = synthetic.make_frames(17 + 1000 * rank + s, a.batch, H, W, a.kind, device="cuda",
                                        poses=poses[s * a.batch:(s + 1) * a.batch])
        rgbs.append(r)
        depths.append(d)
    # Two-stage software pipeline: the encoder of batch s+1 (MFMA-bound) runs on its own HIP stream while
    # bsc_ingest of batch s (HBM / latency-bound) runs on the main stream; tokens are double-buffered.
    NBUF = a.prefetch + 1   # token buffers: the encoder runs `prefetch` batches ahead of bsc_ingest
    if a.no_graph:
        encs = [lambda r: vit.patch_tokens(r, a.tokens == "bf16")] * NBUF
    else:
        encs = [encoder.GraphedEncoder(vit, a.batch, H, W, 4, a.tokens == "bf16") for _ in range(NBUF)]
    enc = encs[0]
    enc_stream = torch.cuda.Stream()
    main_stream = torch.cuda.current_stream()
    tok_ready = [torch.cuda.Event() for _ in range(NBUF)]
    tok_free = [torch.cuda.Event() for _ in range(NBUF)]
    pending = {}

    def encode_async(s):
        b = s % NBUF
        with torch.cuda.stream(enc_stream):
            enc_stream.wait_event(tok_free[b])          # ingest of batch s-2 no longer reads this token buffer
            pending[s] = encs[b](rgbs[s])
            tok_ready[b].record(enc_stream)

    def step(s, stop):
        for k in range(0 if a.no_overlap else NBUF):
            if s + k < stop and s + k not in pending:
                encode_async(s + k)
        if s not in pending:
            encode_async(s)
        b = s % NBUF
        main_stream.wait_event(tok_ready[b])
        eng.ingest(depths[s], rgbs[s], pending.pop(s), Ts[s * a.batch:(s + 1) * a.batch])
        tok_free[b].record(main_stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    bdist.warmup_collectives(torch.device("cuda", local_rank))
    for b in range(NBUF):
        tok_free[b].record(main_stream)
    for s in range(a.warmup):
        step(s, a.warmup)
    barrier()
    c0 = eng.counters()
    eng.kernel_stats(0, reset=True)
    t0 = time.perf_counter()
    for s in range(a.warmup, n_steps):
        step(s, n_steps)
    merge_info = None
    if world > 1:
        merge_info = bdist.merge_dense_maps(eng)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ks = eng.kernel_stats(0)
    c1 = eng.counters() if world == 1 else None

    out = None
    if rank == 0:
        frames = a.steps * a.batch * world
        out = {
            "metric": "RGB-D frames/sec into voxel feature memory", "value": frames / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 geometry / f32 features" + (" (bf16 tokens in)" if a.tokens == "bf16" else ""),
            "data": "synthetic",
            "config": {"workload": f"{a.batch * a.steps} synthetic {W}x{H} RGB-D frames per GPU ({a.kind} depth, every "
                                   f"pixel), {a.arch} random weights {D}-D tokens {g}x{g}, {gs}^3 grid of {cs} m cells, "
                                   f"dense {a.mode} reduce" + (", + RCCL reduce-scatter merge" if world > 1 else ""),
                       "frames_per_step": a.batch, "parallelism": f"frames sharded x{world}"},
        }
        if merge_info:
            out["config"]["merge"] = merge_info
    # ---- per-stage split (untimed extra pass) and roofline of the dominant hand-written kernel ----
    if rank == 0 and world == 1:
        launches = max(1, ks["launches"])
        tok_bytes = 2 if a.tokens == "bf16" else 4
        U = (c1["voxel_rmw"] - c0["voxel_rmw"]) / a.steps           # voxel rows touched per launch
        U_new = (c1["max_id"] - c0["max_id"]) / a.steps
        P_pass = (c1["points_passed"] - c0["points_passed"]) / a.steps
        n_pairs = (c1["pairs"] - c0["pairs"]) / a.steps
        # algorithmic bytes of one k_dense_reduce launch: accumulator rows RMW (new rows are
        # written only) + counts + token tile once + the sorted (voxel,frame,patch) pair list (key 8 B + count 4 B)
        alg = (2 * U - U_new) * D * 4 + 8 * U + a.batch * g * g * D * tok_bytes + 12 * n_pairs
        ms = ks["ms"] / launches
        achieved = alg / (ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_dense_reduce", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
                           "bytes_per_launch": alg, "ms_per_launch": ms, "voxel_rows_per_launch": U,
                           "points_per_launch": P_pass, "pairs_per_launch": n_pairs}
        # stage split (untimed extra pass): the encoder alone, then bsc_ingest alone.  The second half doubles as an
        # un-contended measurement of k_dense_reduce (in the timed region it shares the chip with the encoder's GEMMs).
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        lo, hi = a.warmup, min(n_steps, a.warmup + 8)
        tok = enc(rgbs[lo])
        e0.record()
        for s in range(lo, hi):
            tok = enc(rgbs[s])
        e1.record()
        ci0 = eng.counters()
        eng.kernel_stats(0, reset=True)
        for s in range(lo, hi):
            eng.ingest(depths[s], rgbs[s], tok, Ts[s * a.batch:(s + 1) * a.batch])
        e2.record()
        torch.cuda.synchronize()
        k = hi - lo
        ksi, ci1 = eng.kernel_stats(0), eng.counters()
        Ui = (ci1["voxel_rmw"] - ci0["voxel_rmw"]) / k
        alg_i = (2 * Ui - (ci1["max_id"] - ci0["max_id"]) / k) * D * 4 + 8 * Ui + a.batch * g * g * D * tok_bytes \
            + 12 * (ci1["pairs"] - ci0["pairs"]) / k
        ms_i = ksi["ms"] / max(1, ksi["launches"])
        out["roofline_isolated"] = {"kernel": "k_dense_reduce", "note": "same workload, bsc_ingest running alone",
                                    "achieved": alg_i / (ms_i * 1e-3) / 1e9, "unit": "GB/s",
                                    "frac": alg_i / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_launch": ms_i,
                                    "bytes_per_launch": alg_i}
        enc_ms, ing_ms = e0.elapsed_time(e1) / k, e1.elapsed_time(e2) / k
        out["stages"] = {"encoder_ms_per_step": enc_ms, "ingest_ms_per_step": ing_ms,
                         "encoder_tflops": vit.flops_per_frame() * a.batch / (enc_ms * 1e-3) / 1e12,
                         "voxels": c1["max_id"]}
        if not a.no_iid:
            # The same kernel in the regime the HBM roofline describes: "iid" depth (one voxel per point,
            # U ~ P), where it is a pure accumulator read-modify-write stream instead of an L2-resident token re-read.
            Fi, calls = 16, 4
            engI = B.VoxelEngine(H, W, gs, cs, -half, half, g, D, mode=a.mode, voxel_capacity=min(gs ** 3, calls * Fi * N),
                                 max_points=Fi * N, device=local_rank)
            pi = synthetic.random_walk_poses(77, calls * Fi)
            chain_i = B.PoseChain()
            Ti = np.stack([chain_i.pc_transform(p) for p in pi])
            toki = torch.randn((Fi, g, g, D), device="cuda")
            fr = [synthetic.make_frames(900 + s, Fi, H, W, "iid", device="cuda", poses=pi[s * Fi:(s + 1) * Fi])
                  for s in range(calls)]
            engI.ingest(fr[0][1], fr[0][0], toki, Ti[:Fi])              # first call: every voxel is new (write-only rows)
            torch.cuda.synchronize()
            cj0 = engI.counters()
            engI.kernel_stats(0, reset=True)
            for s in range(1, calls):
                engI.ingest(fr[s][1], fr[s][0], toki, Ti[s * Fi:(s + 1) * Fi])
            torch.cuda.synchronize()
            ksj, cj1 = engI.kernel_stats(0), engI.counters()
            kk = calls - 1
            Uj = (cj1["voxel_rmw"] - cj0["voxel_rmw"]) / kk
            alg_j = (2 * Uj - (cj1["max_id"] - cj0["max_id"]) / kk) * D * 4 + 8 * Uj + Fi * g * g * D * 4 \
                + 12 * (cj1["pairs"] - cj0["pairs"]) / kk
            ms_j = ksj["ms"] / max(1, ksj["launches"])
            out["roofline_iid"] = {"kernel": "k_dense_reduce", "note": f"iid depth, {Fi} frames per call, bsc_ingest alone",
                                   "achieved": alg_j / (ms_j * 1e-3) / 1e9, "unit": "GB/s",
                                   "frac": alg_j / (ms_j * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_launch": ms_j,
                                   "bytes_per_launch": alg_j, "voxel_rows_per_launch": Uj}
            engI.close()
            del fr, toki
        if not a.no_localize:
            # second half of the metric: localize top-K latency over a 2^20-voxel x D map (BASELINE configs[3]/[4] size)
            V = 1 << 20
            gL = 512
            engL = B.VoxelEngine(H, W, gL, cs, -gL * cs / 2, gL * cs / 2, g, D, mode="mean", voxel_capacity=V + 8,
                                 max_points=1024, device=local_rank)
            gen = torch.Generator(device="cuda").manual_seed(5)
            codes = torch.randperm(gL ** 3, device="cuda", generator=gen)[:V]
            keys = torch.stack([codes // (gL * gL), (codes // gL) % gL, codes % gL], dim=1).to(torch.int32).contiguous()
            rows = torch.randn((V, D), device="cuda", generator=gen)
            engL.dense_replace(keys, rows, torch.ones(V, dtype=torch.int32, device="cuda"))
            del rows
            loc = {"voxels": V, "dim": D, "K": 100}
            for Q in (1, 8, 256):
                q = torch.randn(Q, D, device="cuda", generator=gen)
                engL.localize(q, K=100)
                engL.kernel_stats(1, reset=True)
                torch.cuda.synchronize()
                t = time.perf_counter()
                reps = 10
                for _ in range(reps):
                    engL.localize(q, K=100)
                torch.cuda.synchronize()
                lat = (time.perf_counter() - t) / reps
                ls = engL.kernel_stats(1)
                loc[f"q{Q}"] = {"latency_ms": lat * 1e3, "cosine_ms": ls["ms"] / max(1, ls["launches"]),
                                "cosine_GBs": ls["bytes"] / max(1e-9, ls["ms"] * 1e-3) / 1e9,
                                "cosine_frac_of_hbm_peak": ls["bytes"] / max(1e-9, ls["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                if Q >= 16:     # fp32-MFMA GEMM path: price it against the 157.3 TFLOP/s fp32 matrix peak instead
                    tf = 2.0 * V * D * Q / max(1e-9, ls["ms"] / max(1, ls["launches"]) * 1e-3) / 1e12
                    loc[f"q{Q}"].update({"cosine_TFLOPs": tf, "cosine_frac_of_f32_mfma_peak": tf / 157.3})
                    loc[f"q{Q}"].pop("cosine_GBs"), loc[f"q{Q}"].pop("cosine_frac_of_hbm_peak")
            out["localize"] = loc
            engL.close()
    # ---- CPU baseline: the plain-C oracle (port of the reference loop) on a bounded sample ----
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as orc
        oc = orc.make_config(H, W, gs, cs, -half, half, g, D, mode=1 if a.mode == "mean" else 2)
        om = orc.OracleMemory(oc, voxel_capacity=2_000_000)
        host = []
        for s in range(min(n_steps, 2)):                # up to 256 frames staged on the host, outside the CPU clock
            host.append((vit.patch_tokens(rgbs[s]).cpu().numpy(), rgbs[s].cpu().numpy(), depths[s].cpu().numpy()))
        nf, cdt = 0, 0.0
        for s, (tok_h, rgb_h, dep_h) in enumerate(host):
            for f in range(a.batch):
                t = time.perf_counter()
                om.ingest_frame(dep_h[f], rgb_h[f], None, Ts[s * a.batch + f], tok_h[f])
                cdt += time.perf_counter() - t
                nf += 1
                if cdt >= a.cpu_seconds and nf >= 2:
                    break
            if cdt >= a.cpu_seconds and nf >= 2:
                break
        out["cpu_baseline"] = {"value": nf / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"first {nf} frames of the same workload through oracle/bsc_oracle.c "
                                         f"(memory path only: geometry + voxel scatter, encoder excluded), "
                                         f"{cdt:.1f} s on 1 of {os.cpu_count()} host cores"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
