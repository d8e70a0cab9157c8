#!/usr/bin/env python3
"""bench.py — RGB-D frames/s into the voxel feature memory (BASELINE.json metric), one rank per GPU.

A step = one batch (--batch, default 768) of synthetic 640x480 RGB-D frames through the hot path, inputs resident in HBM:
    ViT patch features at the REFERENCE'S precision (memory_2.py:43,738-739: DINOv2 runs f32) — random f32 weights, every dense
       layer, attention and LayerNorm in-tree: libbscnav's split-operand fp16-MFMA kernels (csrc/encoder_gemm.hip: an f32
       operand = two fp16 pieces, products hh + hl + lh in the f32 accumulator; tokens within 1e-5 of PyTorch f32), f32 tokens
    -> libbscnav bsc_ingest (fp64 unprojection, first-touch voxel ids, rgb chain, top-down map, dense per-voxel
       feature reduce).
Headline workload (`value`): BASELINE.json configs[1] — 640x480 frames, ViT-B/16 768-D tokens (14x14 patch grid), 256^3
grid of 0.1 m cells, every pixel ingested (depth_sample_rate 1), "room" depth (camera random-walking inside an 8x3x6 m
box).  The K timed steps are repeated (engine reset in between) and the MEDIAN repeat is reported.
`--precision bf16` times the opt-in fast mode instead (bf16 weights / activations / tokens, GEMMs through hipBLASLt); at the
default precision that configuration is measured as the side key `value_bf16_encoder_library_gemms`.
With N>1 ranks each rank ingests its own frame shard (weak scaling) and the per-rank maps are merged by one
RCCL reduce-scatter at the end of the timed region.  `--gpus N` without a torchrun environment launches the N ranks
itself (python -m torch.distributed.run on 127.0.0.1) and exits non-zero when fewer than N GPUs are visible; under
torchrun WORLD_SIZE must equal --gpus.

Prints ONE JSON line (rank 0) carrying, beside the contract keys:
  roofline        the whole bsc_ingest call priced per SURVEY.md §8(d): algorithmic bytes of the batch / WALL time of the call
                  followed by bsc_sync, alone (main stream + order stage + rgb chain on the side stream); the main-stream
                  HIP-event times ride as side keys; dominant kernel named.  Sub-blocks: `encoder` (bound mfma: the 16-bit
                  MFMA rate of the split-operand GEMMs against the 2.5 PFLOP/s peak, per-kernel rates from HIP events),
                  `k_points_valu` (the dominant ingest kernel against the vector-issue rate it is bound by),
                  `kernels` (every ingest stage with its own bytes)
  value_from_host the same pipeline with the frames starting in pinned HOST memory (as the reference's simulator hands them
                  over, memory_2.py:1090-1096): double-buffered hipMemcpyAsync on a copy stream under the previous step
  exact_mode      the reference-semantics mode (token cache, <= 10 raw tokens per voxel, random replacement, host-shuffled
                  sub-sampling at depth_sample_rate 1000 and 50): frames/s frame by frame through obs2voxeltoken and batched
  workloads       the same pipeline on "hall" (24 x 24 m, 10^5..10^6 voxels) and "iid" (one voxel per point) depth:
                  frames/s, voxels, U/P, fraction of the §8(d) HBM bound
  configs         BASELINE configs[2] per GPU (ViT-L/14 + 4 registers, 1024-D, 512^3; f32 like the headline, bf16 beside it)
                  and configs[3]/[4] localize at 2^20 x 1024
  cpu_baseline    the plain-C oracle (port of the reference loop) on this box's host cores: 1 core and all cores
  value_bf16_encoder_library_gemms / tokens_bf16_*   the opt-in bf16 mode and what it costs in accuracy
Every leg besides the headline is guarded: a failure leaves {"error": ...} under its key instead of losing the line.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
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
    _src = os.environ.get("BSC_TUNABLEOP_FILE") or os.path.join(ROOT, "bsc-nav_amd", "tunableop_gfx950.csv")    # A/B: another choice file
    _ord = os.environ.get("LOCAL_RANK", "0")
    _dst = os.path.join(tempfile.gettempdir(), f"bsc_tunableop_{os.getpid()}_.csv")
    if os.path.exists(_src):
        shutil.copy(_src, _dst[:-4] + _ord + ".csv")      # TunableOp reads/writes <name><device ordinal>.csv
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=_dst,
                      PYTORCH_TUNABLEOP_VERBOSE="0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PMC_FILE = "r06_pmc_ingest_kernels.json"
# k_points by its SQ counters (profiles/r06_pmc_sq_ingest.txt, collected at 725086e: SQ_INSTS_VALU 797.2 M, of them
# SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 136.4 M, SQ_INSTS_SALU 359.5 M per 460 800 wavefronts of 8 rounds of 64 points; round 5: 266.7 / 63 / 105.5)
KP_VALU_PER_64, KP_VALU_F64_PER_64, KP_SALU_PER_64 = 216.3, 37.0, 97.5
N_SIMD, CLOCK_GHZ = 1024, 2.4            # 256 CUs x 4 SIMDs; peak engine clock (MI355X_MICROARCH.md)
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), committed PMC pass (profiles/r06_pmc_mfma_counters.txt: the encoder's
# kernels with the exact erf GELU epilogue; the batched scan from profiles/r05_pmc_mfma_counters.txt, its kernel unchanged since)
MFMA_BUSY_COMMITTED = {"k_gemm_split qkv (LayerNorm in the load)": 0.438, "k_gemm_split fc1 + erf GELU (LayerNorm in the load)": 0.407,
                       "k_gemm_split proj / fc2 (residual + statistics)": 0.432, "k_attention_split": 0.269, "k_cosine_f16x2 (Q = 256)": 0.378,
                       "hipBLASLt bf16 (same shapes, round 4)": "0.43-0.47",
                       "source": "profiles/r06_pmc_mfma_counters.txt (encoder), profiles/r05_pmc_mfma_counters.txt (k_cosine_f16x2)"}
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0
MFMA_F32_PEAK_TF = 157.3
STAGES = {2: "k_points", 3: "k_keys_pairs", 4: "ids+point_order", 5: "pair_sort", 0: "k_dense_reduce", 6: "bsc_ingest",
          7: "k_chain"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=768,
                    help="frames per step (per rank).  Sweep on one MI355X, same box: 384: 20.5 k frames/s, 512: 21.0 k, 768: 21.7 k, "
                         "1024: 21.3 k, 1536: 21.3 k — the encoder's GEMMs run 6-8 %% faster per frame at M = 151 296 rows than at 75 648")
    ap.add_argument("--repeats", type=int, default=5, help="repetitions of the K timed steps; the median is reported")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="encoder precision of the headline: f32 = the reference's (f32 weights, activations and tokens; dense "
                         "layers, attention and LayerNorm in-tree on the fp16 matrix cores with split operands), bf16 = the opt-in "
                         "fast mode (bf16 weights / activations, library GEMMs)")
    ap.add_argument("--tokens", choices=["bf16", "f32"], default=None,
                    help="bf16 precision only — dtype the encoder hands to bsc_ingest: its native bf16 (default; bsc_ingest_typed, "
                         "the reduce widens and accumulates a row element with one v_dot2c_f32_bf16) or the bf16 results widened "
                         "to f32 by the encoder's last kernel; scripts/ab_tokens.sh.  The f32 encoder always hands over f32 tokens")
    ap.add_argument("--kind", default="room", choices=["room", "hall", "iid", "room_off"])
    ap.add_argument("--mode", default="mean", choices=["mean", "max"])
    ap.add_argument("--arch", default="vit_b16")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run the encoder eagerly instead of a HIP graph")
    ap.add_argument("--no-localize", action="store_true", help="skip the localize top-K latency measurements")
    ap.add_argument("--no-f32", "--no-side-precision", dest="no_side", action="store_true",
                    help="skip the other-precision side leg (bf16 + library GEMMs beside an f32 headline)")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the frames-from-pinned-host-memory leg")
    ap.add_argument("--no-workloads", action="store_true", help="skip the hall / iid workloads and the C3 leg")
    ap.add_argument("--no-exact", action="store_true", help="skip the reference-semantics (exact mode) leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc traffic passes of the isolated bsc_ingest call")
    return ap.parse_args()


def ingest_alg_bytes(F, N, g, D, tok_bytes, U, P_sampled=0):
    """SURVEY.md §8(d) bytes of a batch of F frames: depth + RGBA once, patch tokens once, RMW of the feature row and
    count of every voxel the batch touches, its rgb / weight / position, the sample indices."""
    return F * (8 * N + g * g * D * tok_bytes) + U * (2 * D * 4 + 8) + U * 2 * (3 + 4 + 12) + 4 * P_sampled


def pmc_traffic(frames_per_call, tok_bytes):
    """HBM-side bytes per bsc_ingest call from the committed rocprofv3 PMC passes of this same workload
    (profiles/r04_pmc_ingest_kernels.json, scripts/pmc_summary.py: read / write requests of the L2's memory side counted by
    request size — 32 / 64 / 128 B —, two separate --pmc passes, summed over the call's kernels; the counters are checked on
    known-byte kernels in profiles/r03_pmc_calibration.txt).  The file names the commit it was measured at; None when absent."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import pmc_summary
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            d = json.load(f)
        if d.get("frames_per_call", frames_per_call) != frames_per_call or d.get("token_bytes", 2) != tok_bytes:
            return None, None                           # measured at another --batch or token dtype: no figure
        if d.get("ingest_sources_sha16") != pmc_summary.ingest_sources_sha():
            return None, None                           # the ingest kernels changed after the profile was taken: a stale figure is not quoted
        return float(d["ingest_traffic_bytes_per_call"]), d.get("commit")
    except Exception:
        return None, None


def live_pmc_traffic(a, g, D, timeout_s=170):
    """HBM-side bytes per bsc_ingest call MEASURED IN THIS RUN: two rocprofv3 --pmc passes (read requests, write requests — separate
    runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes) of scripts/ingest_only.py, i.e. of the very call roofline.frac
    times: bsc_ingest + bsc_sync alone on the chip, the bench's frames / token rows / grid.  After the timed region, in a child
    process; None when rocprofv3 is missing or a pass fails (the committed figure is then quoted, if it still matches the tree)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pmc_summary
    csvs = []
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for tag, ctr in (("RD", "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"), ("WR", "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B")):
            cmd = [exe, "--pmc", *ctr.split(), "--kernel-trace", "--output-format", "csv", "-d", os.path.join(td, tag), "--", sys.executable,
                   os.path.join(ROOT, "scripts", "ingest_only.py"), "3", "sync", str(a.batch), a.kind, str(g), str(D), str(a.grid)]
            env = {k: v for k, v in os.environ.items() if not k.startswith(("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_", "PYTORCH_TUNABLEOP"))}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            found = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(td, tag)) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not found:
                return None
            csvs.append(found[0])
        d = pmc_summary.summarise(csvs[0], csvs[1], a.batch, commit="this run", command="rocprofv3 --pmc ... -- scripts/ingest_only.py 3 sync "
                                  f"{a.batch} {a.kind} {g} {D} {a.grid}")
    top = sorted(((k, (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_per_call"]) for k, v in d["kernels"].items() if v["ingest"]),
                 key=lambda kv: -kv[1])[:6]
    return {"bytes_per_call": float(d["ingest_traffic_bytes_per_call"]), "largest_kernels_GB_per_call": {k[:48]: round(b / 1e9, 3) for k, b in top}}


class Pipeline:
    """Encoder (HIP graph) then bsc_ingest, back to back on ONE stream; frames resident in HBM.  The library itself keeps
    its deferred rgb chain on a side stream.  (An encoder stream running ahead of the ingest bought <= 2 % in round 2 — the
    stages time-slice the chip — and was dropped.)  precision "f32": the reference's (in-tree split-operand kernels, f32
    tokens); "bf16": the opt-in fast mode.  share: another Pipeline whose frames and poses are reused (same workload)."""

    def __init__(self, a, kind, arch, grid, batch, n_steps, rank, local_rank, vit=None, vcap=None, precision=None, share=None):
        import bsc_nav_amd as B
        from bsc_nav_amd import synthetic, encoder
        self.B, self.a, self.kind, self.batch, self.n_steps = B, a, kind, batch, n_steps
        self.precision = precision or a.precision
        H, W, cs = a.height, a.width, 0.1
        self.H, self.W, self.N = H, W, H * W
        if vit is None:
            vit = encoder.RandomViT(arch, image_size=224, seed=0,
                                    dtype=torch.float32 if self.precision == "f32" else torch.bfloat16).cuda()
        self.vit = vit
        self.g, self.D = self.vit.grid, self.vit.out_dim
        half = grid * cs / 2.0
        n_frames = n_steps * batch
        if vcap is None:
            vcap = {"room": 3_000_000, "hall": 3_000_000}.get(kind, min(grid ** 3, 6_000_000))
        self.eng = B.VoxelEngine(H, W, grid, cs, -half, half, self.g, self.D, mode=a.mode, voxel_capacity=vcap,
                                 max_points=batch * self.N, device=local_rank)
        if share is not None:
            self.Ts, self.rgbs, self.depths = share.Ts, share.rgbs, share.depths
        else:
            poses = synthetic.make_poses(kind, 1000 + rank, n_frames)
            chain = B.PoseChain()
            self.Ts = np.stack([chain.pc_transform(p) for p in poses])
            self.rgbs, self.depths = [], []
            for s in range(n_steps):
                r, d, _ = synthetic.make_frames(17 + 1000 * rank + s, batch, H, W, kind, device="cuda",
                                                poses=poses[s * batch:(s + 1) * batch])
                self.rgbs.append(r)
                self.depths.append(d)
        self.tokens_bf16 = self.precision == "bf16" and a.tokens != "f32"
        self.tok_bytes = 2 if self.tokens_bf16 else 4
        if a.no_graph:
            self.enc = lambda r: self.vit.patch_tokens(r, self.tokens_bf16)
        else:
            self.enc = encoder.GraphedEncoder(self.vit, batch, H, W, 4, self.tokens_bf16)
        self.encs = [self.enc]
        self.enc_events = []              # (start, end) per encoder run

    def step(self, s, stop=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tok = self.enc(self.rgbs[s])
        e1.record()
        self.enc_events.append((e0, e1))
        self.eng.ingest(self.depths[s], self.rgbs[s], tok, self.Ts[s * self.batch:(s + 1) * self.batch])

    def run(self, lo, hi):
        for s in range(lo, hi):
            self.step(s, hi)

    def reset_stats(self):
        self.enc_events = []
        for w in STAGES:
            self.eng.kernel_stats(w, reset=True)

    def stage_ms(self):
        out = {}
        for w, name in STAGES.items():
            k = self.eng.kernel_stats(w)
            out[name] = k["ms"] / max(1, k["launches"])
        if self.enc_events:
            torch.cuda.synchronize()
            out["encoder"] = sum(a.elapsed_time(b) for a, b in self.enc_events) / len(self.enc_events)
        return out

    def isolated(self, lo, hi, power=False):
        """Untimed extra pass: the encoder alone, then bsc_ingest alone (un-contended stage times).  power: socket power sampled by
        rocm-smi in a side thread while the encoder runs alone (joules per frame of the forward)."""
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        enc = self.encs[0]
        tok = enc(self.rgbs[lo])
        watts = None
        if power:
            # enough forwards for a dozen samples (a rocm-smi call takes ~50 ms); the sampler sees only fully loaded seconds
            smp = PowerSampler()
            for _ in range(3):
                tok = enc(self.rgbs[lo])
            torch.cuda.synchronize()
            smp.start()
            for _ in range(4):
                for s in range(lo, hi):
                    tok = enc(self.rgbs[s])
            torch.cuda.synchronize()
            watts = smp.stop()
        e0.record()
        for s in range(lo, hi):
            tok = enc(self.rgbs[s])
        e1.record()
        torch.cuda.synchronize()
        c0 = self.eng.counters()
        self.reset_stats()
        for s in range(lo, hi):
            self.eng.ingest(self.depths[s], self.rgbs[s], tok, self.Ts[s * self.batch:(s + 1) * self.batch])
            self.eng.sync()                     # nothing beside the call: not even the rgb chain of the call before it
        e2.record()
        torch.cuda.synchronize()
        k = hi - lo
        c1 = self.eng.counters()
        return dict(encoder_ms=e0.elapsed_time(e1) / k, encoder_watts=watts, ingest_wall_ms=e1.elapsed_time(e2) / k, stages=self.stage_ms(),
                    U=(c1["voxel_rmw"] - c0["voxel_rmw"]) / k, U_new=(c1["max_id"] - c0["max_id"]) / k,
                    P=(c1["points_passed"] - c0["points_passed"]) / k, pairs=(c1["pairs"] - c0["pairs"]) / k)

    def close(self):
        self.eng.close()
        self.rgbs, self.depths, self.encs = [], [], []
        torch.cuda.empty_cache()


class PowerSampler:
    """Socket power (W) from `rocm-smi --showpower --csv`, sampled by a side thread; stop() -> {"mean_W", "max_W", "samples"} or None."""

    def __init__(self, period=0.05):
        import threading
        self.period, self.vals, self._stop = period, [], False
        self.th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import subprocess
        while not self._stop:
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout.strip().splitlines()
                self.vals.append(float(o[-1].split(",")[-1]))
            except Exception:
                pass
            time.sleep(self.period)

    def start(self):
        self.th.start()

    def stop(self):
        self._stop = True
        self.th.join(timeout=10)
        v = self.vals[1:-1] if len(self.vals) > 4 else self.vals      # the first / last sample may straddle the loaded interval
        return {"mean_W": sum(v) / len(v), "max_W": max(v), "samples": len(v)} if v else None


def stage_rooflines(p, iso, tok_bytes):
    """Every stage with its own algorithmic bytes (what it must read / write once) and measured HIP-event time."""
    F, N, g, D = p.batch, p.N, p.g, p.D
    U, U_new, pairs, P = iso["U"], iso["U_new"], iso["pairs"], F * N
    st = iso["stages"]
    byts = {
        "k_points": 8.0 * P,                                                   # depth f32 + RGBA u8 read once
        "k_keys_pairs": 4.0 * P + 12.0 * pairs,                                # cells in, pairs out
        "pair_sort": 2 * 12.0 * pairs,                                         # pairs in / out once
        "k_dense_reduce": (2 * U - U_new) * D * 4 + 8 * U + F * g * g * D * tok_bytes + 12 * pairs,
        "ids+point_order": 0.25 * P,                                           # per-voxel point order out: a start bit per point + a checkpoint per 64 (runs in: not counted)
        "k_chain": 8.0 * P + 19.0 * U,                                         # 8-byte record per point (round 6; 12 before), voxel state
    }
    out = {}
    for name, b in byts.items():
        ms = st.get(name, 0.0)
        if ms > 0:
            out[name] = {"ms_per_call": ms, "bytes_per_call": b, "GBs": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS}
    return out


def timed_repeats(q, w, n, repeats):
    """the K = n - w steps after w warm-up steps on a reset map, `repeats` times: seconds per repeat"""
    times = []
    for _ in range(repeats):
        q.eng.reset()
        q.run(0, w)
        q.eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        q.run(w, n)
        q.eng.sync(); torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return times


def side_precision_leg(a, p, rank, local_rank, repeats=3, hf=None):
    """Beside an f32 headline: the opt-in fast mode on the same frames — bf16 weights / activations / tokens, the encoder's GEMMs
    through hipBLASLt (PyTorch-ROCm, TunableOp choices committed), everything between them in-tree — timed like `value` (the K
    steps after the warm-up, `repeats` times on a reset map, median), and what it costs in accuracy: the stored per-voxel feature
    means of one batch against the f32 pipeline's (same frames, same voxels).  Also one short pass of the f32 pipeline on
    PyTorch-ROCm's own f32 GEMMs + SDPA (what a straight PyTorch port of the reference's encoder call runs at).
    Beside a bf16 headline (--precision bf16): the f32 in-tree pipeline, the same way."""
    other = "bf16" if p.precision == "f32" else "f32"
    n = p.n_steps
    w = min(a.warmup, n - 1)
    q = Pipeline(a, p.kind, a.arch, a.grid, p.batch, n, rank, local_rank, precision=other, share=p)
    try:
        times = timed_repeats(q, w, n, repeats)
        dt = statistics.median(times)
        iso = q.isolated(w, min(n, w + 4))
        from_host = hf.leg(q, a, repeats) if hf is not None else None
        q.eng.reset()
        q.run(0, 1)
        acc_o, cnt_o = q.eng.export_dense()
        fl = q.vit.flops_per_frame() * q.batch
        enc_ms = iso["encoder_ms"]
    finally:
        q.close()
        del q
        torch.cuda.empty_cache()
    p.eng.reset()
    p.run(0, 1)
    acc_p, cnt_p = p.eng.export_dense()
    assert np.array_equal(cnt_o, cnt_p)
    c = np.maximum(cnt_p, 1)[:, None].astype(np.float32)
    mo, mp_ = (acc_o / c, acc_p / c) if a.mode == "mean" else (acc_o, acc_p)
    m32 = mp_ if p.precision == "f32" else mo
    d = np.abs(mo - mp_)
    key = "value_bf16_encoder_library_gemms" if other == "bf16" else "value_f32_encoder_in_tree"
    out = {key: (n - w) * p.batch / dt, "side_steps": n - w, "side_repeats": repeats, "side_seconds_per_repeat": times,
           "side_encoder_ms_per_step": enc_ms, "side_ingest_ms_per_step": iso["stages"]["bsc_ingest"],
           "side_encoder_tflops": fl / (enc_ms * 1e-3) / 1e12,
           "tokens_bf16_max_abs_err": float(d.max()), "tokens_bf16_mean_abs_err": float(d.mean()),
           "tokens_rms": float(np.sqrt((m32.astype(np.float64) ** 2).mean()))}
    if from_host is not None:
        out["side_from_host"] = from_host
    if p.precision == "f32" and p.vit.split_gemm:
        # the f32 pipeline on PyTorch-ROCm's own f32 GEMMs / SDPA, one short eager pass, library-default solutions
        tuning_was_on = None
        try:
            tuning_was_on = torch.cuda.tunable.tuning_is_enabled()
            torch.cuda.tunable.tuning_enable(False)
        except Exception:
            pass
        p.vit.split_gemm = False
        try:
            def run(lo, hi):
                for s_ in range(lo, hi):
                    p.eng.ingest(p.depths[s_], p.rgbs[s_], p.vit.patch_tokens(p.rgbs[s_]), p.Ts[s_ * p.batch:(s_ + 1) * p.batch])
            p.eng.reset()
            run(0, 1)
            p.eng.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(1, min(n, 3))
            p.eng.sync(); torch.cuda.synchronize()
            out["value_f32_encoder_pytorch_gemms"] = (min(n, 3) - 1) * p.batch / (time.perf_counter() - t0)
        finally:
            p.vit.split_gemm = True
            try:
                if tuning_was_on:
                    torch.cuda.tunable.tuning_enable(True)
            except Exception:
                pass
    out["side_note"] = (f"{key}: {n - w} steps x {p.batch} frames after {w} warm-up steps, median of {repeats} repeats, same frames as "
                        "`value`; bf16 weights / activations / tokens, GEMMs = hipBLASLt via PyTorch (NOT in-tree), the rest in-tree.  "
                        "tokens_bf16_*: per-voxel feature means of one batch, bf16 pipeline against f32 pipeline — the north star's 1e-3 "
                        "bound is met by the f32 pipeline `value` times (tokens within 1e-5 of PyTorch f32; f32 accumulation), not by "
                        "the bf16 mode.  value_f32_encoder_pytorch_gemms: the f32 pipeline with torch f32 GEMMs + SDPA, 2 steps, one pass")
    return out


class HostFrames:
    """Frames that start on the HOST, as the reference's simulator hands them over (memory_2.py:1090-1096, env.py:166-235):
    `cyc` steps' frames in pinned host memory, two device-side step buffers, hipMemcpyAsync of step s + 1 on a copy stream under
    the encoder / ingest of step s.  The same cyclic schedule can be run from the resident frames (the control)."""

    def __init__(self, p, cyc=4):
        self.cyc = cyc = min(cyc, p.n_steps)
        t0 = time.perf_counter()
        self.rgb_h = [torch.empty(p.rgbs[i].shape, dtype=torch.uint8, pin_memory=True) for i in range(cyc)]
        self.dep_h = [torch.empty(p.depths[i].shape, dtype=torch.float32, pin_memory=True) for i in range(cyc)]
        for i in range(cyc):
            self.rgb_h[i].copy_(p.rgbs[i])
            self.dep_h[i].copy_(p.depths[i])
        torch.cuda.synchronize()
        self.setup_s = time.perf_counter() - t0
        self.rgb_d = [torch.empty_like(p.rgbs[0]) for _ in range(2)]
        self.dep_d = [torch.empty_like(p.depths[0]) for _ in range(2)]
        self.copy = torch.cuda.Stream()
        self.bytes_per_step = self.rgb_h[0].numel() + 4 * self.dep_h[0].numel()
        # the copy alone: achieved H2D rate
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.cuda.stream(self.copy):
            self.rgb_d[0].copy_(self.rgb_h[0], non_blocking=True)
            ev[0].record()
            for i in range(2):
                self.rgb_d[i].copy_(self.rgb_h[i % cyc], non_blocking=True)
                self.dep_d[i].copy_(self.dep_h[i % cyc], non_blocking=True)
            ev[1].record()
        torch.cuda.synchronize()
        self.h2d_GBs_alone = 2 * self.bytes_per_step / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9

    def run(self, q, lo, hi, from_host, ahead=False, primed=False):
        """steps [lo, hi).  ahead: also start the copy of step `hi` (the caller runs it next: a continuous stream of steps —
        the warm-up hands the timed region its first frames the way every later step gets them); primed: step `lo` was
        started that way."""
        cyc, B = self.cyc, q.batch
        main = torch.cuda.current_stream()
        if not from_host:
            for s in range(lo, hi):
                c = s % cyc
                tok = q.enc(q.rgbs[c])
                q.eng.ingest(q.depths[c], q.rgbs[c], tok, q.Ts[c * B:(c + 1) * B])
            return
        ch = int(os.environ.get("BSC_H2D_CHUNK_FRAMES", "64"))      # frames per hipMemcpyAsync
        if not primed:
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
            self.free[0].record(main); self.free[1].record(main)
        ready, free = self.ready, self.free

        def feed(s):
            b, c = s & 1, s % cyc
            with torch.cuda.stream(self.copy):
                self.copy.wait_event(free[b])             # the step that last read this buffer has passed its ingest
                for f0 in range(0, B, ch):
                    self.rgb_d[b][f0:f0 + ch].copy_(self.rgb_h[c][f0:f0 + ch], non_blocking=True)
                    self.dep_d[b][f0:f0 + ch].copy_(self.dep_h[c][f0:f0 + ch], non_blocking=True)
                ready[b].record(self.copy)
        if not primed:
            feed(lo)
        for s in range(lo, hi):
            if s + 1 < hi or ahead:
                feed(s + 1)
            b, c = s & 1, s % cyc
            main.wait_event(ready[b])
            tok = q.enc(self.rgb_d[b])
            q.eng.ingest(self.dep_d[b], self.rgb_d[b], tok, q.Ts[c * B:(c + 1) * B])
            free[b].record(main)

    def timed(self, q, w, n, repeats, from_host):
        times = []
        for _ in range(repeats):
            q.eng.reset()
            self.run(q, 0, w, from_host, ahead=True)
            q.eng.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.run(q, w, n, from_host, primed=True)
            q.eng.sync(); torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return statistics.median(times)

    def leg(self, q, a, repeats=3):
        n = q.n_steps
        w = min(a.warmup, n - 1)
        t_res = self.timed(q, w, n, repeats, False)
        t_host = self.timed(q, w, n, repeats, True)
        fps = (n - w) * q.batch / t_host
        return {"value_from_host": fps, "value_resident_same_schedule": (n - w) * q.batch / t_res,
                "from_host_over_resident": t_res / t_host, "h2d_GBs_needed": fps * self.bytes_per_step / q.batch / 1e9,
                "h2d_GBs_copy_alone": self.h2d_GBs_alone, "bytes_per_frame": self.bytes_per_step // q.batch,
                "precision": q.precision}

    def close(self):
        self.rgb_h = self.dep_h = self.rgb_d = self.dep_d = None
        torch.cuda.empty_cache()


def encoder_kernel_rates(vit, batch, reps=3):
    """HIP-event times of the f32 encoder's kernels at the bench's shapes, each alone (M = batch x tokens rows), in the forms the
    forward runs: qkv and fc1 with the LayerNorm folded into their operand load, proj and fc2 with the residual epilogue that
    leaves the row statistics, attention on pieces.  TF = 16-bit MFMA work: 3 piece products per f32 product.  (Alone and back to
    back a kernel runs hotter than inside the forward, whose sustained rate is set by the chip's power limit: the sum of these
    is not the forward's time — profiles/r05_power_probe.txt.)"""
    from bsc_nav_amd import encoder as E
    T = 1 + vit.registers + vit.grid * vit.grid
    M, Wd = batch * T, vit.width
    blk = vit.blocks[0]
    heads = blk.heads
    mlp = blk.fc1.weight.shape[0]
    dev = blk.fc1.weight.device
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((batch * vit.grid * vit.grid, Wd), device=dev, generator=gen)
    u, _, (stats, mu) = E.embed_tokens_f32(vit, x, batch, ln=None, stats=True)
    SL = E.SplitLinear
    qkv_l, fc1_l = vit._split(blk.qkv, blk.ln1), vit._split(blk.fc1, blk.ln2)
    qkv = qkv_l(u, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=1.0)
    att = E.attention_split(qkv, batch, T, heads, out_scale=16.0)
    h = fc1_l(u, SL.GELU, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=4.0)
    jobs = {
        "qkv_layernorm_in_load": (lambda: qkv_l(u, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=1.0, out=qkv), 2.0 * M * Wd * 3 * Wd),
        "attention": (lambda: E.attention_split(qkv, batch, T, heads, out_scale=16.0), 4.0 * batch * heads * T * T * 64),
        "proj_resid_stats": (lambda: vit._split(blk.proj)(att, SL.RESID, resid=u, out=u, a_scale=16.0, a_pieces=True, ln_stats=stats, ln_mu=mu),
                             2.0 * M * Wd * Wd),
        "fc1_gelu_layernorm_in_load": (lambda: fc1_l(u, SL.GELU, a_ln=True, ln_stats=stats, ln_mu=mu, c_pieces_scale=4.0, out=h), 2.0 * M * Wd * mlp),
        "fc2_resid_stats": (lambda: vit._split(blk.fc2)(h, SL.RESID, resid=u, out=u, a_scale=4.0, a_pieces=True, ln_stats=stats, ln_mu=mu),
                            2.0 * M * Wd * mlp),
    }
    out = {}
    for name, (fn, flops) in jobs.items():
        fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        out[name] = {"us": 1e3 * ms, "fp16_mfma_TFLOPs": 3.0 * flops / (ms * 1e-3) / 1e12,
                     "frac_of_16bit_mfma_peak": 3.0 * flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF}
    return out


def small_batch_leg(archs=("vit_b16", "vit_l14"), batches=(1, 8)):
    """The reference's ONLINE use of the encoder: one DINOv2 forward per simulator step (memory_2.py:732-742, a frame per call).
    Latency of the f32 forward at 1 and 8 frames per call: the in-tree kernels (few-rows tiles of the split GEMM: 32 x 128 / 128 x
    128 with split-K, LayerNorm as a pass) eager and HIP-graphed, against PyTorch-ROCm's f32 GEMMs + SDPA on the same module."""
    from bsc_nav_amd import encoder
    out = {}
    for arch in archs:
        vit = encoder.RandomViT(arch, image_size=224, seed=0, dtype=torch.float32).cuda()
        for B in batches:
            rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")

            def timed(fn, n=20):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            e = {"in_tree_eager_ms": timed(lambda: vit.patch_tokens(rgb))}
            g = encoder.GraphedEncoder(vit, B, 480, 640, 4, False)
            e["in_tree_graph_ms"] = timed(lambda: g(rgb))
            vit.split_gemm = False
            try:
                e["pytorch_f32_gemms_ms"] = timed(lambda: vit.patch_tokens(rgb), 10)
            finally:
                vit.split_gemm = True
            out[f"{arch}_B{B}"] = e
            del g
        del vit
        torch.cuda.empty_cache()
    return out


def exact_mode_leg(a, local_rank, frames=192):
    """The reference-semantics mode (the only one whose every output is pinned to the reference's goldens): token cache of
    50 000 rows flushed into <= 10 raw tokens per voxel with random.choice replacement, host-shuffled sub-sampling on NumPy's
    global stream (memory_2.py:747-749), host alpha.  640x480 frames, 16x16x1024 tokens from a stand-in provider (the encoder is
    not part of this leg), 256^3 grid, depth_sample_rate 1000 (the reference's default, args.py) and 50: frame by frame through
    VoxelTokenMemory.obs2voxeltoken — host frames in, as the reference's loop hands them over — with and without the sampling
    drawn one frame ahead on a host thread (prefetched_sampling), and batched through ingest_frames (device frames)."""
    import random
    import tempfile
    import types
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic, geometry
    H, W, g, D, gs = a.height, a.width, 16, 1024, 256
    poses = synthetic.random_walk_poses(3, frames)
    rgb, depth, _ = synthetic.make_frames(3, frames, H, W, "room", poses=poses)
    rgb_h, depth_h = rgb.cpu().numpy(), depth.cpu().numpy()
    tok = torch.randn(1, g * g, D, device="cuda")
    dino = types.SimpleNamespace(forward_features=lambda x: {"x_norm_patchtokens": tok})
    out = {"frames": frames, "tokens": f"{g}x{g}x{D}", "grid": gs}
    t0 = time.perf_counter()
    n_sh = 16
    for _ in range(n_sh):
        geometry.sample_indices_fast(H * W, 1000)
    out["host_shuffle_ms_per_frame"] = (time.perf_counter() - t0) / n_sh * 1e3
    for rate in (1000, 50):
        res = {}
        for name in ("obs2voxeltoken", "obs2voxeltoken_prefetched_sampling", "ingest_frames_batch32"):
            tmp = tempfile.mkdtemp(prefix="bsc_exact_")
            args = B.MemoryArgs(width=W, height=H, grid_size=gs, cell_size=0.1, floor_height=-12.8, map_height=12.8,
                                depth_sample_rate=rate, query_width=224, query_height=224, memory_path=tmp, scene_name="bench",
                                token_dim=D)
            mem = B.VoxelTokenMemory(args, preload_dino=dino, need_diffusion=False, feature_mode="exact", voxel_capacity=400_000,
                                     token_capacity=4_000_000, max_frames_per_call=32)
            np.random.seed(0); random.seed(0)
            w = 8
            torch.cuda.synchronize()
            if name.startswith("obs2voxeltoken"):
                def loop(lo, hi):
                    for f in range(lo, hi):
                        mem.obs2voxeltoken({"rgb": rgb_h[f], "depth": depth_h[f]}, poses[f])
                if name.endswith("prefetched_sampling"):
                    with mem.prefetched_sampling(H * W):
                        loop(0, w)
                        mem.engine.sync(); torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        loop(w, frames)
                        mem.engine.sync(); torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                else:
                    loop(0, w)
                    mem.engine.sync(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    loop(w, frames)
                    mem.engine.sync(); torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                n = frames - w
            else:
                toks = tok.view(1, g, g, D).expand(32, g, g, D).contiguous()
                with mem.prefetched_sampling(H * W, depth=40):
                    mem.ingest_frames(rgb[:32], depth[:32], poses[:32], tokens=toks)
                    mem.engine.sync(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for lo in range(32, frames - 31, 32):
                        mem.ingest_frames(rgb[lo:lo + 32], depth[lo:lo + 32], poses[lo:lo + 32], tokens=toks)
                    mem.engine.sync(); torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                n = (frames - 32) // 32 * 32
            k = mem.engine.counters()
            res[name] = {"frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3, "voxels": k["max_id"], "store_tokens": k["store_tokens"],
                         "flushes": k["flushes"]}
            mem.engine.close()
        res["host_shuffle_share_frame_by_frame"] = out["host_shuffle_ms_per_frame"] / res["obs2voxeltoken"]["ms_per_frame"]
        out[f"depth_sample_rate_{rate}"] = res
    # the reference's whole online step at its own precision: obs2voxeltoken with the f32 ViT-L/14 + 4 registers forward inside
    # (the in-tree encoder's few-rows forms), depth_sample_rate 1000
    def online():
        from bsc_nav_amd import encoder
        vit = encoder.RandomViT("vit_l14", image_size=224, seed=0, dtype=torch.float32).cuda()
        tmp = tempfile.mkdtemp(prefix="bsc_exact_")
        args = B.MemoryArgs(width=W, height=H, grid_size=gs, cell_size=0.1, floor_height=-12.8, map_height=12.8, depth_sample_rate=1000,
                            query_width=224, query_height=224, memory_path=tmp, scene_name="bench", token_dim=D)
        mem = B.VoxelTokenMemory(args, preload_dino=vit, need_diffusion=False, feature_mode="exact", voxel_capacity=400_000,
                                 token_capacity=4_000_000)
        np.random.seed(0); random.seed(0)
        res = {}
        for name in ("obs2voxeltoken", "obs2voxeltoken_prefetched_sampling"):
            mem.engine.reset()
            n0, n1 = 8, min(frames, 104)

            def loop(lo, hi):
                for f in range(lo, hi):
                    mem.obs2voxeltoken({"rgb": rgb_h[f], "depth": depth_h[f]}, poses[f])
            ctx = mem.prefetched_sampling(H * W) if name.endswith("prefetched_sampling") else None
            if ctx is not None:
                ctx.__enter__()
            try:
                loop(0, n0)
                mem.engine.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                loop(n0, n1)
                mem.engine.sync(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                if ctx is not None:
                    ctx.__exit__(None, None, None)
            res[name] = {"frames_per_s": (n1 - n0) / dt, "ms_per_frame": dt / (n1 - n0) * 1e3}
        mem.engine.close()
        res["note"] = ("one frame per call, host frames in, the f32 ViT-L/14 + 4 registers forward (random weights) on the in-tree "
                       "kernels inside every call; the reference's own loop with DINOv2 on its GPU: ~8.5 frames/s (BASELINE.md)")
        return res
    try:
        out["online_step_f32_vit_l14_inside"] = online()
    except Exception as e:      # noqa: BLE001
        out["online_step_f32_vit_l14_inside"] = {"error": f"{type(e).__name__}: {e}"}
    out["note"] = ("frame by frame: host numpy frames -> H2D copies, host-side pose chain + NumPy-stream Fisher-Yates over all "
                   f"{H * W} pixels (bsc_host_shuffled_sample) + host alpha, then one bsc_ingest call per frame; "
                   "prefetched_sampling draws the shuffle of frame f+1 on a host thread under the work of frame f (the stream "
                   "stays sequential); the reference's own loop runs ~8.5 frames/s at rate 1000 (BASELINE.md)")
    return out


def localize_store_leg(B, a, local_rank, D=1024, V=1 << 20, gL=512):
    """BASELINE configs[3] in the REFERENCE'S store shape: V voxels with M ~ U{1..10} raw tokens each
    (sum M ~ 5.8 M rows x 1024-D = 23.6 GB) loaded through bsc_import_store; voxel_localized takes the max over a voxel's
    tokens (memory_2.py:642-663).  Q = 1 and 8, plain and with the region + floor filters; cosine scan priced against HBM."""
    try:
        free_kb = int([ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0].split()[1])
    except Exception:
        free_kb = 0
    if free_kb * 1024 < 3 * V * 5.5 * D * 4:
        V = 1 << 18                                  # small host: the store is staged through host memory like a loaded memory
    gen = torch.Generator(device="cuda").manual_seed(9)
    codes = torch.randperm(gL ** 3, device="cuda", generator=gen)[:V]
    keys = torch.stack([codes // (gL * gL), (codes // gL) % gL, codes % gL], dim=1).to(torch.int32).contiguous()
    cnt = torch.randint(1, 11, (V,), device="cuda", generator=gen, dtype=torch.int32)
    T = int(cnt.sum().item())
    rows = torch.empty((T, D), dtype=torch.float32, device="cuda")
    for lo in range(0, T, 1 << 20):
        rows[lo:lo + (1 << 20)] = torch.randn((min(1 << 20, T - lo), D), device="cuda", generator=gen)
    eng = B.VoxelEngine(a.height, a.width, gL, 0.1, -gL * 0.05, gL * 0.05, 16, D, mode="exact", iter_size=256,
                        voxel_capacity=V + 8, token_capacity=T, max_points=1024, device=local_rank)
    # the store as load_memory hands it over (memory_2.py:189-200): NumPy arrays in pageable host memory, ready before the clock starts
    kk, cnt_h, rows_h, dist_h = keys.cpu().numpy(), cnt.cpu().numpy(), rows.cpu().numpy(), np.zeros(T, np.float32)
    rgb_h, w_h = np.zeros((V, 3), np.uint8), np.ones(V, np.float32)
    t0 = time.perf_counter()
    eng.import_rgb(kk, rgb_h, w_h)
    eng.import_store(kk, cnt_h, rows_h, dist_h)
    load_s = time.perf_counter() - t0
    del rows_h
    out = {"voxels": V, "token_rows": T, "dim": D, "K": 100, "grid": gL, "tokens_per_voxel": "U{1..10}",
           "store_bytes": T * D * 4, "load_seconds_through_host": load_s, "load_GBs_from_pageable_host": T * D * 4 / load_s / 1e9,
           "load_note": "bsc_import_rgb + bsc_import_store of NumPy arrays (pageable memory): host threads stage 64 MB chunks into pinned "
                        "buffers under the DMA of the previous ones (csrc/capi.hip h2d_pipelined); until round 5 one hipMemcpy (4.6 GB/s) "
                        "and the timer also held the device-to-host copies that built the test arrays"}
    seg = torch.repeat_interleave(torch.arange(V, device="cuda"), cnt.to(torch.int64))
    for Q in (1, 8):
        q = torch.randn(Q, D, device="cuda", generator=gen)
        pos, sim, n = eng.localize(q, K=100)
        if Q == 1:      # correctness at this size: top-1 of an independent fp64 scan with the per-voxel max
            qn = q[0].double() / q[0].double().norm()
            best = torch.full((V,), -2.0, dtype=torch.float64, device="cuda")
            for lo in range(0, T, 1 << 19):
                r = rows[lo:lo + (1 << 19)].double()
                best.scatter_reduce_(0, seg[lo:lo + (1 << 19)], (r @ qn) / r.norm(dim=1), reduce="amax")
            assert pos[0, 0].tolist() == keys[best.argmax()].tolist(), "store-shape localize top-1 differs from the fp64 scan"
            del best
        for name, kw in (("", {}), ("_region_floor", dict(radius=150.0, curr=[gL // 2] * 3, floor=(gL // 6, gL - gL // 6)))):
            eng.kernel_stats(1, reset=True)
            lats = []
            for _ in range(10):
                torch.cuda.synchronize()
                t = time.perf_counter()
                eng.localize(q, K=100, **kw)
                torch.cuda.synchronize()
                lats.append(time.perf_counter() - t)
            ls = eng.kernel_stats(1)
            ms = ls["ms"] / max(1, ls["launches"])
            gbs = T * D * 4 / (ms * 1e-3) / 1e9
            out[f"q{Q}{name}"] = {"latency_ms": statistics.median(lats) * 1e3, "cosine_ms": ms, "cosine_GBs": gbs,
                                  "cosine_frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
    eng.close()
    del rows
    torch.cuda.empty_cache()
    return out


def localize_leg(B, a, local_rank, D, V=1 << 20, gL=512):
    g = 16
    engL = B.VoxelEngine(a.height, a.width, gL, 0.1, -gL * 0.05, gL * 0.05, g, D, mode="mean", voxel_capacity=V + 8,
                         max_points=1024, device=local_rank)
    gen = torch.Generator(device="cuda").manual_seed(5)
    codes = torch.randperm(gL ** 3, device="cuda", generator=gen)[:V]
    keys = torch.stack([codes // (gL * gL), (codes // gL) % gL, codes % gL], dim=1).to(torch.int32).contiguous()
    rows = torch.randn((V, D), device="cuda", generator=gen)
    engL.dense_replace(keys, rows, torch.ones(V, dtype=torch.int32, device="cuda"))
    loc = {"voxels": V, "dim": D, "K": 100, "grid": gL}
    for Q in (1, 8, 256):
        q = torch.randn(Q, D, device="cuda", generator=gen)
        pos, sim, cnt = engL.localize(q, K=100)
        if Q == 1:      # correctness at this size, inside the bench: the top-1 of an independent fp64 scan
            rn = rows.double() / rows.double().norm(dim=1, keepdim=True)
            ref = (rn @ (q[0].double() / q[0].double().norm())).argmax().item()
            assert pos[0, 0].tolist() == keys[ref].tolist(), "localize top-1 differs from the fp64 scan"
            del rn
        engL.kernel_stats(1, reset=True)
        lats = []
        for _ in range(10):
            torch.cuda.synchronize()
            t = time.perf_counter()
            engL.localize(q, K=100)
            torch.cuda.synchronize()
            lats.append(time.perf_counter() - t)
        ls = engL.kernel_stats(1)
        ms = ls["ms"] / max(1, ls["launches"])
        e = {"latency_ms": statistics.median(lats) * 1e3, "cosine_ms": ms}
        if Q > 64:      # fp16 matrix cores (k_cosine_f16x2): three fp16 piece products per f32 product, per-row scales cached
            tf = 2.0 * V * D * Q / (ms * 1e-3) / 1e12
            np_ = 6.0 if os.environ.get("BSC_COSINE_BF16") else 3.0
            e.update({"cosine_f32_equivalent_TFLOPs": tf, "cosine_16bit_mfma_TFLOPs": np_ * tf, "piece_products": np_,
                      "cosine_frac_of_16bit_mfma_peak": np_ * tf / MFMA_BF16_PEAK_TF})
            # the first query batch after the rows were replaced: since round 6 whoever changes the rows (ingest: in its reduce;
            # imports / merges: at their end, bsc localize_prepare) leaves name ranks and row scales current — this should equal latency_ms
            engL.dense_replace(keys, rows, torch.ones(V, dtype=torch.int32, device="cuda"))
            engL.sync(); torch.cuda.synchronize()
            t = time.perf_counter()
            engL.localize(q, K=100)
            torch.cuda.synchronize()
            e["latency_ms_first_call_after_the_map_changed"] = (time.perf_counter() - t) * 1e3
        elif Q >= 16:   # fp32-MFMA GEMM path: priced against the 157.3 TFLOP/s fp32 matrix peak
            tf = 2.0 * V * D * Q / (ms * 1e-3) / 1e12
            e.update({"cosine_TFLOPs": tf, "cosine_frac_of_f32_mfma_peak": tf / MFMA_F32_PEAK_TF})
        else:
            gbs = ls["bytes"] / max(1, ls["launches"]) / (ms * 1e-3) / 1e9
            e.update({"cosine_GBs": gbs, "cosine_frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
        loc[f"q{Q}"] = e
    # a larger K through the same sample + filter selection (the sample grows with K so that the survivor lists do not overflow)
    q = torch.randn(8, D, device="cuda", generator=gen)
    engL.localize(q, K=512)
    lats = []
    for _ in range(10):
        torch.cuda.synchronize()
        t = time.perf_counter()
        engL.localize(q, K=512)
        torch.cuda.synchronize()
        lats.append(time.perf_counter() - t)
    loc["q8_K512"] = {"latency_ms": statistics.median(lats) * 1e3}
    # A/B: the round-3/4 scan on bf16 pieces (three pieces, six products)
    os.environ["BSC_COSINE_BF16"] = "1"
    try:
        q = torch.randn(256, D, device="cuda", generator=gen)
        engL.localize(q, K=100)
        engL.kernel_stats(1, reset=True)
        lats = []
        for _ in range(10):
            torch.cuda.synchronize()
            t = time.perf_counter()
            engL.localize(q, K=100)
            torch.cuda.synchronize()
            lats.append(time.perf_counter() - t)
        ls = engL.kernel_stats(1)
        loc["q256_bf16_six_products"] = {"latency_ms": statistics.median(lats) * 1e3, "cosine_ms": ls["ms"] / max(1, ls["launches"])}
    finally:
        del os.environ["BSC_COSINE_BF16"]
    engL.close()
    del rows
    torch.cuda.empty_cache()
    return loc


# ---- CPU baseline (the only place bench.py touches oracle/) ------------------------------------------------------
_CPU = {}


def _cpu_worker(job):
    """One host process: its frame shard through the plain-C oracle into a private map -> (keys, acc, cnt)."""
    from oracle import oracle as orc
    lo, hi, vcap = job
    H, W, gs, half, g, D, mode, Ts, dep, rgb, tok = _CPU["args"]
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.1, -half, half, g, D, mode=mode), voxel_capacity=vcap)
    t = time.perf_counter()
    for f in range(lo, hi):
        om.ingest_frame(dep[f], rgb[f], None, Ts[f], tok[f])
    dt = time.perf_counter() - t
    pos = om.export_rgb()[0]
    acc, cnt = om.export_dense()
    return dt, pos, acc, cnt


def cpu_baseline(a, p, seconds):
    """1 core: first frames of the workload, sequential.  All cores: frame-sharded private maps (one process per hardware
    thread) + a NumPy merge by voxel key — the CPU counterpart of the multi-GPU path."""
    import multiprocessing as mp
    from oracle import oracle as orc
    H, W, g, D = p.H, p.W, p.g, p.D
    gs = p.eng.cfg.grid_size
    half = gs * 0.05
    mode = 1 if a.mode == "mean" else 2
    nb = min(p.n_steps, 4)
    tok = np.concatenate([p.vit.patch_tokens(p.rgbs[s]).float().cpu().numpy() for s in range(nb)])
    rgb = np.concatenate([p.rgbs[s].cpu().numpy() for s in range(nb)])
    dep = np.concatenate([p.depths[s].cpu().numpy() for s in range(nb)])
    Ts = p.Ts[:len(dep)]
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.1, -half, half, g, D, mode=mode), voxel_capacity=2_000_000)
    nf, cdt = 0, 0.0
    while nf < len(dep) and (cdt < seconds or nf < 2):
        t = time.perf_counter()
        om.ingest_frame(dep[nf], rgb[nf], None, Ts[nf], tok[nf])
        cdt += time.perf_counter() - t
        nf += 1
    del om
    # the reference's own style of execution: a Python loop over the points with small NumPy calls (oracle/numpy_loop.py,
    # pinned to the reference's goldens) — a few seconds of it, scaled to the frame's point count
    from oracle.numpy_loop import NumpyLoopMemory
    nl = NumpyLoopMemory(H, W, gs, 0.1, -half, half, g, D, iter_size=50000)
    t = time.perf_counter()
    npts = nl.ingest_frame(dep[0], rgb[0], None, Ts[0], tok[0], max_points=20000)
    ldt = time.perf_counter() - t
    numpy_loop = {"value": (npts / ldt) / (H * W), "unit": "frames/s", "cores": 1, "kind": "port",
                  "sample": f"first {npts} points of one frame through oracle/numpy_loop.py (per-point Python loop as in "
                            f"memory_2.py:863-903, {1e3 * ldt / npts:.3f} ms per point), scaled to {H * W} points per frame"}
    del nl
    one = {"value": nf / cdt, "unit": "frames/s", "cores": 1, "kind": "port", "numpy_loop": numpy_loop,
           "sample": f"first {nf} frames of the same workload through oracle/bsc_oracle.c (memory path only: geometry + "
                     f"voxel scatter, encoder excluded), {cdt:.1f} s on 1 of {os.cpu_count()} host cores"}
    # all cores
    ncpu = os.cpu_count() or 1
    per = max(1, min(len(dep) // ncpu, int(one["value"] * seconds)))         # frames per worker: ~`seconds` of work each
    W_ = min(ncpu, len(dep) // per)
    vcap = 400_000
    try:
        free_kb = int([ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0].split()[1])
        W_ = max(1, min(W_, int(free_kb * 1024 * 0.5 / (vcap * D * 4 + gs * gs * p.eng.nh * 4))))
    except Exception:
        pass
    _CPU["args"] = (H, W, gs, half, g, D, mode, Ts, dep, rgb, tok)
    jobs = [(w * per, (w + 1) * per, vcap) for w in range(W_)]
    ctx = mp.get_context("fork")                       # children share the staged frames copy-on-write; they never touch HIP
    with ctx.Pool(W_) as pool:
        pool.map(abs, range(W_))                       # workers up before the clock starts
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, jobs, chunksize=1)
        t_ingest = time.perf_counter() - t0
    t1 = time.perf_counter()
    keys = np.concatenate([r[1] for r in res]).astype(np.int64)
    code = (keys[:, 0] << 42) | (keys[:, 1] << 21) | keys[:, 2]
    order = np.argsort(code, kind="stable")
    starts = np.flatnonzero(np.concatenate([[True], code[order][1:] != code[order][:-1]]))
    rows = np.concatenate([r[2] for r in res])[order]
    acc = np.add.reduceat(rows, starts, axis=0) if mode == 1 else np.maximum.reduceat(rows, starts, axis=0)
    cnt = np.add.reduceat(np.concatenate([r[3] for r in res]).astype(np.int64)[order], starts)
    uniq = starts
    assert acc.shape[0] == len(starts) == len(cnt)
    t_merge = time.perf_counter() - t1
    _CPU.clear()
    allc = {"value": W_ * per / (t_ingest + t_merge), "unit": "frames/s", "cores": W_, "kind": "port",
            "sample": f"{W_} processes x {per} frames of the same workload, private maps + NumPy merge by voxel key "
                      f"({t_ingest:.1f} s ingest incl. map allocation, {t_merge:.1f} s merge of {len(uniq)} voxels); "
                      f"{ncpu} hardware threads on the box"}
    return one, allc


def launch_ranks(a, backend):
    """`python bench.py --gpus N` outside torchrun: run the N ranks of this script under torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1) and return their exit status.  Fewer than N visible GPUs is an error, not a
    silently smaller run."""
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and n_dev < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {n_dev} GPU(s) visible on this node — one rank per GPU is required "
                         f"(RCCL); nothing was measured")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    a = parse()
    backend = os.environ.get("BSC_BENCH_BACKEND", "nccl")      # "gloo" only to exercise the N>1 path on a 1-GPU box
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a, backend))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    # under a launcher (torchrun sets WORLD_SIZE) the process group exists at ANY world size — `--gpus 1` then times the same code
    # path as N > 1 (collectives of one rank: RCCL runs them, they move nothing) and must reproduce the plain N = 1 line
    dist_active = world > 1 or "WORLD_SIZE" in os.environ
    dist_info = None
    if dist_active:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # evidence that the ranks are really connected: a SUM all-reduce of ones through the backend the merge will use
        ones = torch.ones(1, dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        dist_info = {"backend": dist.get_backend(), "is_rccl": dist.get_backend() == "nccl", "ranks_counted_by_all_reduce": int(ones.item()),
                     "world_size": world, "gpus_visible": torch.cuda.device_count()}
    import bsc_nav_amd as B
    from bsc_nav_amd import dist as bdist

    torch.cuda.set_stream(torch.cuda.Stream())          # a real stream: the encoder graph cannot be captured on the null stream
    n_steps = a.steps + a.warmup
    p = Pipeline(a, a.kind, a.arch, a.grid, a.batch, n_steps, rank, local_rank)
    g, D, N = p.g, p.D, p.N
    tok_bytes = p.tok_bytes
    f32 = p.precision == "f32"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    bdist.warmup_collectives(torch.device("cuda", local_rank))
    p.run(0, a.warmup)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, max over ranks; repeated, median ----
    times, merge_info, stage_timed, c0, c1 = [], None, None, None, None
    for rep in range(max(1, a.repeats)):
        if rep > 0:                       # identical work in every repeat: empty map, same frames
            p.eng.reset()
            p.run(0, a.warmup)
        barrier()
        c0 = p.eng.counters()
        p.reset_stats()
        t0 = time.perf_counter()
        p.run(a.warmup, n_steps)
        if dist_active:
            p.eng.sync(); torch.cuda.synchronize()
            tm = time.perf_counter()
            merge_info = bdist.merge_dense_maps(p.eng)
            torch.cuda.synchronize()
            merge_info["seconds_inside_timed_region"] = time.perf_counter() - tm
        p.eng.sync()                      # incl. the rgb chain of the last step, which the library launches lazily
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        times.append(dt)
        if rank == 0:
            stage_timed = p.stage_ms()
            c1 = p.eng.counters()
    dt = statistics.median(times)

    out = None
    if rank == 0:
        frames = a.steps * a.batch * world
        out = {
            "metric": "RGB-D frames/sec into voxel feature memory", "value": frames / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 results via fp16 split-operand MFMA (3 piece products per f32 product, f32 accumulate) in the encoder, f32 "
                      "tokens; f64 geometry, u8/f32 rgb chain, f32 feature accumulation" if f32 else
                      f"bf16 encoder (MFMA, f32 accumulate, library GEMMs; the reference's DINOv2 runs f32) -> "
                      f"{'bf16' if p.tokens_bf16 else 'f32'} tokens; f64 geometry, u8/f32 rgb chain, f32 feature accumulation"),
            "data": "synthetic",
            "repeats": len(times), "seconds_per_repeat": times, "timed_seconds_total": sum(times),
            "config": {"workload": f"{a.batch * a.steps} synthetic {p.W}x{p.H} RGB-D frames per GPU ({a.kind} depth, every "
                                   f"pixel), {a.arch} random {'f32' if f32 else 'bf16'} weights {D}-D tokens {g}x{g}, {a.grid}^3 grid of 0.1 m "
                                   f"cells, dense {a.mode} reduce" + ((", + RCCL reduce-scatter merge" if backend == "nccl" else
                                                                f", + {backend} reduce-scatter merge (RCCL stand-in on a box without {world} GPUs)")
                                                               if world > 1 else ""),
                       "frames_per_step": a.batch, "parallelism": f"frames sharded x{world}",
                       "encoder": ("in-tree: k_gemm_split (LayerNorm folded in) / k_attention_split (csrc/encoder_gemm.hip), HIP graph" if f32
                                   else "hipBLASLt GEMMs via PyTorch + in-tree attention / LayerNorm / preprocessing, HIP graph")},
        }
        if merge_info:
            out["config"]["merge"] = merge_info
        if dist_info:
            out["config"]["distributed"] = dist_info
            out["rccl_ranks"] = dist_info["ranks_counted_by_all_reduce"] if dist_info["is_rccl"] else 0
    def guarded(key, fn, into=None):
        """an optional leg: its result under `key`, or {"error": ...} — never the loss of the headline line"""
        tgt = out if into is None else into
        try:
            r = fn()
            if key is None:
                tgt.update(r)
            else:
                tgt[key] = r
        except Exception as e:           # noqa: BLE001
            import traceback
            tgt[key or "error"] = {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc(limit=2).splitlines()[-2].strip()}
            try:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            except Exception:
                pass

    if rank == 0:
        # the bsc_ingest call of THIS rank priced per SURVEY.md §8(d); at N > 1 the in-pipeline stage times of rank 0 (the
        # isolated wall figure needs the chip to itself and is measured at N = 1 only)
        U = (c1["voxel_rmw"] - c0["voxel_rmw"]) / a.steps           # voxel rows touched per call
        alg = ingest_alg_bytes(a.batch, N, g, D, tok_bytes, U)
        if world > 1:
            ing_ms = stage_timed["bsc_ingest"]
            out["roofline"] = {"bound": "hbm", "kernel": "bsc_ingest of rank 0 (main-stream HIP-event time inside the pipeline; per rank)",
                               "achieved": alg / ing_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / ing_ms / 1e6 / HBM_PEAK_GBS,
                               "traffic": None, "bytes_per_call": alg, "ms_per_call": ing_ms, "voxel_rows_per_call": U,
                               "stage_ms_in_pipeline": stage_timed}
            if merge_info and merge_info.get("seconds_inside_timed_region"):
                mb = merge_info.get("reduce_scatter_bytes_sent_per_rank", 0)          # rows each rank sends in the reduce-scatter
                merge_info["merge_GBs_per_rank"] = mb / merge_info["seconds_inside_timed_region"] / 1e9
                # xGMI is point-to-point: a rank's reduce-scatter traffic leaves over min(world - 1, 7) links of ~153 GB/s each
                links = max(1, min(world - 1, 7))
                merge_info["bytes_per_link_equivalent"] = mb / links
                merge_info["expected_reduce_scatter_ms_on_xgmi"] = mb / links / 153e9 * 1e3
                merge_info["xgmi_note"] = ("expected = bytes a rank sends / links / 153 GB/s (MI355X_MICROARCH.md: 7 links per GPU); the measured "
                                           "phases_ms are of the backend named in config.distributed")
    if rank == 0 and world == 1:
        iso = p.isolated(a.warmup, min(n_steps, a.warmup + 8), power=rank == 0 and world == 1)
        ing_ms = stage_timed["bsc_ingest"]
        single = {k: v for k, v in stage_timed.items() if k in ("k_points", "k_keys_pairs", "k_dense_reduce")}
        dom = max(single, key=single.get)
        traffic, traffic_commit = pmc_traffic(a.batch, tok_bytes)
        traffic_src = None if traffic is None else f"profiles/{PMC_FILE} @ {traffic_commit} (rocprofv3 --pmc passes of this command, committed; its source hash matches the tree)"
        traffic_live = False
        wall = iso["ingest_wall_ms"]
        out["roofline"] = {
            "bound": "hbm", "kernel": "bsc_ingest: one call followed by bsc_sync, alone on the chip — main-stream kernels, the per-voxel point "
                                     "order and the rgb chain on the library's side stream (SURVEY.md 8d bytes of the batch / that wall time)",
            "achieved": alg / wall / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / wall / 1e6 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_measured_in_this_run": traffic_live, "traffic_source": traffic_src,
            "bytes_per_call": alg, "ms_per_call": wall,
            "ms_per_call_main_stream_isolated": iso["stages"]["bsc_ingest"], "frac_main_stream_isolated": alg / iso["stages"]["bsc_ingest"] / 1e6 / HBM_PEAK_GBS,
            "ms_per_call_main_stream_in_pipeline": ing_ms, "frac_main_stream_in_pipeline": alg / ing_ms / 1e6 / HBM_PEAK_GBS,
            "voxel_rows_per_call": U, "points_per_call": (c1["points_passed"] - c0["points_passed"]) / a.steps,
            "pairs_per_call": (c1["pairs"] - c0["pairs"]) / a.steps, "U_over_P": U / max(1.0, a.batch * N),
            "dominant_kernel": dom, "dominant_kernel_ms": single[dom], "stage_ms_in_pipeline": stage_timed,
            "share_of_step": ing_ms / (1e3 * dt / a.steps),
        }
        # the dominant ingest kernel is bound by vector-instruction issue, not by HBM: instructions per 64 points from the SQ
        # counters of the committed PMC pass (profiles/), issue cycles = 4 per wave64 instruction, 8 for the f64 ones (half rate)
        kp_ms = iso["stages"]["k_points"]
        pts = iso["P"]
        cyc = (KP_VALU_PER_64 - KP_VALU_F64_PER_64) * 2 + KP_VALU_F64_PER_64 * 4
        need = pts / 64.0 * cyc
        out["roofline"]["k_points_valu"] = {
            "bound": "vector instruction issue (CDNA4 SIMD-32: a wave64 instruction issues over 2 cycles, an f64 one over 4) — the kernel's "
                     "largest single resource; by its phase clocks (profiles/README.md) the geometry phase runs at the issue rate of its five "
                     "co-resident wavefronts, the grouping phases wait on LDS round trips (half of its LDS cycles are bank conflicts) and on the "
                     "CU's one scalar unit (~100 scalar instructions per 64 points)",
            "kernel": "k_points", "ms_per_call": kp_ms,
            "valu_instructions_per_64_points": KP_VALU_PER_64, "of_them_f64": KP_VALU_F64_PER_64, "salu_instructions_per_64_points": KP_SALU_PER_64,
            "achieved": need / (kp_ms * 1e-3) / 1e12, "peak": N_SIMD * CLOCK_GHZ * 1e9 / 1e12, "unit": "T issue-cycles/s",
            "frac": need / (kp_ms * 1e-3) / (N_SIMD * CLOCK_GHZ * 1e9),
            "hbm_frac_of_own_bytes": 8.0 * pts / (kp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "source": "profiles/r06_pmc_sq_ingest.txt (SQ_INSTS_VALU / _F64 / SQ_INSTS_SALU per wavefront of 512 points) x points of this run / HIP-event time of this run"}
        out["roofline"]["kernels"] = {"note": "bsc_ingest running alone (no encoder beside it, a synchronize per call); own algorithmic bytes per stage",
                                      **stage_rooflines(p, iso, tok_bytes)}
        fl = p.vit.flops_per_frame() * a.batch
        enc_tf = fl / (iso["encoder_ms"] * 1e-3) / 1e12
        mf = 3.0 if f32 else 1.0                                 # 16-bit MFMA flops per flop of the forward
        enc_blk = {
            "bound": "mfma", "kernel": ("k_gemm_split (+ k_attention_split, k_layernorm_split): the whole f32-accuracy forward, fp16 pieces, "
                                        "three MFMA products per f32 product" if f32 else "hipBLASLt bf16 GEMMs (library) + in-tree attention / LayerNorm"),
            "achieved": mf * enc_tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s of 16-bit MFMA", "frac": mf * enc_tf / MFMA_BF16_PEAK_TF,
            "f32_equivalent_TFLOPs": enc_tf, "ms_per_step_alone": iso["encoder_ms"], "ms_per_step_in_pipeline": stage_timed.get("encoder"),
            "share_of_step": stage_timed.get("encoder", 0.0) / (1e3 * dt / a.steps),
            "sustained_mfma_note": "the forward runs at the chip's power limit: 1.36 kW socket power, engine clock 2.05-2.1 of 2.4 GHz while it "
                                   "runs (profiles/r05_power_probe.txt); a bare stream of 16-bit MFMAs on non-zero operands sustains ~1.0-1.25 "
                                   "PFLOP/s (0.40-0.50 of the data-sheet peak); kernel-level gains measured alone (LayerNorm passes removed, "
                                   "-5 % kernel time) do not show in the sustained forward",
            "mfma_busy_committed": MFMA_BUSY_COMMITTED}
        if iso.get("encoder_watts"):
            # the forward is power-bound: what it costs is joules — socket power while it runs alone (rocm-smi, sampled in this run) x
            # its time / frames of the step
            w = iso["encoder_watts"]
            enc_blk["socket_power_W_while_running_alone"] = w
            enc_blk["joules_per_frame"] = w["mean_W"] * iso["encoder_ms"] * 1e-3 / a.batch
            enc_blk["joules_per_forward"] = w["mean_W"] * iso["encoder_ms"] * 1e-3
        if f32 and p.vit.split_gemm:
            guarded("kernels", lambda: encoder_kernel_rates(p.vit, a.batch), into=enc_blk)
        out["roofline"]["encoder"] = enc_blk
        out["stages"] = {"note": "each stage alone: encoder; bsc_ingest main stream (HIP events); its rgb chain on the side stream; "
                                 "wall time of a call followed by bsc_sync (main stream, then the chain)",
                         "encoder_ms_per_step": iso["encoder_ms"], "ingest_ms_per_step": iso["stages"]["bsc_ingest"],
                         "ingest_plus_chain_wall_ms_per_step": iso["ingest_wall_ms"],
                         "chain_ms_per_step_side_stream": iso["stages"]["k_chain"],
                         "encoder_tflops": enc_tf, "encoder_16bit_mfma_tflops": mf * enc_tf,
                         "encoder_frac_of_16bit_mfma_peak": mf * enc_tf / MFMA_BF16_PEAK_TF, "voxels": c1["max_id"]}
        out["memory_path_frames_per_s"] = a.batch / (iso["ingest_wall_ms"] * 1e-3)     # bsc_ingest + its rgb chain alone, tokens as timed
        hf = None
        if not a.no_host_feed:
            def host_leg():
                nonlocal hf
                hf = HostFrames(p)
                r = hf.leg(p, a)
                r["note"] = (f"frames in pinned host memory ({hf.cyc} steps' frames cycled), double-buffered hipMemcpyAsync on a copy stream under "
                             "the previous step; value_resident_same_schedule = the same cyclic schedule from HBM-resident frames (the control)")
                return r
            guarded("from_host", host_leg)
            if isinstance(out.get("from_host"), dict) and "value_from_host" in out["from_host"]:
                out["value_from_host"] = out["from_host"]["value_from_host"]
        if not a.no_side:
            guarded(None, lambda: side_precision_leg(a, p, rank, local_rank, hf=hf))
        if hf is not None:
            hf.close()
            hf = None
        # ---- CPU baseline on the same frames (before they are freed) ----
        if not a.no_cpu_baseline:
            def cpu_leg():
                one, allc = cpu_baseline(a, p, a.cpu_seconds)
                return {"cpu_baseline": one, "cpu_baseline_all_cores": allc}
            guarded(None, cpu_leg)
            if "cpu_baseline" not in out:
                out["cpu_baseline"] = out.pop("error", {"error": "cpu baseline failed"})
        vit = p.vit
        p.close()
        # ---- the same pipeline on the other depth distributions, and BASELINE configs[2] per GPU ----
        if not a.no_workloads:
            out["workloads"] = {"room": {"frames_per_s": out["value"], "frames_per_step": a.batch, "voxels": out["stages"]["voxels"], "U_over_P": out["roofline"]["U_over_P"],
                                         "ingest_ms_per_step": out["roofline"]["ms_per_call_main_stream_in_pipeline"],
                                         "frac_of_hbm_bound": out["roofline"]["frac_main_stream_in_pipeline"]}}
            # as many steps as the headline where the map keeps growing over the run (a short run is mostly start-up: every
            # voxel new), half of them for the one-voxel-per-point stress case.  "cold": the timed steps follow two warm-up
            # steps on an empty map (new voxels all along the run); "warm": the same frames again over the map the cold pass
            # built (the steady state of a scene that is revisited) — separate keys, medians of `reps` passes each.
            def timed_pass(q, lo, hi):
                torch.cuda.synchronize()
                k0 = q.eng.counters()
                q.reset_stats()
                t0 = time.perf_counter()
                q.run(lo, hi)
                q.eng.sync()
                torch.cuda.synchronize()
                return time.perf_counter() - t0, k0, q.eng.counters(), q.stage_ms()

            def workload(q, steps, reps=3):
                cold, warm, last = [], [], None
                for r in range(reps):
                    if r:
                        q.eng.reset()
                    q.run(0, 2)
                    last = timed_pass(q, 2, steps + 2)
                    cold.append(last[0])
                    warm.append(timed_pass(q, 2, steps + 2)[0])
                dtk, k0, k1, st = last
                Uk = (k1["voxel_rmw"] - k0["voxel_rmw"]) / steps
                algk = ingest_alg_bytes(q.batch, q.N, q.g, q.D, q.tok_bytes, Uk)
                return {"frames_per_s": steps * q.batch / statistics.median(cold), "frames_per_s_cold": steps * q.batch / statistics.median(cold),
                        "frames_per_s_warm": steps * q.batch / statistics.median(warm), "steps": steps, "repeats": reps,
                        "voxels": k1["max_id"], "U_over_P": Uk / (q.batch * q.N), "pairs_per_call": (k1["pairs"] - k0["pairs"]) / steps,
                        "ingest_ms_per_step": st["bsc_ingest"], "bytes_per_call": algk,
                        "frac_of_hbm_bound": algk / st["bsc_ingest"] / 1e6 / HBM_PEAK_GBS, "stage_ms": st}

            def kind_leg(kind, steps):
                # the side workloads keep 384 frames per step (their voxel / pair capacities and the numbers of earlier rounds)
                q = Pipeline(a, kind, a.arch, a.grid, min(a.batch, 384), steps + 2, rank, local_rank, vit=vit)
                try:
                    return dict(workload(q, steps, reps=3 if kind != "iid" else 2), frames_per_step=q.batch)
                finally:
                    q.close()

            for kind, steps in (("hall", a.steps), ("iid", max(4, a.steps // 2)), ("room_off", a.steps)):
                guarded(kind, lambda: kind_leg(kind, steps), into=out["workloads"])

            # configs[2] (C3) per GPU: ViT-L/14 + 4 register tokens (the reference's dinov2_vitl14_reg, memory_2.py:43, args.py:50;
            # T = 261, 16x16x1024 tokens) into a 512^3 grid, as many steps as the headline: at the headline's precision, and the
            # other one beside it
            def c3_leg():
                a3 = argparse.Namespace(**vars(a))
                b3, s3 = 128, max(a.steps, 8)
                res = {}
                share = None
                for prec in (a.precision, "bf16" if a.precision == "f32" else "f32"):
                    q = Pipeline(a3, "hall", "vit_l14", 512, b3, s3 + 2, rank, local_rank, vcap=3_000_000, precision=prec, share=share)
                    try:
                        w3 = workload(q, s3, reps=3 if share is None else 2)
                        iso3 = q.isolated(2, min(s3 + 2, 10))
                        mf3 = 3.0 if prec == "f32" else 1.0
                        tf3 = q.vit.flops_per_frame() * b3 / (iso3["encoder_ms"] * 1e-3) / 1e12
                        r = {"frames_per_s_cold": w3["frames_per_s_cold"], "frames_per_s_warm": w3["frames_per_s_warm"],
                             "encoder_ms_per_step": iso3["encoder_ms"], "ingest_ms_per_step": iso3["stages"]["bsc_ingest"],
                             "memory_path_frames_per_s": b3 / (iso3["ingest_wall_ms"] * 1e-3), "encoder_tflops": tf3,
                             "encoder_16bit_mfma_tflops": mf3 * tf3, "encoder_frac_of_16bit_mfma_peak": mf3 * tf3 / MFMA_BF16_PEAK_TF,
                             "repeats": w3["repeats"], "voxels": w3["voxels"]}
                        if share is None:
                            res.update({"frames_per_s": r["frames_per_s_cold"], "precision": prec, "frames_per_step": b3, "steps": s3,
                                        "depth": "hall", "tokens": f"{q.g}x{q.g}x{q.D}", "sequence_length": 1 + q.vit.registers + q.g * q.g})
                            share = argparse.Namespace(Ts=q.Ts, rgbs=q.rgbs, depths=q.depths)
                        res["value_f32" if prec == "f32" else "value_bf16_library_gemms"] = r["frames_per_s_cold"]
                        res[prec] = r
                    finally:
                        q.close()
                return res
            out.setdefault("configs", {})
            guarded("C3_vit_l14_1024d_grid512_per_gpu", c3_leg, into=out["configs"])
        del vit
        torch.cuda.empty_cache()
        if not a.no_exact:
            guarded("exact_mode", lambda: exact_mode_leg(a, local_rank))
            guarded("encoder_f32_frame_by_frame", small_batch_leg)
        if not a.no_localize:
            # second half of the metric: localize top-K latency over a 2^20-voxel map (BASELINE configs[3]/[4] size)
            guarded("localize", lambda: localize_leg(B, a, local_rank, D))
            out.setdefault("configs", {})
            if D != 1024:
                guarded("C4_C5_localize_2pow20_x_1024_grid512", lambda: localize_leg(B, a, local_rank, 1024), into=out["configs"])
            guarded("C4_store_shape_2pow20_voxels_M_1to10_x_1024", lambda: localize_store_leg(B, a, local_rank), into=out["configs"])
    if rank == 0 and world == 1 and not a.no_pmc and "roofline" in out:
        # roofline.traffic measured in THIS run (after everything that is timed; the bench's own engines are idle meanwhile)
        def _live():
            torch.cuda.empty_cache()
            t0 = time.time()
            lv = live_pmc_traffic(a, g, D)
            if lv is not None:
                out["roofline"].update({"traffic": lv["bytes_per_call"], "traffic_measured_in_this_run": True,
                                        "traffic_source": "two rocprofv3 --pmc passes (TCC_EA0_RDREQ* / TCC_EA0_WRREQ*, by request size; --kernel-trace only) of "
                                                          "scripts/ingest_only.py inside this run: bsc_ingest + bsc_sync alone on the chip, this run's frames, "
                                                          "token rows and grid",
                                        "traffic_largest_kernels_GB_per_call": lv["largest_kernels_GB_per_call"],
                                        "traffic_over_algorithmic_bytes": lv["bytes_per_call"] / out["roofline"]["bytes_per_call"],
                                        "traffic_passes_seconds": round(time.time() - t0, 1)})
            return {}
        guarded(None, _live)
    if rank == 0:
        print(json.dumps(out))
    if dist_active:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
