"""The Python drop-in (VoxelTokenMemory) driven exactly like the reference drives its own class."""
import os
import random
import types

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


class FakeDino:
    """forward_features(x) -> the fixture's tokens for the current frame (the reference test harness did the same)."""

    def __init__(self, tokens):
        self.tokens, self.frame, self.query = tokens, 0, None

    def forward_features(self, x):
        import torch
        if self.query is not None:
            return {"x_norm_patchtokens": torch.from_numpy(self.query).cuda()}
        t = torch.from_numpy(self.tokens[self.frame]).cuda()
        return {"x_norm_patchtokens": t.reshape(1, -1, t.shape[-1])}


def _memory(cfg, tokens, tmp, **kw):
    import bsc_nav_amd as B
    args = B.MemoryArgs(width=cfg["W"], height=cfg["H"], grid_size=cfg["gs"], cell_size=cfg["cs"],
                        floor_height=cfg["floor_height"], map_height=cfg["map_height"], depth_sample_rate=cfg["s"],
                        query_width=cfg["g"] * 14, query_height=cfg["g"] * 14, memory_path=str(tmp), scene_name="scene",
                        token_dim=cfg["D"], iter_size=cfg.get("iter_size", 50000))
    dino = FakeDino(tokens)
    return B.VoxelTokenMemory(args, preload_dino=dino, need_diffusion=False, alpha_source="host", **kw), dino, args


@pytest.mark.parametrize("name", ["g2_mini_s7_yaw", "g2_c1_s1000", "g3_flush_small_cache"])
def test_dropin_class_matches_reference_end_to_end(tmp_path, name):
    import torch
    z = gu.load(name)
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    mem, dino, args = _memory(cfg, tokens, tmp_path)
    np.random.seed(cfg["seed"])
    random.seed(cfg["seed"])
    for f in range(cfg["F"]):
        dino.frame = f
        mem.obs2voxeltoken({"rgb": rgb[f], "depth": depth[f]}, poses[f])
    assert mem.max_id == int(z["max_id"]) and mem.iter_id == int(z["iter_id"])
    assert np.array_equal(mem.grid_rgb_pos, z["grid_rgb_pos"])
    assert np.array_equal(mem.grid_rgb, z["grid_rgb"])          # host alpha == NumPy's exp: bit-exact bytes
    assert np.array_equal(mem.weight, z["weight"])
    mem.update_memory_dist_base()
    pos, cnt, feats, dists = mem.engine.export_store()
    assert np.array_equal(pos, z["store_pos"]) and np.array_equal(cnt, z["store_cnt"])
    for q in gu.query_specs(z):
        if q["floor"] is not None:
            continue
        dino.query = gu.query_tokens(q, cfg["seed"], cfg["D"], feats)
        kw = {} if q["radius"] is None else dict(region_radius=q["radius"], curr_grid=q["curr"])
        top1, tpos, tsim = mem.voxel_localized(torch.zeros(q["B"], 3, 8, 8), K=q["K"], **kw)
        dino.query = None
        assert top1.shape == (1, 3) and tpos.dtype == np.int64 and tsim.dtype == np.float64   # reference types
        gu.assert_topk_matches(tpos, tsim, q["pos"], q["sim"], tol=5e-6)
    # save in the reference's on-disk layout, reload into a fresh object, same answers
    mem.initial_memory()
    mem.save_memory(original_pos=np.zeros(3, np.float32))
    d = mem.memory_save_path
    assert np.load(d + "/grid_rgb_pos.npy").dtype == np.int32 and np.load(d + "/grid_rgb.npy").dtype == np.uint8
    assert np.load(d + "/weight.npy").dtype == np.float32
    occ = np.load(d + "/occupied_ids.npy")
    assert occ.shape == (cfg["gs"], cfg["gs"], mem.maxh - mem.minh) and occ.dtype == np.int32
    assert int(np.load(d + "/max_id.npy")) == int(z["max_id"])
    assert list(np.load(d + "/map_height.npy")) == [mem.minh, mem.maxh]
    assert os.path.exists(d + "/long_memory.json") and os.path.exists(d + "/feat_features.npy")
    mem2, dino2, args2 = _memory(cfg, tokens, tmp_path)
    args2.load_memory_path = d
    mem2.load_memory()
    q = next(gu.query_specs(z))
    qt = torch.from_numpy(q["pooled"]).cuda()
    a, b = mem.voxel_localized(qt, K=q["K"]), mem2.voxel_localized(qt, K=q["K"])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(mem2.occupied_ids, occ)


def test_dataset_loop_with_random_vit(tmp_path):
    import torch
    import bsc_nav_amd as B
    from bsc_nav_amd import dataset, encoder
    H, W = 96, 128
    vit = encoder.RandomViT("vit_tiny_test", image_size=224).cuda()
    args = B.MemoryArgs(width=W, height=H, grid_size=128, floor_height=-6.4, map_height=6.4, depth_sample_rate=1,
                        query_width=224, query_height=224, patch_size=16, token_dim=vit.out_dim,
                        memory_path=str(tmp_path))
    scenes = [dataset.SyntheticScene("sceneA", 1, 12, H, W, batch=4), dataset.SyntheticScene("sceneB", 2, 8, H, W, batch=4)]
    out = dataset.create_memory_for_dataset(args, scenes, vit, feature_mode="mean", voxel_capacity=200000)
    for name, d in out.items():
        n = int(np.load(d + "/max_id.npy"))
        assert n > 500
        acc, cnt = np.load(d + "/dense_acc.npy"), np.load(d + "/dense_cnt.npy")
        assert acc.shape == (n, vit.out_dim) and cnt.sum() > H * W
        assert np.isfinite(acc).all()
    # second call finds the directories and loads instead of rebuilding (create_memory_for_dataset.py:103)
    out2 = dataset.create_memory_for_dataset(args, scenes, vit, feature_mode="mean", voxel_capacity=200000)
    assert out2 == out


def test_fused_encoder_ops_match_pytorch():
    """bsc_enc_add_layernorm (+ GELU in the GEMM epilogue) against the same ViT evaluated with plain PyTorch ops."""
    import torch
    from bsc_nav_amd import encoder
    torch.manual_seed(0)
    vit = encoder.RandomViT("vit_b16", seed=3).cuda()
    for ln in [m for m in vit.modules() if isinstance(m, torch.nn.LayerNorm)]:   # non-trivial affine parameters
        ln.weight.data = (1 + 0.1 * torch.randn_like(ln.weight.float())).to(ln.weight.dtype)
        ln.bias.data = (0.1 * torch.randn_like(ln.bias.float())).to(ln.bias.dtype)
    rgb = torch.randint(0, 255, (3, 96, 128, 4), dtype=torch.uint8, device="cuda")
    vit.fused = True
    a = vit.patch_tokens(rgb)
    vit.fused = False
    b = vit.patch_tokens(rgb)
    assert a.shape == (3, 14, 14, 768) and torch.isfinite(a).all()
    # bf16 activations: agreement to bf16 resolution of O(1) LayerNorm outputs
    assert (a - b).abs().max().item() < 0.15
    assert (a - b).abs().mean().item() < 0.01
    # the kernel alone, against torch in fp32, is tight
    x = torch.randn(1000, 768, device="cuda").bfloat16()
    d = torch.randn(1000, 768, device="cuda").bfloat16()
    ln = torch.nn.LayerNorm(768, eps=1e-6).cuda().bfloat16()
    ln.weight.data.uniform_(0.5, 1.5)
    ln.bias.data.uniform_(-0.5, 0.5)
    xo, y = vit._add_ln(x.view(1, 1000, 768), d.view(1, 1000, 768), ln, True)
    s_ref = (x + d)
    y_ref = torch.nn.functional.layer_norm(s_ref.float(), (768,), ln.weight.float(), ln.bias.float(), 1e-6)
    assert torch.equal(xo.view(1000, 768), s_ref)
    assert (y.view(1000, 768).float() - y_ref).abs().max().item() < 0.02     # one bf16 rounding of |y| <~ 4


@pytest.mark.parametrize("hw", [(480, 640), (240, 320), (680, 680), (224, 224)])
def test_fused_preprocess_matches_torch_interpolate(hw):
    """bsc_enc_preprocess_patches == /255 -> F.interpolate(bilinear, antialias) -> normalise -> unfold (bf16 rounding)."""
    import ctypes as C
    import torch
    from bsc_nav_amd import _lib, encoder
    H, W = hw
    vit = encoder.RandomViT("vit_tiny_test").cuda()
    rgb = torch.randint(0, 255, (2, H, W, 4), dtype=torch.uint8, device="cuda")
    ref = vit.preprocess(rgb)                                              # (B,3,224,224) float
    g, p = vit.grid, vit.patch
    ref = ref.reshape(2, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(2, g * g, 3 * p * p)
    out = torch.empty((2, g * g, 3 * p * p), dtype=torch.bfloat16, device="cuda")
    mean = (C.c_float * 3)(*encoder.IMAGENET_MEAN)
    std = (C.c_float * 3)(*encoder.IMAGENET_STD)
    _lib.check(_lib.load().bsc_enc_preprocess_patches(C.c_void_p(rgb.data_ptr()), 2, H, W, 4, 224, p,
                                                      C.c_void_p(out.data_ptr()), mean, std,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    assert err.max().item() < 0.02, err.max().item()       # bf16 rounding of values up to ~2.6
    assert err.mean().item() < 0.004


@pytest.mark.parametrize("hw,S,p", [((480, 640), 224, 16), ((240, 320), 224, 16), ((480, 640), 224, 14), ((300, 300), 224, 16)])
def test_tiled_preprocess_equals_generic_kernel_bit_for_bit(hw, S, p):
    """k_preprocess_patches_tiled (RGBA frames: window in LDS, row sums shared by the output rows, window bounds computed inline)
    against the generic one-thread-per-output-pixel kernel (RGB frames): the same taps, weights and summation order — f32 outputs
    must be identical bit for bit, for patch sizes that divide the workgroup (16: shared row sums) and that do not (14)."""
    import ctypes as C
    import torch
    from bsc_nav_amd import _lib, encoder
    H, W = hw
    g = S // p
    torch.manual_seed(H + W + p)
    rgb3 = torch.randint(0, 256, (2, H, W, 3), dtype=torch.uint8, device="cuda")
    rgba = torch.cat([rgb3, torch.full((2, H, W, 1), 255, dtype=torch.uint8, device="cuda")], dim=-1).contiguous()
    mean = (C.c_float * 3)(*encoder.IMAGENET_MEAN)
    std = (C.c_float * 3)(*encoder.IMAGENET_STD)
    outs = []
    for img, ch in ((rgb3, 3), (rgba, 4)):
        out = torch.empty((2, g * g, 3 * p * p), dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().bsc_enc_preprocess_patches_typed(C.c_void_p(img.data_ptr()), 2, H, W, ch, S, p,
                                                                C.c_void_p(out.data_ptr()), 1, mean, std,
                                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all() and outs[0].abs().max().item() > 1.0
    assert torch.equal(outs[0], outs[1])


def test_long_memory_matches_reference(tmp_path):
    """long_memory + long_memory_integration (memory_2.py:905-945, 993-1025) with seeded detector boxes: same objects,
    same voxel locations, same confidences, frame by frame."""
    import torch
    import synth
    z = gu.load("g6_long_memory")
    cfg = gu.cfg_of(z)
    rgb, depth, poses = synth.make_frames(cfg["seed"], cfg["F"], cfg["H"], cfg["W"], cfg["kind"])
    tokens = gu.tag_tokens(synth.make_tokens(cfg["seed"], cfg["F"], cfg["g"], cfg["D"]))
    assert synth.checksum(rgb, depth, poses, tokens) == str(z["input_sha"])
    classes = [str(c) for c in z["classes"]]
    mem, dino, args = _memory(cfg, tokens, tmp_path)
    args.detect_conf, args.detect_classes = 0.55, classes
    mem.args = args

    class Boxes:
        def __init__(self, xyxy, conf, cls):
            self.xyxy, self.conf, self.cls = torch.tensor(xyxy, dtype=torch.float32), torch.tensor(conf), torch.tensor(cls)

        def __len__(self):
            return len(self.conf)

    def detections(f):      # same seeded stand-in as tests/golden/gen_golden.py:fake_detections
        rs = np.random.RandomState(cfg["seed"] * 1000 + f)
        n = rs.randint(0, 6)
        x0 = rs.uniform(0, cfg["W"] - 40, size=n); y0 = rs.uniform(0, cfg["H"] - 40, size=n)
        x1 = x0 + rs.uniform(8, 39, size=n); y1 = y0 + rs.uniform(8, 39, size=n)
        return np.stack([x0, y0, x1, y1], 1).reshape(n, 4), rs.uniform(0.55, 0.99, size=n), rs.randint(0, len(classes), size=n)

    np.random.seed(cfg["seed"])
    random.seed(cfg["seed"])
    counts = []
    for f in range(cfg["F"]):
        dino.frame = f
        b = Boxes(*detections(f))
        mem.yolow = types.SimpleNamespace(predict=lambda img, conf=None, _b=b: [types.SimpleNamespace(boxes=_b)])
        obs = {"rgb": rgb[f], "depth": depth[f]}
        mem.obs2voxeltoken(obs, poses[f])
        mem.long_memory(obs)
        counts.append(len(mem.long_memory_dict))
    assert counts == [int(v) for v in z["per_frame"]]
    lm = mem.long_memory_dict
    assert [classes.index(o["label"]) for o in lm] == [int(v) for v in z["label"]]
    assert np.array_equal(np.array([o["loc"] for o in lm]), z["loc"])
    np.testing.assert_allclose([o["confidence"] for o in lm], z["confidence"], rtol=0, atol=0)


@pytest.mark.parametrize("case", ["f1", "f2", "f3", "f4", "f5", "f6", "f7"])
def test_frontier_helpers_match_reference(tmp_path, case):
    """FrontierExplorer helpers (memory_2.py:1147-1311) on the device-resident top-down map vs the reference's goldens:
    through the C-ABI (engine) and through the drop-in class's own method names."""
    import bsc_nav_amd as B
    z = gu.load("g7_frontier")
    cv, nav = z[f"{case}_cv_map"], z[f"{case}_nav"]
    gs, min_size, radius = (int(v) for v in z[f"{case}_params"])
    args = B.MemoryArgs(width=64, height=48, grid_size=gs, cell_size=0.1, floor_height=-1.0, map_height=2.0,
                        query_width=224, query_height=224, memory_path=str(tmp_path), scene_name="scene", token_dim=16)
    origin = np.array([1.5, 0.25, -2.0])

    def is_navigable(loc):
        col = int(round((loc[0] - origin[0]) / 0.1)) + gs // 2
        row = int(round((loc[2] - origin[2]) / 0.1)) + gs // 2
        return bool(nav[row, col])
    env = types.SimpleNamespace(original_state=types.SimpleNamespace(position=origin),
                                plnner=types.SimpleNamespace(pathfinder=types.SimpleNamespace(is_navigable=is_navigable)))
    mem = B.VoxelTokenMemory(args, preload_dino=None, need_diffusion=False, env=env, feature_mode="mean")
    mem.engine.import_cv_map(cv)
    mem.min_cluster_size, mem.ig_radius = min_size, radius
    # ---- C-ABI
    mask = mem.engine.frontier_mask(nav)
    assert np.array_equal(np.argwhere(mask & 2), z[f"{case}_frontiers"])
    assert np.array_equal((mask & 1).astype(bool), cv.sum(-1) != 0)
    r = mem.engine.frontier_clusters(None, min_size, radius, max_clusters=gs * gs)
    assert r["n"] == len(z[f"{case}_sizes"])
    assert np.array_equal(r["labels"], z[f"{case}_labels"])
    assert np.array_equal(r["sizes"], z[f"{case}_sizes"]) and np.array_equal(r["first"], z[f"{case}_first"])
    assert np.array_equal(r["centers"], z[f"{case}_centers"]) and np.array_equal(r["gains"], z[f"{case}_gains"])
    want = z[f"{case}_best"]
    assert (r["best"] == -1) if np.isnan(want[0]) else np.array_equal(r["centers"][r["best"]], want)
    small = mem.engine.frontier_clusters(None, min_size, radius, max_clusters=2, labels=False)      # truncated output
    assert small["n"] == r["n"] and np.array_equal(small["sizes"], r["sizes"][:2])
    # ---- the reference's method names
    navigable_mask = mem.build_navigable_mask()
    assert np.array_equal(navigable_mask, z[f"{case}_navigable_mask"])
    frontiers = mem.find_frontiers(navigable_mask)
    assert frontiers == [tuple(int(v) for v in f) for f in z[f"{case}_frontiers"]]
    clusters = mem.cluster_frontiers(frontiers)
    assert [len(c) for c in clusters] == z[f"{case}_sizes"].tolist()
    assert [min(c) for c in clusters] == [tuple(int(v) for v in f) for f in z[f"{case}_first"]]
    for k, c in enumerate(clusters):
        assert mem.compute_cluster_center(c) == tuple(z[f"{case}_centers"][k])
        assert mem.compute_information_gain(*mem.compute_cluster_center(c)) == z[f"{case}_gains"][k]
    best = mem.select_best_cluster_center_by_ig(clusters)
    assert (best is None) if np.isnan(want[0]) else best == tuple(want)
    assert np.array_equal(np.stack([mem.grid2loc_2d(3, 7), mem.grid2loc_2d(gs - 1, 0)]), z[f"{case}_loc"])
    assert [list(mem.loc2grid_2d(0.37, -1.21)), list(mem.loc2grid_2d(-2.05, 0.0))] == z[f"{case}_l2g"].tolist()
    mem.update_frontier_map(frontiers, clusters, best, navigable_mask)
    assert mem.FrontierMap.shape == (gs, gs, 3) and (mem.FrontierMap[cv.sum(-1) == 0].sum() == 0 or best is not None)
    mem.engine.close()


@pytest.mark.parametrize("B,T,H", [(3, 197, 12), (2, 261, 16), (1, 16, 2), (2, 33, 3), (1, 224, 1)])
def test_fused_attention_matches_fp32_softmax(B, T, H):
    """bsc_enc_attention (K, V resident in LDS, MFMA bf16, f32 softmax) against softmax(QK^T/8)V evaluated in fp32 on the
    same bf16 inputs: output error bounded by the bf16 rounding of P and of the result."""
    import ctypes as C
    import torch
    from bsc_nav_amd import _lib
    torch.manual_seed(B * 1000 + T)
    qkv = (torch.randn(B, T, 3, H, 64, device="cuda") * 1.5).bfloat16()
    out = torch.empty(B, T, H * 64, dtype=torch.bfloat16, device="cuda")
    _lib.check(_lib.load().bsc_enc_attention(C.c_void_p(qkv.data_ptr()), B, T, H, 64, C.c_void_p(out.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, T, H * 64)
    err = (out.float() - ref).abs()
    assert torch.isfinite(out.float()).all()
    assert err.max().item() < 0.03 and err.mean().item() < 0.003, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("arch", ["vit_b16", "vit_l14"])
def test_fused_encoder_ends_match_pytorch(arch):
    """The fused bf16 encoder paths — bias-lagged residual stream (bsc_enc_bias_layernorm, beta = 1 GEMMs) and residual adds
    in the LayerNorm kernel (bsc_enc_add_layernorm / _final_layernorm), both with bsc_enc_embed_layernorm (registers, cls,
    pos) — against the same ViT evaluated in f32 with plain PyTorch ops: no further from it than plain bf16 PyTorch is."""
    import copy
    import torch
    from bsc_nav_amd import encoder
    torch.manual_seed(1)
    vit = encoder.RandomViT(arch, seed=5).cuda()
    for prm in (vit.cls, vit.pos) + ((vit.reg,) if vit.reg is not None else ()):
        prm.data = (0.5 * torch.randn_like(prm.float())).to(prm.dtype)
    for blk in vit.blocks:                      # biases that matter (the lagged stream carries their running sum)
        for lin in (blk.proj, blk.fc2, blk.fc1, blk.qkv):
            lin.bias.data = (0.1 * torch.randn_like(lin.bias.float())).to(lin.bias.dtype)
    rgb = torch.randint(0, 255, (2, 60, 80, 4), dtype=torch.uint8, device="cuda")
    ref = copy.deepcopy(vit).float()
    ref.compute_dtype, ref.fused = torch.float32, False
    r = ref.patch_tokens(rgb)
    vit.fused = False
    e_plain = (vit.patch_tokens(rgb) - r).abs()
    g = vit.grid
    for lagged in (True, False):
        vit.fused, vit.lagged = True, lagged
        a32 = vit.patch_tokens(rgb)
        a16 = vit.patch_tokens(rgb, keep_dtype=True)
        assert a32.shape == (2, g, g, vit.width) and a32.dtype == torch.float32 and a16.dtype == torch.bfloat16
        assert torch.equal(a16.float(), a32)                      # the f32 output is the bf16 result widened
        e = (a32 - r).abs()
        assert e.mean().item() < 1.1 * e_plain.mean().item() + 1e-4, (lagged, e.mean().item(), e_plain.mean().item())
        assert e.max().item() < 2.0 * e_plain.max().item() and e.max().item() < 0.25, (lagged, e.max().item())


def _dinov2_state_dict(width, depth, heads, mlp, regs, patch=14, pos_grid=37, seed=0):
    """A state_dict with DINOv2's parameter names and shapes (facebookresearch/dinov2 vision_transformer.py), random values:
    LayerScale gammas far from 1 and a position embedding on the 37x37 training grid, so that a wrong fold or a wrong
    resampling shows."""
    import torch
    g = torch.Generator().manual_seed(seed)
    rn = lambda *sh, std=0.02: torch.randn(*sh, generator=g) * std
    sd = {"patch_embed.proj.weight": rn(width, 3, patch, patch, std=0.05), "patch_embed.proj.bias": rn(width, std=0.1),
          "cls_token": rn(1, 1, width, std=0.5), "pos_embed": rn(1, 1 + pos_grid * pos_grid, width, std=0.5),
          "mask_token": rn(1, width), "norm.weight": 1 + rn(width, std=0.2), "norm.bias": rn(width, std=0.2)}
    if regs:
        sd["register_tokens"] = rn(1, regs, width, std=0.5)
    for i in range(depth):
        k = f"blocks.{i}."
        sd.update({k + "norm1.weight": 1 + rn(width, std=0.2), k + "norm1.bias": rn(width, std=0.2),
                   k + "attn.qkv.weight": rn(3 * width, width, std=0.04), k + "attn.qkv.bias": rn(3 * width, std=0.1),
                   k + "attn.proj.weight": rn(width, width, std=0.04), k + "attn.proj.bias": rn(width, std=0.1),
                   k + "ls1.gamma": 0.5 + torch.rand(width, generator=g), k + "ls2.gamma": 0.5 + torch.rand(width, generator=g),
                   k + "norm2.weight": 1 + rn(width, std=0.2), k + "norm2.bias": rn(width, std=0.2),
                   k + "mlp.fc1.weight": rn(mlp, width, std=0.04), k + "mlp.fc1.bias": rn(mlp, std=0.1),
                   k + "mlp.fc2.weight": rn(width, mlp, std=0.03), k + "mlp.fc2.bias": rn(width, std=0.1)})
    return sd


def _dinov2_forward_f32(sd, x, heads, regs, antialias, offset):
    """DinoVisionTransformer.forward_features()['x_norm_patchtokens'] restated with plain f32 PyTorch ops: strided patch
    convolution, cls + interpolated pos (bicubic), registers after the cls token, pre-LN blocks with LayerScale, erf GELU."""
    import torch
    import torch.nn.functional as F
    sd = {k: v.cuda().float() for k, v in sd.items()}
    p = sd["patch_embed.proj.weight"].shape[-1]
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=p)
    B, D, gh, gw = t.shape
    t = t.flatten(2).transpose(1, 2)
    pos = sd["pos_embed"]
    m = int(round((pos.shape[1] - 1) ** 0.5))
    if m != gh:
        grid = pos[:, 1:].reshape(1, m, m, D).permute(0, 3, 1, 2)
        if offset:
            grid = F.interpolate(grid, scale_factor=((gh + offset) / m, (gw + offset) / m), mode="bicubic", antialias=antialias)
        else:
            grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", antialias=antialias)
        pos = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, D)], dim=1)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1) + pos
    if regs:
        t = torch.cat([t[:, :1], sd["register_tokens"].expand(B, -1, -1), t[:, 1:]], dim=1)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    hd = D // heads
    for i in range(depth):
        k = f"blocks.{i}."
        y = F.layer_norm(t, (D,), sd[k + "norm1.weight"], sd[k + "norm1.bias"], 1e-6)
        qkv = F.linear(y, sd[k + "attn.qkv.weight"], sd[k + "attn.qkv.bias"]).reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        a = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) * hd ** -0.5, dim=-1) @ qkv[2]
        a = F.linear(a.transpose(1, 2).reshape(B, -1, D), sd[k + "attn.proj.weight"], sd[k + "attn.proj.bias"])
        t = t + sd[k + "ls1.gamma"] * a
        y = F.layer_norm(t, (D,), sd[k + "norm2.weight"], sd[k + "norm2.bias"], 1e-6)
        h = F.linear(F.gelu(F.linear(y, sd[k + "mlp.fc1.weight"], sd[k + "mlp.fc1.bias"])), sd[k + "mlp.fc2.weight"], sd[k + "mlp.fc2.bias"])
        t = t + sd[k + "ls2.gamma"] * h
    t = F.layer_norm(t, (D,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return t[:, 1 + regs:]


@pytest.mark.parametrize("shape", [dict(width=1024, depth=24, heads=16, mlp=4096, regs=4), dict(width=768, depth=12, heads=12, mlp=3072, regs=0)])
def test_dinov2_state_dict_runs_through_the_fused_encoder(shape):
    """RandomViT.from_dinov2_state_dict: a checkpoint in DINOv2's own layout (the reference's `preload_dino`,
    dinov2_vitl14_reg by default) evaluated by the fused bf16 path equals the architecture restated in f32 to bf16 accuracy —
    patch convolution as GEMM, pos-embedding resampling (antialiased for the register model, offset 0.1 otherwise),
    register insertion, LayerScale folded into the weights, lagged residual stream."""
    import torch
    from bsc_nav_amd import encoder
    sd = _dinov2_state_dict(shape["width"], shape["depth"], shape["heads"], shape["mlp"], shape["regs"], seed=3)
    vit = encoder.RandomViT.from_dinov2_state_dict(sd, image_size=224).cuda()
    assert vit.arch == ("vit_l14" if shape["regs"] else "vit_b14") and vit.grid == 16
    torch.manual_seed(0)
    x = torch.randn(2, 3, 224, 224, device="cuda")
    ref = _dinov2_forward_f32(sd, x, shape["heads"], shape["regs"], antialias=shape["regs"] > 0, offset=0.0 if shape["regs"] else 0.1)
    vit.fused = False
    e_plain = (vit.forward_features(x)["x_norm_patchtokens"] - ref).abs()
    vit.fused = True
    out = vit.forward_features(x)["x_norm_patchtokens"]
    e = (out - ref).abs()
    assert out.shape == ref.shape == (2, 256, shape["width"])
    assert e_plain.mean().item() < 0.03 and e.mean().item() < 0.03, (e.mean().item(), e_plain.mean().item())     # outputs are O(1)
    assert e.mean().item() < 1.15 * e_plain.mean().item() + 1e-3 and e.max().item() < 0.5, (e.mean().item(), e.max().item())


def test_dinov2_state_dict_at_the_reference_precision():
    """the same checkpoint through the f32 encoder (dtype=torch.float32: split-operand GEMMs, LayerNorm folded into them,
    LayerScale in the weights, the exact erf GELU of DINOv2's nn.GELU() in the fc1 epilogue): equals the architecture restated in
    f32 — erf form — to 1e-5 on the tokens (round 5 ran tanh-GELU there: 5e-4 mean, 2e-2 max); VoxelTokenMemory(fuse_encoder="f32")
    selects it"""
    import torch
    import bsc_nav_amd as B
    from bsc_nav_amd import encoder
    shape = dict(width=1024, depth=24, heads=16, mlp=4096, regs=4)
    sd = _dinov2_state_dict(shape["width"], shape["depth"], shape["heads"], shape["mlp"], shape["regs"], seed=3)
    vit = encoder.RandomViT.from_dinov2_state_dict(sd, image_size=224, dtype=torch.float32).cuda()
    assert vit.split_gemm and vit.arch == "vit_l14"
    torch.manual_seed(0)
    x = torch.randn(2, 3, 224, 224, device="cuda")
    ref = _dinov2_forward_f32(sd, x, shape["heads"], shape["regs"], antialias=True, offset=0.0)
    out = vit.forward_features(x)["x_norm_patchtokens"]
    e = (out - ref).abs()
    assert out.dtype == torch.float32 and out.shape == ref.shape
    assert vit.gelu == "erf"
    assert e.mean().item() < 2e-6 and e.max().item() < 2e-5, (e.mean().item(), e.max().item())

    class HubModule:
        def state_dict(self):
            return sd

    args = B.MemoryArgs(width=160, height=120, grid_size=128, cell_size=0.1, floor_height=-6.4, map_height=6.4, depth_sample_rate=1,
                        query_width=224, query_height=224, memory_path="/tmp", scene_name="fuse32", token_dim=1024, patch_size=14)
    mem = B.VoxelTokenMemory(args, preload_dino=HubModule(), need_diffusion=False, feature_mode="mean", fuse_encoder="f32",
                             max_frames_per_call=2, voxel_capacity=100_000)
    assert isinstance(mem.dinov2, encoder.RandomViT) and mem.dinov2.compute_dtype == torch.float32 and mem.dinov2.split_gemm
    mem.engine.close()


def test_graphed_encoder_equals_eager_and_does_not_keep_old_frames():
    """encoder.GraphedEncoder (what bench.py times): the captured graph starts at the patch matrix, the preprocessing kernel
    runs eagerly on the caller's frames — replaying it on new frames must give the eager result for THOSE frames (bf16 and f32
    token outputs), also for a non-contiguous view of a larger buffer."""
    import torch
    from bsc_nav_amd import encoder
    vit = encoder.RandomViT("vit_b16", image_size=224, seed=4).cuda()
    a = torch.randint(0, 255, (3, 120, 160, 4), dtype=torch.uint8, device="cuda")
    b = torch.randint(0, 255, (6, 120, 160, 4), dtype=torch.uint8, device="cuda")
    for keep in (True, False):
        enc = encoder.GraphedEncoder(vit, 3, 120, 160, 4, keep)
        for frames in (a, b[3:], b[::2]):
            out = enc(frames).clone()
            ref = vit.patch_tokens(frames.contiguous(), keep)
            assert out.dtype == ref.dtype and out.shape == ref.shape == (3, 14, 14, 768)
            assert torch.equal(out, ref)
    # the f32 encoder (what bench.py's headline runs): graph from the padded piece matrix, patch 16 and patch 14
    for arch in ("vit_b16", "vit_s14_reg"):
        v32 = encoder.RandomViT(arch, image_size=224, seed=4, dtype=torch.float32).cuda()
        enc = encoder.GraphedEncoder(v32, 3, 120, 160, 4, False)
        assert enc.f32 and enc.from_patches
        for frames in (a, b[3:], b[::2]):
            out = enc(frames).clone()
            ref = v32.patch_tokens(frames.contiguous())
            assert out.dtype == torch.float32 and out.shape == ref.shape and torch.equal(out, ref)


def test_fuse_encoder_option_takes_a_dinov2_module():
    """VoxelTokenMemory(preload_dino=<module with DINOv2's state_dict>, fuse_encoder=True): the weights move into the fused
    encoder, ingest_frames runs through its batch entry, and the map equals the one built from that encoder's own tokens."""
    import torch
    import bsc_nav_amd as B
    import synth
    from bsc_nav_amd import encoder
    sd = _dinov2_state_dict(768, 12, 12, 3072, 0, seed=9)

    class HubModule:                                    # what torch.hub returns, as far as this path looks at it
        def state_dict(self):
            return sd

        def forward_features(self, x):
            raise AssertionError("the f32 module must not run when fuse_encoder=True")

    H, W, F = 120, 160, 3
    args = B.MemoryArgs(width=W, height=H, grid_size=128, cell_size=0.1, floor_height=-6.4, map_height=6.4, depth_sample_rate=1,
                        query_width=224, query_height=224, memory_path="/tmp", scene_name="fuse", token_dim=768, patch_size=14)
    mem = B.VoxelTokenMemory(args, preload_dino=HubModule(), need_diffusion=False, feature_mode="mean", fuse_encoder=True,
                             max_frames_per_call=F, voxel_capacity=200_000)
    assert isinstance(mem.dinov2, encoder.RandomViT) and mem.dinov2.arch == "vit_b14" and mem.dinov2.grid == 16
    rgb, depth, poses = synth.make_frames(3, F, H, W, "room")
    rgb4 = np.concatenate([rgb, np.full(rgb.shape[:3] + (1,), 255, np.uint8)], axis=-1)
    d_rgb, d_depth = torch.from_numpy(rgb4).cuda(), torch.from_numpy(depth).cuda()
    mem.ingest_frames(d_rgb, d_depth, poses)
    acc, cnt = mem.engine.export_dense()
    tok = mem.dinov2.patch_tokens(d_rgb)
    eng = B.VoxelEngine(H, W, 128, 0.1, -6.4, 6.4, 16, 768, mode="mean", voxel_capacity=200_000, max_points=F * H * W)
    Ts = np.stack([mem.chain.pc_transform(p) for p in poses])       # the memory's own pose chain: same anchor, same transforms
    eng.ingest(d_depth, d_rgb, tok, Ts)
    acc2, cnt2 = eng.export_dense()
    assert np.array_equal(cnt, cnt2) and cnt.sum() > 0.9 * F * H * W
    np.testing.assert_allclose(acc, acc2, rtol=1e-5, atol=1e-4)
    eng.close()


def _write_reference_dir(path, z, name, store_arrays=None):
    """A memory directory as the reference leaves it (memory_2.py:1136-1145): npy set + long_memory.json; the token store
    only as feat.h5df (through the stand-in) when `store_arrays` is given."""
    import json
    gs, cs, fh, mh = int(z["grid"][0]), float(z["grid"][1]), float(z["grid"][2]), float(z["grid"][3])
    minh, maxh = int(fh / cs), int(mh / cs)
    pos, rgb, w = z[f"{name}_pos"], z[f"{name}_rgb"], z[f"{name}_weight"]
    occ = np.full((gs, gs, maxh - minh), -1, np.int32)
    occ[pos[:, 0], pos[:, 1], pos[:, 2]] = np.arange(len(pos))
    os.makedirs(path, exist_ok=True)
    np.save(path + "/grid_rgb_pos.npy", pos); np.save(path + "/grid_rgb.npy", rgb); np.save(path + "/weight.npy", w)
    np.save(path + "/occupied_ids.npy", occ); np.save(path + "/max_id.npy", np.array(len(pos)))
    np.save(path + "/original_pos.npy", np.array([1.0, float(z[f"{name}_current_height"]), -2.0], np.float32))
    np.save(path + "/map_height.npy", np.array([minh, maxh])); np.save(path + "/base_height.npy", z[f"{name}_base_height"])
    zlo, zhi = int(pos[:, 2].min()), int(pos[:, 2].max())
    with open(path + "/long_memory.json", "w") as f:
        json.dump([{"label": "chair", "loc": [3, 4, zlo + 1], "confidence": 0.9},
                   {"label": "sofa", "loc": [7, 8, zhi - 1], "confidence": 0.8}], f)
    if store_arrays is not None:
        from bsc_nav_amd import store
        store.write_h5_store(path + "/feat.h5df", *store_arrays)
        open(path + "/feat.h5df", "w").close()
    return gs, cs, fh, mh


@pytest.mark.parametrize("name", ["one_floor", "two_floors_high", "three_floors_mid_with_noise", "few_samples"])
def test_load_memory_single_floor_matches_reference(tmp_path, name):
    """--load_single_floor (the flag of every README benchmark command): a reference-layout directory whose token store
    exists only as feat.h5df is loaded; floor heights / z-range / per-floor files / long-memory filter equal the
    reference's own load_memory outputs (g8_floor_split.npz) and voxel_localized only returns voxels of that floor."""
    import sys
    import torch
    import bsc_nav_amd as B
    import fake_h5py
    z = gu.load("g8_floor_split")
    saved = sys.modules.get("h5py")
    fake_h5py.install()
    try:
        pos = z[f"{name}_pos"]
        rs = np.random.RandomState(5)
        D = 16
        # exact-mode token store: 1..3 tokens per voxel, in HDF5 name order
        from bsc_nav_amd import dist as bd
        k0, k1, k2 = bd.name_keys_np(pos)
        order = np.lexsort((k2, k1, k0))
        spos = pos[order]
        cnt = rs.randint(1, 4, size=len(spos)).astype(np.int32)
        feats = rs.standard_normal((int(cnt.sum()), D)).astype(np.float32)
        dists = rs.uniform(0.1, 9, size=int(cnt.sum())).astype(np.float32)
        d = str(tmp_path / "scene")
        gs, cs, fh, mh = _write_reference_dir(d, z, name, (spos, cnt, feats, dists))
        args = B.MemoryArgs(width=64, height=48, grid_size=gs, cell_size=cs, floor_height=fh, map_height=mh,
                            query_width=56, query_height=56, token_dim=D, memory_path=str(tmp_path), scene_name="other",
                            load_memory_path=d, load_single_floor=True)
        mem = B.VoxelTokenMemory(args, need_diffusion=False, voxel_capacity=100, token_capacity=64)   # too small on purpose
        mem.load_memory()
        assert mem.engine.cfg.voxel_capacity >= len(pos) and mem.engine.cfg.token_capacity >= len(feats)
        np.testing.assert_array_equal(np.array(mem.floor_heights), z[f"{name}_floor_heights"])
        assert mem.num_floors == int(z[f"{name}_num_floors"])
        assert [mem.floor_min_height, mem.floor_max_height] == z[f"{name}_range"].tolist()
        k = int(z[f"{name}_current_floor"])
        assert np.array_equal(np.load(d + f"/grid_rgb_pos_floor_{k}.npy"), z[f"{name}_floor_pos"])
        assert np.array_equal(np.load(d + f"/grid_rgb_floor_{k}.npy"), z[f"{name}_floor_rgb"])
        assert [o["label"] for o in mem.long_memory_filter()] == [str(s) for s in z[f"{name}_long_memory_filtered"]]
        assert np.array_equal(mem.grid_rgb_pos, pos) and np.array_equal(mem.weight, z[f"{name}_weight"])
        # the scan honours the floor z-range (memory_2.py:633-640): same answer as an explicit NumPy scan of the floor
        q = rs.standard_normal(D).astype(np.float32)
        top1, tpos, tsim = mem.voxel_localized(torch.from_numpy(q), K=30)
        off = np.concatenate([[0], np.cumsum(cnt)])
        fn = feats / np.maximum(np.linalg.norm(feats, axis=1, keepdims=True), 1e-8)
        sims = fn @ (q / np.linalg.norm(q))
        best = np.array([sims[off[i]:off[i + 1]].max() for i in range(len(spos))])
        on_floor = (spos[:, 2] >= mem.floor_min_height) & (spos[:, 2] <= mem.floor_max_height)
        cand = np.flatnonzero(on_floor)
        top = cand[np.argsort(-best[cand], kind="stable")[:30]]
        assert len(tpos) == min(30, len(cand)) and np.all((tpos[:, 2] >= mem.floor_min_height) & (tpos[:, 2] <= mem.floor_max_height))
        assert np.array_equal(tpos, spos[top].astype(np.int64))
        np.testing.assert_allclose(tsim, best[top], atol=3e-6, rtol=0)
        # small exported arrays are cached between map changes (BSCAgent.py:179-218 reads them repeatedly) and read-only:
        # an in-place edit by a caller must not leak into later reads; the 800 MB occupied_ids is a fresh copy per read
        assert mem.grid_rgb is mem.grid_rgb and not mem.grid_rgb.flags.writeable and not mem.weight.flags.writeable
        with pytest.raises(ValueError):
            mem.grid_rgb[0, 0] = 1
        assert mem.occupied_ids is not mem.occupied_ids
    finally:
        if saved is None:
            sys.modules.pop("h5py", None)
        else:
            sys.modules["h5py"] = saved


def test_exact_mode_defaults_to_reference_exact_alpha(tmp_path):
    """feature_mode='exact' without an explicit alpha_source reproduces the reference's rgb bytes bit for bit."""
    z = gu.load("g2_mini_s7_yaw")
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    import bsc_nav_amd as B
    args = B.MemoryArgs(width=cfg["W"], height=cfg["H"], grid_size=cfg["gs"], cell_size=cfg["cs"],
                        floor_height=cfg["floor_height"], map_height=cfg["map_height"], depth_sample_rate=cfg["s"],
                        query_width=cfg["g"] * 14, query_height=cfg["g"] * 14, memory_path=str(tmp_path), token_dim=cfg["D"])
    dino = FakeDino(tokens)
    mem = B.VoxelTokenMemory(args, preload_dino=dino, need_diffusion=False)
    assert mem.alpha_source == "host" and B.VoxelTokenMemory(args, need_diffusion=False, feature_mode="mean").alpha_source == "device"
    np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    for f in range(cfg["F"]):
        dino.frame = f
        mem.obs2voxeltoken({"rgb": rgb[f], "depth": depth[f]}, poses[f])
    assert np.array_equal(mem.grid_rgb, z["grid_rgb"]) and np.array_equal(mem.weight, z["weight"])


def test_batched_ingest_honours_host_alpha(tmp_path):
    """ingest_frames with alpha_source='host' (the exact mode's default) gives the same rgb bytes / weights as the
    frame-by-frame entry, i.e. the reference's: both entry points of one object agree bit for bit."""
    import torch
    z = gu.load("g2_mini_s7_yaw")
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    import bsc_nav_amd as B
    args = B.MemoryArgs(width=cfg["W"], height=cfg["H"], grid_size=cfg["gs"], cell_size=cfg["cs"],
                        floor_height=cfg["floor_height"], map_height=cfg["map_height"], depth_sample_rate=cfg["s"],
                        query_width=cfg["g"] * 14, query_height=cfg["g"] * 14, memory_path=str(tmp_path), token_dim=cfg["D"])
    mem = B.VoxelTokenMemory(args, need_diffusion=False, max_frames_per_call=cfg["F"])
    assert mem.alpha_source == "host"
    np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    mem.ingest_frames(torch.from_numpy(np.ascontiguousarray(rgb)).cuda(), torch.from_numpy(np.ascontiguousarray(depth)).cuda(),
                      poses, tokens=torch.from_numpy(np.ascontiguousarray(tokens)).cuda())
    assert np.array_equal(mem.grid_rgb_pos, z["grid_rgb_pos"])
    assert np.array_equal(mem.grid_rgb, z["grid_rgb"]) and np.array_equal(mem.weight, z["weight"])


def test_ingest_frames_with_a_forward_features_only_encoder(tmp_path):
    """An encoder that only has the reference's forward_features contract (a real DINOv2) works in the batched entry."""
    import torch
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic
    H, W, g, D, F = 48, 64, 4, 16, 3

    class Dino:
        def forward_features(self, x):
            assert x.shape[1:] == (3, 56, 56) and x.dtype == torch.float32
            t = x.reshape(x.shape[0], 3, g, 14, g, 14).mean(dim=(3, 5)).permute(0, 2, 3, 1).reshape(x.shape[0], g * g, 3)
            return {"x_norm_patchtokens": torch.cat([t, torch.ones(x.shape[0], g * g, D - 3, device=x.device)], -1)}

    args = B.MemoryArgs(width=W, height=H, grid_size=128, floor_height=-6.4, map_height=6.4, depth_sample_rate=1,
                        query_width=56, query_height=56, token_dim=D, memory_path=str(tmp_path))
    mem = B.VoxelTokenMemory(args, preload_dino=Dino(), need_diffusion=False, feature_mode="mean", max_frames_per_call=F)
    poses = synthetic.random_walk_poses(3, F)
    rgb, depth, _ = synthetic.make_frames(3, F, H, W, "room", poses=poses)
    mem.ingest_frames(rgb, depth, poses)
    acc, cnt = mem.engine.export_dense()
    assert mem.max_id > 100 and np.isfinite(acc).all() and int(cnt.sum()) == mem.engine.counters()["points_passed"]
    np.testing.assert_allclose(acc[:, 3:] / cnt[:, None], 1.0, rtol=1e-5)


class _FakeEnv:
    """The slice of NavEnv (env.py:49-296) the memory-building loops touch, over the seeded synthetic room."""

    def __init__(self, H, W, seed=0):
        import synth
        self.H, self.W, self.rs = H, W, np.random.RandomState(seed)
        self.pos, self.k = np.zeros(3), 0
        self._synth = synth
        env = self

        class _Sims:
            def get_sensor_observations(self, _):
                return env._obs()

            def step(self, action):
                if action == "move_forward":
                    th = env.k * np.pi / 6
                    nxt = env.pos + np.array([-np.sin(th), 0.0, -np.cos(th)]) * 0.25
                    if np.all(np.abs(nxt[[0, 2]]) < [3.3, 2.3]):
                        env.pos = nxt
                elif action == "turn_left":
                    env.k += 1
                elif action == "turn_right":
                    env.k -= 1
                return env._obs()

        class _Agent:
            def get_state(self):
                th = env.k * np.pi / 6
                return types.SimpleNamespace(position=env.pos.copy(),
                                             rotation=types.SimpleNamespace(x=0.0, y=np.sin(th / 2), z=0.0, w=np.cos(th / 2)))

        class _Pathfinder:
            def get_random_navigable_point(self):
                return np.array([env.rs.uniform(-3, 3), 0.0, env.rs.uniform(-2, 2)])

            def get_island(self, p):
                return 0

            def is_navigable(self, p):
                return bool(abs(p[0]) < 3.4 and abs(p[2]) < 2.4)

        self.sims, self.agent = _Sims(), _Agent()
        self.plnner = types.SimpleNamespace(pathfinder=_Pathfinder())
        self.original_state = types.SimpleNamespace(position=np.zeros(3, np.float32))
        self.resets = 0

    def _obs(self):
        th = self.k * np.pi / 6
        pose = np.array([*self.pos, 0.0, np.sin(th / 2), 0.0, np.cos(th / 2)])
        d = self._synth._room_depth(self.H, self.W, pose).astype(np.float32)
        rgb = self.rs.randint(0, 255, size=(self.H, self.W, 4)).astype(np.uint8)
        return {"rgb": rgb, "depth": d}

    def reset(self, args, init_state=None, build_map=False):
        self.resets += 1

    def move2point(self, goal):
        return ["move_forward"] * 3 + ["turn_left"], goal

    def get_random_navigable_point_near(self, p):
        return np.asarray(p)


def test_simulator_driven_loops_over_a_fake_env(tmp_path):
    """excute / exploring_create_memory / explore_entire_space (memory_2.py:1086-1145, :1347-1391) driven by a stand-in
    for NavEnv: the call order of the reference (initial_memory -> steps with obs2voxeltoken -> final flush -> save) produces
    a memory directory in the reference layout; the frontier loop runs on the resident top-down map."""
    import bsc_nav_amd as B
    H, W, g, D = 48, 64, 4, 16
    args = B.MemoryArgs(width=W, height=H, grid_size=128, floor_height=-6.4, map_height=6.4, depth_sample_rate=5,
                        query_width=g * 14, query_height=g * 14, token_dim=D, memory_path=str(tmp_path), scene_name="sim",
                        random_move_num=2, turn_left=90)
    tok = np.random.RandomState(1).standard_normal((1, g, g, D)).astype(np.float32)
    dino = FakeDino(tok)
    env = _FakeEnv(H, W)
    mem = B.VoxelTokenMemory(args, preload_dino=dino, need_diffusion=False, env=env)
    np.random.seed(0); random.seed(0)
    mem.exploring_create_memory()
    d = mem.memory_save_path
    n = int(np.load(d + "/max_id.npy"))
    assert n > 100 and len(np.load(d + "/base_height.npy")) == 2 and os.path.exists(d + "/feat_features.npy")
    assert np.load(d + "/grid_rgb_pos.npy").shape == (n, 3) and mem.iter_id == 0          # final flush emptied the cache
    steps = 2 * (4 + 4)                                                                    # 2 goals x (path of 4 + 360 sweep of 4)
    assert mem.engine.counters()["points_seen"] == steps * len(range(0, H * W, 5))
    # the frontier exploration loop on the top-down map the ingest maintains
    mem2 = B.VoxelTokenMemory(args, preload_dino=dino, need_diffusion=False, env=_FakeEnv(H, W, seed=3), memory_path=str(tmp_path / "sim2"))
    mem2.explore_entire_space(max_iterations=2)
    assert mem2.max_id > 100 and (mem2.cv_map.sum(-1) != 0).sum() > 50
    assert hasattr(mem2, "FrontierMap") or mem2.find_frontiers(mem2.build_navigable_mask()) == []


@pytest.mark.parametrize("B,T,skip,out_f32", [(3, 7, 0, 0), (5, 9, 2, 0), (2, 197, 1, 1), (1, 1, 0, 0)])
def test_bias_layernorm_rows_and_tails(B, T, skip, out_f32):
    """bsc_enc_bias_layernorm (four rows per wavefront): row counts that are not multiples of four, the skip form (final
    LayerNorm of the patch rows only), f32 output = the bf16 result widened; against LayerNorm(u + bias_sum) in f32."""
    import ctypes as C
    import torch
    from bsc_nav_amd import _lib
    torch.manual_seed(B * 100 + T)
    Wd = 768
    u = torch.randn(B, T, Wd, device="cuda").to(torch.bfloat16)
    bs = 0.3 * torch.randn(Wd, device="cuda")
    ga = (1 + 0.2 * torch.randn(Wd, device="cuda")).to(torch.bfloat16)
    be = (0.2 * torch.randn(Wd, device="cuda")).to(torch.bfloat16)
    y = torch.full((B, T - skip, Wd), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    _lib.check(_lib.load().bsc_enc_bias_layernorm(C.c_void_p(u.data_ptr()), C.c_void_p(bs.data_ptr()), C.c_void_p(ga.data_ptr()),
                                                  C.c_void_p(be.data_ptr()), C.c_void_p(y.data_ptr()), out_f32, B, T, skip, Wd, 1e-6,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ref = torch.nn.functional.layer_norm(u.float()[:, skip:] + bs, (Wd,), ga.float(), be.float(), 1e-6)
    assert torch.isfinite(y.float()).all()
    if out_f32:
        assert torch.equal(y, y.to(torch.bfloat16).float())
    # one bf16 rounding of the result: half an ulp of |ref| (2^-9 relative) plus the f32 noise of the statistics
    assert ((y.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-3).all()
