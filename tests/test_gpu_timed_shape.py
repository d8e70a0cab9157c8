"""Parity AT THE SHAPES bench.py TIMES (VERDICT r5 "what's weak" (i)): the exact configuration of the headline and of the side
legs — one every-pixel call of the bench's own synthetic frames, f32 token rows, device alpha — against the sequential oracle.
No environment switches: the 8-byte point records, the hot-segment register tiles of the rgb chain (>= 32768 points of one voxel
in one call), the run-order checkpoints, the in-tree radix sorts and the two-pass dense reduce (token tile beyond the 256 MB
MALL) are reached because the data reaches them.  Reference lines reproduced: /root/reference memory_2.py:842-903.

  (i)   BASELINE configs[1]: ONE 768-frame call, 640x480, 14x14x768 f32 tokens, 256^3 grid, "room"
  (ii)  BASELINE configs[2] per GPU: ONE 128-frame call, 16x16x1024 f32 tokens, 512^3 grid
  (iii) the one-voxel-per-point regime: ONE 32-frame "iid" call

Bar: voxel ids / positions / counts / top-down map bit-exact; with the reference's own alpha expression evaluated on the host
(alpha_source="host", what feature_mode="exact" uses) rgb bytes and weights bit-exact too; with the device's exp (what the bench
times) fewer than 1e-3 of the rgb bytes differ, by one, and weights agree to 1e-6 relative; per-voxel means within 1e-3.
The C oracle does ~14 frames/s on one core; alpha for the host-alpha build comes from a thread pool."""
import concurrent.futures as cf
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(H, W, g, D, gs, half, F, kind, seed, vcap, host_alpha_too=True):
    import torch
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic
    from oracle import oracle as orc
    t0 = time.time()
    poses = synthetic.make_poses(kind, 1000, F)
    rgb_d, depth_d, _ = synthetic.make_frames(seed, F, H, W, kind, poses=poses)        # the bench's generator: RGBA u8, f32 depth
    gen = torch.Generator(device="cuda").manual_seed(seed)
    tok_d = torch.randn((F, g, g, D), device="cuda", generator=gen)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    depth, rgb, tok = depth_d.cpu().numpy(), rgb_d.cpu().numpy(), tok_d.cpu().numpy()
    oc = orc.make_config(H, W, gs, 0.1, -half, half, g, D, mode=1)
    with cf.ThreadPoolExecutor(16) as ex:            # memory_2.py:873-875 alpha = exp(-r2 / 1.2), NumPy's exp on the oracle's r2
        alphas = list(ex.map(lambda f: np.exp(-orc.geometry(oc, depth[f], None, Ts[f])["r2"] / (2 * 0.6)), range(F)))
    om = orc.OracleMemory(oc, voxel_capacity=vcap)
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], None, Ts[f], tok[f], alphas[f])
    t_or = time.time() - t0
    opos, orgb, ow = om.export_rgb()
    omh, ocv = om.export_heightmap()
    oacc, ocnt = om.export_dense()
    del om
    out = {}
    for variant in (("device", "host") if host_alpha_too else ("device",)):
        eng = B.VoxelEngine(H, W, gs, 0.1, -half, half, g, D, mode="mean", voxel_capacity=vcap, max_points=F * H * W)
        al = torch.from_numpy(np.concatenate(alphas)).cuda() if variant == "host" else None
        eng.ingest(depth_d, rgb_d, tok_d, Ts, None, None, al)           # ONE call, as bench.py issues it
        eng.sync()
        k = eng.counters()
        pos, rgbv, w = eng.export_rgb()
        mh, cv = eng.export_heightmap()
        acc, cnt = eng.export_dense()
        eng.close()
        del al
        assert k["max_id"] == len(opos) and np.array_equal(pos, opos), variant                   # ids in first-touch order, positions
        assert np.array_equal(cnt, ocnt) and int(ocnt.astype(np.int64).sum()) == k["points_passed"], variant
        assert np.array_equal(mh, omh) and np.array_equal(cv, ocv), variant                      # top-down map: heights and colours
        c = np.maximum(cnt, 1)[:, None].astype(np.float64)
        np.testing.assert_allclose(acc / c, oacc / c, rtol=1e-3, atol=1e-3)
        if variant == "host":
            assert np.array_equal(rgbv, orgb) and np.array_equal(w, ow)
        else:
            d = np.abs(rgbv.astype(np.int32) - orgb.astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() < 1e-3, (int(d.max()), float((d != 0).mean()))
            np.testing.assert_allclose(w, ow, rtol=1e-6)
        out[variant] = float((rgbv != orgb).mean())
    return dict(voxels=len(opos), longest=int(ocnt.max()), pairs=None, oracle_s=t_or, rgb_mismatch=out)


def test_headline_shape_one_768_frame_call_against_oracle():
    r = _run(480, 640, 14, 768, 256, 12.8, 768, "room", 17, 400_000)
    assert r["voxels"] > 20_000
    assert r["longest"] >= 32768 * 3            # voxels far inside the hot-segment path of the rgb chain (>= 2^15 points in the call)


def test_c3_shape_one_128_frame_call_1024d_grid512_against_oracle():
    r = _run(480, 640, 16, 1024, 512, 25.6, 128, "room", 19, 400_000)
    assert r["voxels"] > 15_000 and r["longest"] >= 32768


def test_iid_one_32_frame_call_against_oracle():
    r = _run(480, 640, 14, 768, 256, 12.8, 32, "iid", 23, 4_000_000, host_alpha_too=False)
    assert r["voxels"] > 500_000 and r["longest"] < 2000        # 9.8e6 points scattered over 6e5 voxels: runs of one, segments of a few points
