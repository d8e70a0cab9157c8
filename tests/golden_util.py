"""Helpers shared by the golden-vector tests: fixture loading and input regeneration."""
import ast
import os

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def cfg_of(z):
    out = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        v = str(v)
        try:
            out[str(k)] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[str(k)] = v
    return out


def tag_tokens(tokens):
    F, g, _, D = tokens.shape
    tokens = tokens.copy()
    tokens[..., 0] = np.arange(F * g * g, dtype=np.float32).reshape(F, g, g)
    return tokens


def ingest_inputs(z):
    """Regenerate the seeded inputs of an ingest fixture and verify their checksum."""
    cfg = cfg_of(z)
    rgb, depth, poses = synth.make_frames(cfg["seed"], cfg["F"], cfg["H"], cfg["W"], cfg["kind"],
                                          start_yaw_steps=cfg.get("yaw0", 0))
    tokens = tag_tokens(synth.make_tokens(cfg["seed"], cfg["F"], cfg["g"], cfg["D"]))
    assert synth.checksum(rgb, depth, poses, tokens) == str(z["input_sha"]), "synthetic input drifted from fixture"
    return cfg, rgb, depth, poses, tokens


INGEST_FIXTURES = ["g2_mini_s1", "g2_mini_s7_yaw", "g2_c1_s50_iid", "g2_c1_s1000", "g3_flush_small_cache",
                   "g3_flush_640x480_s97"]
GEOMETRY_FIXTURES = ["g1_geometry_320x240", "g1_geometry_640x480", "g1_geometry_680x680"]


def query_specs(z):
    i = 0
    while f"q{i}_K" in z:
        r = float(z[f"q{i}_radius"])
        fl = [int(x) for x in z[f"q{i}_floor"]]
        yield dict(i=i, B=int(z[f"q{i}_B"]), T=int(z[f"q{i}_T"]), K=int(z[f"q{i}_K"]),
                   radius=None if r < 0 else r, curr=[int(x) for x in z[f"q{i}_curr"]],
                   floor=None if fl[0] > fl[1] else fl, from_store=int(z[f"q{i}_from_store"]),
                   pooled=z[f"q{i}_pooled"], pos=z[f"q{i}_pos"], sim=z[f"q{i}_sim"], top1=z[f"q{i}_top1"])
        i += 1


def query_tokens(spec, seed, D, store_feats):
    qtok = synth.make_query_tokens(seed + spec["i"], spec["B"], spec["T"], D)
    if spec["from_store"] >= 0:
        row = store_feats[spec["from_store"] % max(1, len(store_feats))]
        base = np.broadcast_to(row, (spec["B"], spec["T"], D)).copy()
        qtok = base + 0.01 * qtok
        qtok = qtok.astype(np.float32)
    return qtok


def assert_topk_matches(pos, sim, ref_pos, ref_sim, tol=2e-6):
    """Top-K must equal the reference's: same scores (within `tol`) and the same positions in the same
    order, except that candidates whose reference scores differ by a non-zero amount below `tol` (float
    summation-order noise) may swap places.  Exact ties must keep the reference's HDF5-name order."""
    pos = np.asarray(pos).reshape(-1, 3)
    ref_pos = np.asarray(ref_pos).reshape(-1, 3)
    ref_sim = np.asarray(ref_sim, np.float64).reshape(-1)
    assert len(pos) == len(ref_pos), (len(pos), len(ref_pos))
    np.testing.assert_allclose(np.asarray(sim, np.float64), ref_sim, rtol=0, atol=tol)
    if np.array_equal(pos, ref_pos):
        return
    i, n = 0, len(ref_pos)
    while i < n:
        j = i + 1
        while j < n and abs(ref_sim[j] - ref_sim[j - 1]) <= tol:
            j += 1
        if j - i == 1 or np.all(ref_sim[i:j] == ref_sim[i]):
            assert np.array_equal(pos[i:j], ref_pos[i:j]), f"top-K rows [{i},{j}) differ"
        elif j < n:   # near-tie group fully inside the top-K: same members
            a = sorted(map(tuple, pos[i:j].tolist()))
            b = sorted(map(tuple, ref_pos[i:j].tolist()))
            assert a == b, f"top-K near-tie group [{i},{j}) differs"
        i = j


def assert_topk_near(pos, sim, ref_pos, ref_sim, tol):
    """Top-K against an independently computed ranking: scores within `tol`; members of every group of reference
    scores closer than `tol` to each other (summation-order noise can reorder them) must agree as a set, unless the
    group touches the K boundary."""
    pos, ref_pos = np.asarray(pos).reshape(-1, 3), np.asarray(ref_pos).reshape(-1, 3)
    sim, ref_sim = np.asarray(sim, np.float64).reshape(-1), np.asarray(ref_sim, np.float64).reshape(-1)
    assert len(pos) == len(ref_pos), (len(pos), len(ref_pos))
    np.testing.assert_allclose(sim, ref_sim, rtol=0, atol=tol)
    i, n = 0, len(ref_pos)
    while i < n:
        j = i + 1
        while j < n and abs(ref_sim[j] - ref_sim[j - 1]) <= tol:
            j += 1
        if j < n or j - i == 1:
            assert sorted(map(tuple, pos[i:j].tolist())) == sorted(map(tuple, ref_pos[i:j].tolist())), f"top-K rows [{i},{j}) differ"
        i = j
