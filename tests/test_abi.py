"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports what include/bscnav.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bscnav.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsc_[a-z_0-9]+)\s*\(", text)) - {"bsc_draw_fn"})


def test_library_exports_every_declared_symbol():
    import bsc_nav_amd
    L = bsc_nav_amd._lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"libbscnav.so does not export {name}"
    assert set(declared) == set(bsc_nav_amd._lib.SIGNATURES), "ctypes signatures out of sync with bscnav.h"
    assert b"gfx950" in L.bsc_version()


def test_no_cpu_fallback():
    import torch
    import bsc_nav_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        bsc_nav_amd.VoxelEngine(48, 64, 128, 0.1, -2.0, 4.4, 16, 16)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bsc-nav_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "__never__", f"{f} mentions the oracle"


def test_config_struct_layout_matches_header():
    """Field order of the ctypes mirror == field order of struct bsc_config."""
    import bsc_nav_amd
    text = open(os.path.join(ROOT, "include", "bscnav.h")).read()
    body = re.search(r"typedef struct bsc_config \{(.*?)\} bsc_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(",")[0:]:
            m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)\s*(\[\d+\])?\s*$", part.strip())
            names.append(m.group(1))
    assert names == [f[0] for f in bsc_nav_amd._lib.BscConfig._fields_]


def test_host_shuffle_matches_numpy_bit_for_bit():
    """bsc_host_shuffled_sample == `idx = arange(n); np.random.shuffle(idx); idx[::rate]` (memory_2.py:747-749): same
    permutation and the same global MT19937 state afterwards (host code only, no GPU needed)."""
    import numpy as np
    from bsc_nav_amd import geometry as G
    for seed in (0, 1, 20250928):
        for n, rate in ((1, 1), (2, 1), (3, 2), (7, 3), (1000, 7), (76800, 1000), (307200, 1000), (65536, 1), (462400, 50)):
            np.random.seed(seed)
            a = G.sample_indices(n, rate)
            sa = np.random.randint(0, 1 << 30, 4)
            np.random.seed(seed)
            b = G.sample_indices_fast(n, rate)
            sb = np.random.randint(0, 1 << 30, 4)
            assert np.array_equal(a, b) and b.dtype == np.int32, (seed, n, rate)
            assert np.array_equal(sa, sb), "NumPy's stream must end where np.random.shuffle leaves it"
    np.random.seed(3)                       # a running stream across frames, crossing many 624-word refills
    a = [G.sample_indices(4801, 13) for _ in range(30)]
    np.random.seed(3)
    b = [G.sample_indices_fast(4801, 13) for _ in range(30)]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_host_shuffle_plain_path_matches_numpy_too():
    """the same with the AVX2 block functions / eight-at-a-time rejection switched off (BSC_HOST_NO_AVX2, read once per process:
    a child process): what a host without AVX2 runs"""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from bsc_nav_amd import geometry as G\n"
            "for seed in (0, 7):\n"
            "    for n, rate in ((1, 1), (9, 2), (1000, 7), (307200, 1000), (65537, 3)):\n"
            "        np.random.seed(seed); a = G.sample_indices(n, rate); sa = np.random.randint(0, 1 << 30, 4)\n"
            "        np.random.seed(seed); b = G.sample_indices_fast(n, rate); sb = np.random.randint(0, 1 << 30, 4)\n"
            "        assert np.array_equal(a, b) and np.array_equal(sa, sb), (seed, n, rate)\n"
            "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BSC_HOST_NO_AVX2="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_host_choice_draws_match_python_random():
    """bsc_host_choice_draws == [random.choice(range(k)) ...] (memory_2.py:352) incl. the state of Python's stream."""
    import ctypes as C
    import random
    import numpy as np
    from bsc_nav_amd import _lib
    lib = _lib.load()
    for seed, k, n in ((0, 10, 1), (1, 10, 5000), (2, 7, 777), (3, 16, 300), (4, 1, 50), (5, 33, 2000), (6, 10, 50000)):
        random.seed(seed)
        want = [random.choice(range(k)) for _ in range(n)]
        tail = [random.random() for _ in range(3)]
        random.seed(seed)
        version, internal, gauss = random.getstate()
        key = np.array(internal[:624], dtype=np.uint32)
        pos = C.c_int32(internal[624])
        out = np.zeros(n, np.uint32)
        _lib.check(lib.bsc_host_choice_draws(key.ctypes.data_as(C.c_void_p), C.byref(pos), k, n, out.ctypes.data_as(C.c_void_p)))
        random.setstate((version, tuple(key.tolist()) + (pos.value,), gauss))
        assert out.tolist() == want, (seed, k, n)
        assert [random.random() for _ in range(3)] == tail


def test_sample_prefetcher_keeps_the_reference_stream():
    """geometry.SamplePrefetcher: the host thread that draws the shuffles of coming frames ahead hands out exactly the arrays
    the sequential calls would have, and close() leaves np.random's stream where they would have left it — however far the
    thread had run ahead."""
    import time
    import numpy as np
    from bsc_nav_amd import geometry as G
    n, rate = 4801, 13
    np.random.seed(11)
    want = [G.sample_indices(n, rate) for _ in range(9)]
    tail = np.random.randint(0, 1 << 30, 4)
    np.random.seed(11)
    pf = G.SamplePrefetcher(n, rate, depth=3)
    got = [pf.next() for _ in range(5)]
    time.sleep(0.05)                         # let the worker fill its queue past what is consumed
    got += [pf.next() for _ in range(4)]
    pf.close()
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    assert np.array_equal(np.random.randint(0, 1 << 30, 4), tail)
    pf2 = G.SamplePrefetcher(n, rate)        # opened and closed without a draw: the stream does not move
    pf2.close()
    np.random.seed(11)
    G.sample_indices(n, rate)
    s1 = np.random.get_state()[2]
    np.random.seed(11)
    pf3 = G.SamplePrefetcher(n, rate)
    pf3.next()
    pf3.close()
    assert np.random.get_state()[2] == s1
