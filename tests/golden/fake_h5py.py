"""In-memory stand-in for the slice of h5py the memory path touches (memory_2.py:330-354, :611-660).

h5py is not part of this image.  The stand-in keeps h5py's observable contract for that slice:
``File(path, mode)`` as a context manager, ``in``, ``[]``, ``create_group``, ``keys()`` / iteration in HDF5's
native link order (= bytewise name order, so ``grid_19_…`` < ``grid_1_…`` because ``'9' < '_'``),
``create_dataset(name, data=, maxshape=, chunks=)``, dataset ``.shape``, ``.resize``, ``[...]`` get / set.
Files live in a process-wide dict keyed by path.  Shared by the golden generators (which hand it to the
reference as ``h5py``) and by the tests of bsc_nav_amd.store's HDF5 adapter.
"""
import sys
import types

import numpy as np


class Dataset:
    def __init__(self, data, maxshape=None, chunks=None):
        self.a = np.array(data, copy=True)
        self.maxshape, self.chunks = maxshape, chunks

    @property
    def shape(self):
        return self.a.shape

    @property
    def dtype(self):
        return self.a.dtype

    def resize(self, shape):
        if self.maxshape is None:
            raise TypeError("only chunked datasets with a maxshape can be resized")
        new = np.zeros(shape, dtype=self.a.dtype)
        n = min(shape[0], self.a.shape[0])
        new[:n] = self.a[:n]
        self.a = new

    def __setitem__(self, k, v):
        self.a[k] = v

    def __getitem__(self, k):
        return self.a[k]


class Group(dict):
    def create_dataset(self, name, data=None, maxshape=None, chunks=None):
        self[name] = Dataset(data, maxshape, chunks)
        return self[name]


class File:
    _stores = {}

    def __init__(self, path, mode="r"):
        if mode == "w":
            File._stores[path] = {}
        elif mode == "r" and path not in File._stores:
            raise FileNotFoundError(path)
        self.g = File._stores.setdefault(path, {})

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __contains__(self, k):
        return k in self.g

    def __getitem__(self, k):
        return self.g[k]

    def create_group(self, k):
        if k in self.g:
            raise ValueError(f"unable to create group (name already exists): {k}")
        self.g[k] = Group()
        return self.g[k]

    def keys(self):
        return sorted(self.g.keys())  # HDF5 native link order == name order (bytewise)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.g)


def install():
    """Register the stand-in as the importable module ``h5py`` -> the module object."""
    m = types.ModuleType("h5py")
    m.File, m.Group, m.Dataset = File, Group, Dataset
    m.__fake__ = True
    sys.modules["h5py"] = m
    return m


def manifest(path):
    """Layout of a stand-in file: [(group name, [(dataset, shape, dtype str, resizable)])] in iteration order."""
    st = File._stores[path]
    out = []
    for name in sorted(st.keys()):
        out.append((name, sorted((d, tuple(st[name][d].shape), str(st[name][d].dtype), st[name][d].maxshape is not None)
                                 for d in st[name])))
    return out
