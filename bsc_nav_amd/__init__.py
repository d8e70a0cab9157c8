"""Import shim: the package sources live in ``bsc-nav_amd/`` (not an importable name)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "bsc-nav_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
