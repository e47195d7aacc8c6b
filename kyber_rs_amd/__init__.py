"""Importable alias of the `kyber-rs_amd/` package (a hyphen cannot appear in a Python module name)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "kyber-rs_amd")
__path__.insert(0, _real)
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
