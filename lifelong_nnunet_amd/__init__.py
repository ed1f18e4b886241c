"""Import alias: ``import lifelong_nnunet_amd`` -> the package directory ``lifelong-nnunet_amd/``.

The product package directory carries the repository's name (with a hyphen), which is not a
valid Python identifier; this stub makes it importable by pointing the package search path at it.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "lifelong-nnunet_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
