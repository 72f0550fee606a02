"""Importable alias for the package directory ``4d-gaussian-splatting_amd/``.

The package directory name required by the repository layout is not a valid
Python identifier, so ``import fdgs`` maps onto it: this module's ``__path__``
points at that directory and its ``__init__.py`` is executed in this namespace.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "4d-gaussian-splatting_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
