"""Importable alias of the ``gcn-max-cut_amd/`` package directory (a hyphen cannot appear in
an import statement).  Sub-modules resolve inside that directory via ``__path__``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gcn-max-cut_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
