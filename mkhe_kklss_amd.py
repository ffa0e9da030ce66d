"""Import shim: the package directory is named `mkhe-kklss_amd/` (not a valid Python identifier),
so this module exposes it as the importable package `mkhe_kklss_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mkhe-kklss_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
