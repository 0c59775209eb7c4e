"""Importable alias of the `bevy-hikari_amd/` package directory (a hyphen is not a valid module
name).  Everything lives in ../bevy-hikari_amd; this shim only redirects the package path."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "bevy-hikari_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
