"""Import shim: the package lives in the directory `tsar-mvs_amd/` (the name the project layout
prescribes); a hyphen cannot appear in a Python import, so `import tsar_mvs_amd` resolves here and
re-roots the package path onto that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tsar-mvs_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
