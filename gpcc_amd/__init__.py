"""Import alias: the product package lives in the directory ``gpcc.jl_amd/`` (the name the
project layout prescribes), which is not a valid Python identifier.  ``import gpcc_amd`` loads it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gpcc.jl_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f
