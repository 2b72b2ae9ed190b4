"""Import alias: the package directory is named ``audio-denoising_amd`` (not a valid Python
identifier), so this module makes ``import audio_denoising_amd`` resolve to it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "audio-denoising_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
