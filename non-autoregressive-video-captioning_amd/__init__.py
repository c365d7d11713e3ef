"""nacf_amd -- MI355X-native implementation of the NACF video-captioning hot
path (yangbang18/Non-Autoregressive-Video-Captioning): model forward+backward
and non-autoregressive coarse-to-fine decoding as hand-written HIP kernels
behind a C ABI (libnacf_hip.so), with the reference's `get_model(opt)` /
`Seq2Seq` / `Translator` / `decoding.generate` surface on top.

The directory is called `non-autoregressive-video-captioning_amd`; import it
as `nacf_amd` (see nacf_amd.py at the repo root).
"""
from . import config, opts  # noqa: F401
from .config import Constants  # noqa: F401
from .models import get_model  # noqa: F401
from .models.Translator import Translator  # noqa: F401
from .decoding import generate  # noqa: F401
from .runtime import lib as _lib

__version__ = '0.1.0'


def library_path():
    return _lib.LIB_PATH
