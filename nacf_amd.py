"""Import alias: the package directory name mandated for this repo
(`non-autoregressive-video-captioning_amd`) is not a Python identifier, so
`import nacf_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "non-autoregressive-video-captioning_amd")
_spec = importlib.util.spec_from_file_location("nacf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nacf_amd"] = _mod
_spec.loader.exec_module(_mod)
