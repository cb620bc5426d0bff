"""Import alias: `import gsr_amd` is the package in `torch-gaussian-splatting-rasterizer_amd/`.

The directory name (fixed by the project layout) is not a Python identifier, so this module installs an
import hook: `gsr_amd` and every `gsr_amd.<sub>` resolve to the very same module objects as
`torch-gaussian-splatting-rasterizer_amd[.<sub>]` — one copy of each module, one dlopen of libgsr.so.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "torch-gaussian-splatting-rasterizer_amd"
_ALIAS = __name__
_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)


class _SameModuleLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _ALIAS and not fullname.startswith(_ALIAS + "."):
            return None
        module = importlib.import_module(_REAL + fullname[len(_ALIAS):])
        return importlib.util.spec_from_loader(fullname, _SameModuleLoader(module))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[_ALIAS] = importlib.import_module(_REAL)
