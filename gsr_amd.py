"""Import alias: `import gsr_amd` == the package in `torch-gaussian-splatting-rasterizer_amd/`."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("torch-gaussian-splatting-rasterizer_amd")
sys.modules[__name__] = _pkg
