"""MI355X-native forward rasterizer for pretrained 3D Gaussian Splatting scenes.

Drop-in for the render path of arnaudstiegler/torch-gaussian-splatting-rasterizer
(`rasterize.py`, `spherical_harmonics.py`, `utils.py`, `data_reader.py`): the same
module and function names live in this package; the per-gaussian work runs in the
hand-written HIP library `csrc/libgsr.so` behind the C ABI of `include/gsr.h`.

The directory name is not a Python identifier; import it through the `gsr_amd`
alias module at the repository root or `importlib.import_module`.
"""
__version__ = "0.1.0"
