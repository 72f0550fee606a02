"""fdgs -- MI355X-native differentiable 4D Gaussian rasterizer (hot path only).

Drop-in for the reference's ``gaussian_renderer.render`` /
``GaussianRasterizer`` (gaussian_renderer/__init__.py:19,
gaussian_renderer/diff_gaussian_rasterization.py:247).  The Python host stays
PyTorch-ROCm; all rasterization arithmetic runs in hand-written HIP kernels for
gfx950 behind the C-ABI declared in ``include/fdgs.h`` (``csrc/libfdgs.so``).
There is no CPU or PyTorch fallback: using the rasterizer without the built
library, or without a GPU, raises.

Import as ``fdgs`` (see ``fdgs/__init__.py`` at the repository root).
"""
__version__ = "0.1.0"
