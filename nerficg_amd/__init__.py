"""nerficg_amd -- MI355X (gfx950) native hot path for nerficg's Methods plugins.

Sub-packages mirror the native modules the reference's method packages import:
  VolumeRenderingV2, MortonEncoding, tinycudann (subset), diff_gaussian_rasterization
All of them call libnerficg_hip.so (hand-written HIP, C ABI in include/nerficg_hip.h) through ctypes; there is no
CPU fallback -- importing works without a GPU, calling an op does not.
"""
__version__ = '0.1.0'
