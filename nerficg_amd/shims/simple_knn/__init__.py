"""`from simple_knn import _C` (src/Thirdparty/SimpleKNN.py:17)."""
from nerficg_amd.simple_knn import _C, distCUDA2  # noqa: F401
