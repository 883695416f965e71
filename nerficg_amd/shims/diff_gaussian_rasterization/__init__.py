from nerficg_amd.diff_gaussian_rasterization import *  # noqa: F401,F403
from nerficg_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
