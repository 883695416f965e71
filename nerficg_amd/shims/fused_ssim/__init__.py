from nerficg_amd.fused_ssim import fused_ssim  # noqa: F401
