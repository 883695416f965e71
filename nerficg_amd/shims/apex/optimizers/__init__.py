from nerficg_amd.apex_optimizers import FusedAdam  # noqa: F401
