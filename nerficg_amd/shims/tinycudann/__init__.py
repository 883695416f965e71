from nerficg_amd.tinycudann import *  # noqa: F401,F403
from nerficg_amd.tinycudann import __all__  # noqa: F401
