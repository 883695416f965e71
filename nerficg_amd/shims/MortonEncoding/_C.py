from nerficg_amd.MortonEncoding import morton_encode  # noqa: F401
