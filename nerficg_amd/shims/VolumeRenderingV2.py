"""Top-level module name of the reference's compiled extension (csrc/binding.cpp:234-250): the 12 raw ops, forwarded to the HIP library.
The reference's own autograd layer (custom_functions.py) runs on top of these."""
from nerficg_amd.VolumeRenderingV2 import (composite_test_fw, composite_train_bw, composite_train_fw, distortion_loss_bw,  # noqa: F401
                                           distortion_loss_fw, morton3D, morton3D_invert, packbits, ray_aabb_intersect,
                                           ray_sphere_intersect, raymarching_test, raymarching_train)

__all__ = ['ray_aabb_intersect', 'ray_sphere_intersect', 'packbits', 'morton3D', 'morton3D_invert', 'raymarching_train', 'raymarching_test',
           'composite_train_fw', 'composite_train_bw', 'composite_test_fw', 'distortion_loss_fw', 'distortion_loss_bw']
