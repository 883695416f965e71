"""Only the part of NVIDIA apex the reference imports: apex.optimizers.FusedAdam."""
