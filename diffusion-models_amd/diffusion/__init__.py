"""Drop-in `diffusion` package (schedulers and samplers) backed by libdm_hip kernels."""
