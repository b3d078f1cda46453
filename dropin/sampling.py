from nerf_amd.sampling import sample_pdf  # noqa: F401
