from nerf_amd.rays import get_rays, get_rays_np, ndc_rays  # noqa: F401
