from nerf_amd.model import create_nerf  # noqa: F401
