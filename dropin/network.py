from nerf_amd.network import NeRF  # noqa: F401
