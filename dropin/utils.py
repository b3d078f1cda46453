from nerf_amd.utils import DEBUG, device, img2mse, mse2psnr, to8b  # noqa: F401
