from nerf_amd.render import (batchify, run_network, batchify_rays, raw2outputs, render_rays, render,  # noqa: F401
                             render_path)
