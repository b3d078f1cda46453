"""D-NeRF stack -- counterpart of the reference's vanilla_dynamic/ package."""
from .positional_encoding import Embedder, get_embedder, get_time_embedder  # noqa: F401
from .network import NeRF, DNeRF_Straightforward, DNeRF_Deformation  # noqa: F401
from .render import (batchify, run_network_dnerf, raw2outputs, sample_pdf, render_rays_dnerf,  # noqa: F401
                     batchify_rays_dnerf, get_rays, render_dnerf, render_path_dnerf)
from .model import create_dnerf, save_checkpoint, load_checkpoint  # noqa: F401
