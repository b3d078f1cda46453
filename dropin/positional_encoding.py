from nerf_amd.positional_encoding import *  # noqa: F401,F403
from nerf_amd.positional_encoding import Embedder, get_embedder  # noqa: F401
