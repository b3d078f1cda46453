"""Model construction -- counterpart of the reference's model.py:11-107 (create_nerf): embedders, coarse / fine NeRF,
the `network_query_fn` closure over run_network, one optimizer over both nets, resume from the newest checkpoint, and
the train / test keyword sets render() receives.  The shared steps live in nerf_amd/_setup.py."""
from . import _setup
from .network import NeRF
from .optim import make_adam
from .positional_encoding import get_embedder
from .render import run_network
from .utils import device


def create_nerf(args):
    """-> (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer)"""
    embed_fn, pos_ch = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, dir_ch = (get_embedder(args.multires_views, args.i_embed) if args.use_viewdirs else (None, 0))
    two_pass = args.N_importance > 0

    def net(depth, width):   # output_ch 5 with a fine pass (model.py:30), skip after layer 4 (model.py:31)
        return NeRF(D=depth, W=width, input_ch=pos_ch, input_ch_views=dir_ch, output_ch=5 if two_pass else 4,
                    skips=[4], use_viewdirs=args.use_viewdirs).to(device)

    coarse = net(args.netdepth, args.netwidth)
    fine = net(args.netdepth_fine, args.netwidth_fine) if two_pass else None
    _setup.apply_precision(args, coarse, fine)
    grad_vars = [p for m in (coarse, fine) if m is not None for p in m.parameters()]

    def network_query_fn(inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           netchunk=args.netchunk)

    optimizer = make_adam(grad_vars, lr=args.lrate, betas=(0.9, 0.999))   # model.py:52; one launch per step
    start = _setup.resume(args, coarse, fine, optimizer, device)
    extra = {}
    if args.dataset_type != 'llff' or args.no_ndc:   # NDC is for forward-facing LLFF scenes only (model.py:97-101)
        print('Not ndc!')
        extra = dict(ndc=False, lindisp=args.lindisp)
    train, test = _setup.render_kwargs(args, network_query_fn, coarse, fine, **extra)
    return train, test, start, grad_vars, optimizer
