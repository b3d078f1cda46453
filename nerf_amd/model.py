"""Model construction -- counterpart of the reference's model.py:11-107 (create_nerf).

Same wiring: embedders, coarse/fine NeRF, the `network_query_fn` closure over run_network, Adam over both
nets, newest-checkpoint reload, and the train/test kwargs dicts that render() receives.
"""
import os

import torch

from .optim import make_adam

from .network import NeRF
from .positional_encoding import get_embedder
from .render import run_network
from .utils import device


def create_nerf(args):
    """-> (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer)"""
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, input_ch_views = None, 0
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]

    def make(depth, width):
        return NeRF(D=depth, W=width, input_ch=input_ch, output_ch=output_ch, skips=skips,
                    input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)

    model = make(args.netdepth, args.netwidth)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = make(args.netdepth_fine, args.netwidth_fine)
        grad_vars += list(model_fine.parameters())

    def network_query_fn(inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           netchunk=args.netchunk)

    optimizer = make_adam(grad_vars, lr=args.lrate, betas=(0.9, 0.999))   # model.py:52, one launch per step on the GPU

    # checkpoint discovery: explicit --ft_path, else the newest *.tar under basedir/expname (model.py:59-81)
    start = 0
    if args.ft_path is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        ckpt_dir = os.path.join(args.basedir, args.expname)
        ckpts = sorted(os.path.join(ckpt_dir, f) for f in os.listdir(ckpt_dir) if 'tar' in f) \
            if os.path.exists(ckpt_dir) else []
        if not os.path.exists(ckpt_dir):
            print(f'Warning: Checkpoint directory {ckpt_dir} does not exist.')
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not args.no_reload:
        print('Reloading from', ckpts[-1])
        ckpt = torch.load(ckpts[-1], map_location=device)
        start = ckpt['global_step']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(ckpt['network_fn_state_dict'])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])

    render_kwargs_train = {
        'network_query_fn': network_query_fn,
        'perturb': args.perturb,
        'N_importance': args.N_importance,
        'network_fine': model_fine,
        'N_samples': args.N_samples,
        'network_fn': model,
        'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:  # NDC only for forward-facing LLFF data
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp

    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer
