"""Counterpart of vanilla_dynamic/model.py: create_dnerf :25-186, save_checkpoint :189-211, load_checkpoint :214-240.
Same argument names, same render_kwargs dictionaries, same checkpoint keys."""
import os

import torch

from ..optim import make_adam
from .network import DNeRF_Deformation, DNeRF_Straightforward
from .positional_encoding import get_embedder, get_time_embedder
from .render import run_network_dnerf
from .utils import device


def _safe_load(path):
    # checkpoints hold tensors, ints, the args dict and Adam state: nothing that needs code execution
    return torch.load(path, map_location=device, weights_only=True)


def create_dnerf(args):
    include_input = not getattr(args, 'no_include_input', False)
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed, input_dims=3, include_input=include_input)
    input_ch_views, embeddirs_fn = 0, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed, input_dims=3,
                                                    include_input=include_input)
    embed_time_fn, input_ch_time = get_time_embedder(args.multires_time, include_input=include_input)

    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]

    def make(D, W):
        if args.network_type == 'straightforward':
            return DNeRF_Straightforward(D=D, W=W, input_ch=input_ch, output_ch=output_ch, skips=skips,
                                         input_ch_views=input_ch_views, input_ch_time=input_ch_time,
                                         use_viewdirs=args.use_viewdirs).to(device)
        m = DNeRF_Deformation(D=D, W=W, D_deform=8, W_deform=256, input_ch=input_ch, output_ch=output_ch,
                              skips=skips, skips_deform=skips, input_ch_views=input_ch_views,
                              input_ch_time=input_ch_time, use_viewdirs=args.use_viewdirs,
                              zero_canonical=args.zero_canonical,
                              use_layernorm=getattr(args, 'use_layernorm', False)).to(device)
        m.embed_fn = embed_fn
        return m

    model = make(args.netdepth, args.netwidth)
    model_fine = None
    if args.N_importance > 0 and (args.network_type == 'straightforward'
                                  or getattr(args, 'use_two_models_for_fine', False)):
        model_fine = make(args.netdepth_fine, args.netwidth_fine)

    grad_vars = list(model.parameters())
    if model_fine is not None:
        grad_vars += list(model_fine.parameters())

    def network_query_fn(inputs, viewdirs, times, network_fn):
        return run_network_dnerf(inputs, viewdirs, times, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                 embed_time_fn=embed_time_fn, netchunk=args.netchunk,
                                 network_type=args.network_type)

    optimizer = make_adam(grad_vars, lr=args.lrate, betas=(0.9, 0.999))   # vanilla_dynamic/model.py:115

    start = 0
    if args.ft_path is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        ckpt_dir = os.path.join(args.basedir, args.expname)
        best_ckpt = os.path.join(ckpt_dir, 'best.tar')
        if os.path.exists(best_ckpt):
            ckpts = [best_ckpt]
            print(f'Using best checkpoint: {best_ckpt}')
        elif os.path.exists(ckpt_dir):
            ckpts = [os.path.join(ckpt_dir, f) for f in sorted(os.listdir(ckpt_dir)) if 'tar' in f]
        else:
            ckpts = []
            print(f'Warning: Checkpoint directory {ckpt_dir} does not exist.')

    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not args.no_reload:
        ckpt_path = ckpts[-1]
        print('Reloading from', ckpt_path)
        ckpt = _safe_load(ckpt_path)
        start = ckpt.get('global_step', 0)
        model.load_state_dict(ckpt.get('network_fn_state_dict', {}), strict=True)
        if model_fine is not None and 'network_fine_state_dict' in ckpt:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])
        try:
            if 'optimizer_state_dict' in ckpt:
                optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        except ValueError as e:
            print(f'Warning: Could not load optimizer state (OK for rendering): {e}')
        print(f'[OK] Checkpoint loaded successfully! (step {start})')

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
        'network_type': args.network_type,
        'ndc': False,
        'lindisp': args.lindisp,
    }
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


def save_checkpoint(path, global_step, network_fn, network_fine, optimizer, args):
    save_dict = {
        'global_step': global_step,
        'network_fn_state_dict': network_fn.state_dict(),
        'optimizer_state_dict': optimizer.state_dict(),
        'args': vars(args),
    }
    if network_fine is not None:
        save_dict['network_fine_state_dict'] = network_fine.state_dict()
    torch.save(save_dict, path)
    print(f'Saved checkpoint at {path}')


def load_checkpoint(path, model, model_fine=None, optimizer=None):
    ckpt = _safe_load(path)
    model.load_state_dict(ckpt['network_fn_state_dict'])
    if model_fine is not None and 'network_fine_state_dict' in ckpt:
        model_fine.load_state_dict(ckpt['network_fine_state_dict'])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return ckpt.get('global_step', 0), ckpt.get('args', {})
