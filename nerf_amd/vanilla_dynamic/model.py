"""Counterpart of vanilla_dynamic/model.py: create_dnerf :25-186, save_checkpoint :189-211, load_checkpoint :214-240.
Same argument names, same render_kwargs dictionaries, same checkpoint keys."""
import torch

from .. import _setup
from ..optim import make_adam
from .network import DNeRF_Deformation, DNeRF_Straightforward
from .positional_encoding import get_embedder, get_time_embedder
from .render import run_network_dnerf
from .utils import device


def _safe_load(path):
    # checkpoints hold tensors, ints, the args dict and Adam state: nothing that needs code execution
    return torch.load(path, map_location=device, weights_only=True)


def create_dnerf(args):
    """-> (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer), vanilla_dynamic/model.py:25-186."""
    with_input = not getattr(args, 'no_include_input', False)
    embed_fn, pos_ch = get_embedder(args.multires, args.i_embed, input_dims=3, include_input=with_input)
    embeddirs_fn, dir_ch = (get_embedder(args.multires_views, args.i_embed, input_dims=3, include_input=with_input)
                            if args.use_viewdirs else (None, 0))
    embed_time_fn, time_ch = get_time_embedder(args.multires_time, include_input=with_input)
    two_pass = args.N_importance > 0
    straight = args.network_type == 'straightforward'
    common = dict(input_ch=pos_ch, input_ch_views=dir_ch, input_ch_time=time_ch, output_ch=5 if two_pass else 4,
                  skips=[4], use_viewdirs=args.use_viewdirs)

    def net(depth, width):
        if straight:
            return DNeRF_Straightforward(D=depth, W=width, **common).to(device)
        # the deformation net's size is fixed at 8 x 256 and its skips follow the canonical net's (model.py:75-77)
        m = DNeRF_Deformation(D=depth, W=width, D_deform=8, W_deform=256, skips_deform=common['skips'],
                              zero_canonical=args.zero_canonical, use_layernorm=getattr(args, 'use_layernorm', False),
                              **common).to(device)
        m.embed_fn = embed_fn
        return m

    coarse = net(args.netdepth, args.netwidth)
    # one deformation model serves both passes unless asked otherwise (model.py:93-104)
    own_fine = two_pass and (straight or getattr(args, 'use_two_models_for_fine', False))
    fine = net(args.netdepth_fine, args.netwidth_fine) if own_fine else None
    _setup.apply_precision(args, coarse, fine)
    grad_vars = [p for m in (coarse, fine) if m is not None for p in m.parameters()]

    def network_query_fn(inputs, viewdirs, times, network_fn):
        return run_network_dnerf(inputs, viewdirs, times, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                 embed_time_fn=embed_time_fn, netchunk=args.netchunk,
                                 network_type=args.network_type)

    optimizer = make_adam(grad_vars, lr=args.lrate, betas=(0.9, 0.999))   # vanilla_dynamic/model.py:115
    start = _setup.resume(args, coarse, fine, optimizer, device, prefer_best=True, lenient=True)
    train, test = _setup.render_kwargs(args, network_query_fn, coarse, fine, network_type=args.network_type,
                                       ndc=False, lindisp=args.lindisp)
    return train, test, start, grad_vars, optimizer


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
