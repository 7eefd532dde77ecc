"""trainers/utils.py of the reference (set_device_from_args :5-11, toggle_grad :14-16)."""
import torch


def set_device_from_args(args):
    use_gpu = torch.cuda.is_available() and not getattr(args, 'no_cuda', False)
    setattr(args, 'device', 'cuda' if use_gpu else 'cpu')


def toggle_grad(model, on_or_off):
    for param in model.parameters():
        param.requires_grad_(on_or_off)
