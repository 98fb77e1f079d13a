"""Where the lane-level mirror classes keep their state."""
import torch

_device = None


def get():
    global _device
    if _device is None:
        if not torch.cuda.is_available():
            from ._lib import DhtsError
            raise DhtsError("no GPU visible: the dhts operators run on MI355X only (there is no CPU fallback)")
        _device = torch.device("cuda", torch.cuda.current_device())
    return _device


def set(device):
    global _device
    _device = torch.device(device)


def as_f32(x, n=None):
    """Tensor / array / list / scalar -> float32 tensor on the dhts device (keeps autograd history of tensors)."""
    dev = get()
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=torch.float32)
    elif isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], torch.Tensor):
        t = torch.stack([e.to(device=dev, dtype=torch.float32).reshape(()) for e in x])
    else:
        t = torch.as_tensor(x, dtype=torch.float32, device=dev)
    if n is not None and t.dim() == 0:
        t = t.expand(n)
    return t
