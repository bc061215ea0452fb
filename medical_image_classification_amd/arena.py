"""Zero-initialised fp32 scratch for ONE backward pass, from one allocation and one fill launch.

The kernels of this package accumulate their reductions with atomics (weight / bias gradients, split-K partial products, the
per-block `dA | dD | ddelta_bias` buffers), so their outputs start from zero: ~60 `torch.zeros` launches of 4-5 us per MedMamba-T
step.  `zeros(shape, device)` hands out 256-byte aligned slices of one buffer that is sized by the previous pass's demand and
filled once; no region is ever handed out twice and every backward pass starts a FRESH buffer (views that autograd keeps as
`.grad` own a reference to their pass's buffer, so nothing is overwritten behind a tensor that is still alive); the end of a
pass is learnt from the autograd engine's own callback queue.  Outside a backward pass, for requests above `_MAX_ELEMS` and on
the first pass (demand unknown) `torch.zeros` is used: always correct, merely one launch more.
"""
import os

import torch

_ENABLED = os.environ.get("MEDSCAN_GRAD_ARENA", "1") == "1"
_MAX_ELEMS = 1 << 22          # 16 MB: larger fills are bandwidth-bound, not launch-bound, and would pin memory behind a small view
_ALIGN = 64                   # floats


class _Arena:
    """Hands out every region of a zero-filled buffer at most ONCE (correctness never depends on the end-of-pass callback: if a
    pass dies with an exception and the callback is lost, the next pass keeps consuming untouched regions or starts a new buffer)."""
    __slots__ = ("buf", "off", "demand", "last", "armed", "stream")

    def __init__(self):
        self.buf, self.off, self.demand, self.last, self.armed, self.stream = None, 0, 0, 0, False, None

    def _end_of_pass(self):
        self.last = self.demand
        self.buf, self.off, self.demand, self.armed = None, 0, 0, False      # views keep this pass's buffer alive as long as needed

    def take(self, numel, device):
        if not self.armed:
            try:        # only legal while the engine is running a backward pass
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
            except RuntimeError:
                return None
            self.armed = True
        n = (numel + _ALIGN - 1) // _ALIGN * _ALIGN
        stream = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
        if self.buf is not None and stream != self.stream:
            return None               # the buffer was filled in another stream's order (two-stream blocks): plain torch.zeros
        self.demand += n
        if self.buf is None or self.off + n > self.buf.numel():
            if self.last <= 0:
                return None                                       # first pass: demand unknown
            self.buf = torch.zeros(max(self.last, n), device=device, dtype=torch.float32)
            self.off, self.stream = 0, stream
        v = self.buf[self.off:self.off + numel]
        self.off += n
        return v


_ARENAS = {}


def zeros(shape, device, dtype=torch.float32):
    """`torch.zeros(shape, device=device, dtype=float32)`, from the current backward pass's arena when there is one."""
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    numel = 1
    for s in shape:
        numel *= s
    if _ENABLED and dtype == torch.float32 and 0 < numel <= _MAX_ELEMS and device.type == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        a = _ARENAS.get(idx)
        if a is None:
            a = _ARENAS[idx] = _Arena()
        v = a.take(numel, device)
        if v is not None:
            return v.view(shape)
    return torch.zeros(shape, device=device, dtype=dtype)


def zeros_like(t):
    return zeros(t.shape, t.device, t.dtype)
