"""bf16 working copies ("shadows") of fp32 master parameters, refreshed for the whole model by ONE launch per step.

Under `torch.autocast(bfloat16)` every weight that feeds a MIOpen convolution or a library GEMM is cast per use: one
`aten::_to_copy` launch per weight per forward (~90 of 4-5 us each for MedMamba-T, MedMamba.py:517-527, 284, 326) plus a layout
copy per convolution weight on the channels_last path.  `bf16(p)` returns a cached bf16 copy instead; the first request that
finds its copy stale refreshes ALL registered copies of that device with `ms_cast_bf16_multi` (one grid over a descriptor table
in device memory; convolution weights are written in channels_last order directly).

A copy is stale when the parameter's version counter or data pointer changed (`load_state_dict`, in-place updates under
no_grad, `.to()` bump / change them) or when `invalidate()` was called since.  The version counter alone is NOT enough: fused
optimizers (`torch.optim.Adam(fused=True)`) and writes through `p.data` do not bump it -- so every autograd.Function that uses
a copy calls `invalidate()` from its backward, a process-wide optimizer post-step hook does the same after EVERY
`Optimizer.step()`, and VSSM.forward does so once per training step as well.  The copies carry no autograd history: the caller's autograd.Function returns the gradient for the
fp32 master itself.  MEDSCAN_BF16_SHADOWS=0 turns the cache off (every request casts).
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import MsCastDesc

_ENABLED = os.environ.get("MEDSCAN_BF16_SHADOWS", "1") == "1"


class _Shadow:
    __slots__ = ("t", "version", "ptr", "conv", "epoch", "owner")

    def __init__(self, t, conv, owner):
        self.t, self.version, self.ptr, self.conv, self.epoch, self.owner = t, -1, 0, conv, -1, owner


_BY_ID = {}                       # (id(parameter), kind) -> _Shadow (entries removed when the parameter dies: nothing is stored ON the
                                  # parameter, so pickling / deepcopy of a model never sees the cache).  kind: False = plain copy,
                                  # True = convolution weight in channels_last order, "flip" = the input-gradient convolution's weight


class _Registry:
    def __init__(self):
        self.params = []          # ids, registration order
        self.epoch = 0
        self.table = None         # (key, device tensor holding the descriptors, device block table, n_blocks)
        self.stream = None        # raw handle of the stream the last whole-model refresh was launched on
        self.event = None         # recorded behind that launch: a request from ANOTHER stream waits for it (two-stream blocks)


_REG = {}


def _registry(idx):
    r = _REG.get(idx)
    if r is None:
        r = _REG[idx] = _Registry()
    return r


def invalidate(device=None):
    """Mark every copy (of one device, or of all) stale: the next request refreshes them all in one launch."""
    for idx, r in _REG.items():
        if device is None or device.index is None or device.index == idx:
            r.epoch += 1


# Every optimizer step of the process invalidates the copies too (fused optimizers update parameters without bumping their
# version counters): together with the invalidation from the backward of every Function that used a copy, no training loop can
# see a stale weight, whatever its order of forward / backward / step.
try:
    from torch.optim.optimizer import register_optimizer_step_post_hook
    register_optimizer_step_post_hook(lambda _opt, _args, _kwargs: invalidate())
except Exception:          # pragma: no cover  (an old torch without global optimizer hooks: the backward invalidation remains)
    pass


def _stale(p, sh, epoch):
    return sh.version != p._version or sh.ptr != p.data_ptr() or sh.epoch != epoch


def _refresh(reg, device, only=None):
    """Refresh every stale copy of `device` in one launch -- or just `only` = (p, shadow) (a request from inside a backward pass)."""
    live, todo, seen = [], [], set()
    if only is not None:
        todo = [only]
    else:
        for pid in reg.params:
            sh = _BY_ID.get(pid)
            p = sh.owner() if sh is not None else None
            if p is None or pid in seen:                # dead, or an id the interpreter has recycled for a newer parameter
                continue
            seen.add(pid)
            live.append(pid)
            if p.is_cuda and p.device == device and sh.t.device == device and _stale(p, sh, reg.epoch):
                todo.append((p, sh))
        reg.params = live
    if not todo:
        return
    # the cached descriptor table is valid only for EXACTLY these tensors: pointers alone are not enough -- the caching allocator
    # hands a freed parameter's address to the next model, whose tensor of another size would then be cast with the old extent
    key = tuple((p.data_ptr(), sh.t.data_ptr(), p.numel(), (tuple(p.shape), sh.conv) if sh.conv else 0) for p, sh in todo)
    if only is not None or reg.table is None or reg.table[0] != key:
        arr = (MsCastDesc * len(todo))()
        blocks = []
        for i, (p, sh) in enumerate(todo):
            taps = p.shape[2] * p.shape[3] if (sh.conv and p.dim() == 4) else 1
            arr[i].src, arr[i].dst, arr[i].n = p.data_ptr(), sh.t.data_ptr(), p.numel()
            if sh.conv == "flip":
                arr[i].inner, arr[i].taps = p.shape[1], -taps
            else:
                arr[i].inner, arr[i].taps = (p.shape[1] if taps > 1 else 1), taps
            if 2 <= taps <= _lib.CAST_TILE_MAX_TAPS:        # convolution weights: tiles of 64 output x 16 input channels
                pieces = -(-p.shape[0] // _lib.CAST_TILE_O) * -(-p.shape[1] // _lib.CAST_TILE_I)
            else:
                pieces = -(-p.numel() // _lib.CAST_CHUNK)
            blocks.extend((i, c) for c in range(pieces))
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        bt = torch.tensor(blocks, dtype=torch.int32).reshape(-1, 2).contiguous()
        table = (key, host.to(device), bt.to(device), len(blocks))
        if only is None:
            reg.table = table                      # (a one-off table is not worth caching over the whole-model one)
    else:
        table = reg.table
    _, tab, bt, n_blocks = table
    with _lib.on_device(device):
        _lib.check(_lib.lib().ms_cast_bf16_multi(tab.data_ptr(), bt.data_ptr(), n_blocks, _lib.current_stream_ptr(device)), "ms_cast_bf16_multi")
    if only is None:
        reg.stream = _lib.current_stream_ptr(device).value
        if reg.event is None:
            reg.event = torch.cuda.Event()
        reg.event.record(torch.cuda.current_stream(device))
    for p, sh in todo:
        sh.version, sh.ptr, sh.epoch = p._version, p.data_ptr(), reg.epoch


def bf16(p, conv=False, in_backward=False):
    """bf16 copy of the fp32 CUDA tensor `p` (a Parameter).  conv=True: `p` is a (O, I, kh, kw) convolution weight and the copy has
    channels_last strides (what the NHWC convolution kernels read); conv="flip": the weight of the INPUT-GRADIENT convolution,
    (I, kh, kw, O) memory with the taps flipped.  No autograd history.
    in_backward=True (a request from a Function's backward): the invalidations that the running backward pass has already issued
    for the NEXT forward are ignored -- the weights have not changed yet -- so a copy refreshed during this step's forward is
    served as is (otherwise every such request would refresh the whole model again)."""
    if not (_ENABLED and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
        t = p.detach().to(torch.bfloat16)
        if conv == "flip":
            return t.flip(2, 3).permute(1, 2, 3, 0).contiguous()
        return t.contiguous(memory_format=torch.channels_last) if (conv and p.dim() == 4) else t
    reg = _registry(p.device.index)
    pid = (id(p), conv)
    sh = _BY_ID.get(pid)
    if sh is None or sh.owner() is not p or sh.t.numel() != p.numel() or sh.t.device != p.device:
        if conv == "flip":             # (Ci, kh, kw, Co) memory: the weight of the input-gradient convolution
            t = torch.empty((p.shape[1], p.shape[2], p.shape[3], p.shape[0]), device=p.device, dtype=torch.bfloat16)
        else:
            fmt = torch.channels_last if (conv and p.dim() == 4) else torch.contiguous_format
            t = torch.empty(p.shape, device=p.device, dtype=torch.bfloat16, memory_format=fmt)
        reg.params.append(pid)
        sh = _BY_ID[pid] = _Shadow(t, conv, weakref.ref(p, lambda _r, pid=pid: _BY_ID.pop(pid, None)))
    if in_backward:
        if sh.version != p._version or sh.ptr != p.data_ptr() or sh.epoch < 0:
            _refresh(reg, p.device, only=(p, sh))
            sh.epoch = reg.epoch - 1               # current for THIS backward, still due for the next forward's refresh
    elif _stale(p, sh, reg.epoch):
        _refresh(reg, p.device)
    elif reg.event is not None and _lib.current_stream_ptr(p.device).value != reg.stream:
        # the epoch bookkeeping is host-side: a copy that is "fresh" may have been refreshed by a launch on another stream (the conv
        # branch's side stream made the step's first request) that this stream's kernels are not ordered behind
        torch.cuda.current_stream(p.device).wait_event(reg.event)
    return sh.t
