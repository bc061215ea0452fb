"""`optim.Adam(net.parameters(), lr=0.0001)` of the reference's training loops (train.py:62,76; ddp_train.py:137,165) with the
update of ALL parameters in one launch (`ms_adam_multi`, csrc/adam.hip).

`MsAdam` IS a `torch.optim.Adam` (same constructor defaults, same `state_dict()` layout: `step`, `exp_avg`, `exp_avg_sq` per
parameter -- checkpoints written by either load into the other, which the reference's `{epoch, model, optimizer, best_acc}`
checkpoints rely on); only `step()` differs, and only when every parameter with a gradient is a contiguous fp32 CUDA tensor of
one device and the group uses plain Adam (no weight decay, no amsgrad, no maximize, a float learning rate).  Anything else runs
torch's own implementation.  The stable pointers (parameter, exp_avg, exp_avg_sq) and the workgroup -> (tensor, chunk) map are
built once per set of tensors and kept in device memory; the gradient pointers change every step and travel as kernel arguments.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import ADAM_CHUNK, ADAM_MAX_TENSORS, MsAdamDesc


class _Mixed(Exception):
    """parameters of one group with different step counts (some joined later): torch's implementation handles them"""


class MsAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("fused", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._ms_tables = {}          # group index -> (key, [(desc tensor, block tensor, n_blocks, first, count), ...])

    def _ms_ok(self, group, params, grads, exp_avgs, exp_avg_sqs, steps):
        if group["weight_decay"] != 0 or group["amsgrad"] or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if not isinstance(group["lr"], float) or not params:
            return False
        dev = params[0].device
        if dev.type != "cuda":
            return False
        for t in (*params, *grads, *exp_avgs, *exp_avg_sqs):
            if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() or t.is_sparse:
                return False
        return all(s.device.type == "cpu" for s in steps)          # (device-resident counters: capturable / fused state -> torch's path)

    def _ms_table(self, gi, params, exp_avgs, exp_avg_sqs, steps):
        """Device tables for this exact set of tensors + the step count they share (read from the state once per set: as long as the
        same tensors come back, the count is advanced on the host without looking at 355 scalars again)."""
        key = tuple((p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, m, v in zip(params, exp_avgs, exp_avg_sqs))
        cached = self._ms_tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        ts = {float(s) for s in steps}
        if len(ts) != 1:
            raise _Mixed()
        dev = params[0].device
        launches = []
        for first in range(0, len(params), ADAM_MAX_TENSORS):
            sub = key[first:first + ADAM_MAX_TENSORS]
            arr = (MsAdamDesc * len(sub))()
            blocks = []
            for i, (pp, pm, pv, n) in enumerate(sub):
                arr[i].p, arr[i].m, arr[i].v, arr[i].n = pp, pm, pv, n
                blocks.extend((i, c) for c in range((n + ADAM_CHUNK - 1) // ADAM_CHUNK))
            desc = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            bt = torch.tensor(blocks, dtype=torch.int32).reshape(-1, 2).contiguous().to(dev)
            launches.append((desc, bt, len(blocks), first, len(sub)))
        self._ms_tables[gi] = [key, launches, ts.pop()]
        return launches, self._ms_tables[gi][2]

    def load_state_dict(self, state_dict):
        """torch's `Optimizer.load_state_dict` leaves `state['step']` where the checkpoint was mapped (`map_location=device` puts every
        counter on the GPU for non-fused groups); the one-launch path keeps the counters on the host, so they are brought back here --
        otherwise every step after a resume would silently run torch's foreach implementation with ~2 host syncs per parameter."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            s = st.get("step")
            if torch.is_tensor(s) and s.device.type != "cpu":
                st["step"] = s.detach().to("cpu", torch.float32)
        self._ms_tables.clear()

    def step(self, closure=None):
        # the closure first, as torch.optim.Adam.step does: it may (re)create the gradient tensors the lists below refer to
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # first pass: the tensor lists of every group (creates missing state, as torch's step does) and whether the one-launch path
        # serves all of them; if not, torch's own implementation runs the whole step
        work = []
        with torch.no_grad():
            for gi, group in enumerate(self.param_groups):
                params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
                self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
                if not params:
                    continue
                try:
                    ok = self._ms_ok(group, params, grads, exp_avgs, exp_avg_sqs, steps)
                    launches, t = self._ms_table(gi, params, exp_avgs, exp_avg_sqs, steps) if ok else (None, 0)
                except _Mixed:
                    ok = False
                if not ok:
                    self._ms_tables.clear()              # torch's step advances the counters: the cached count would go stale
                    super().step(None)                   # (the closure has already been evaluated)
                    return loss
                work.append((gi, group, params, grads, launches, t, steps))
        with torch.no_grad():
            for gi, group, params, grads, launches, t, steps in work:
                torch._foreach_add_(steps, 1)            # host scalars (one C++ loop); every parameter keeps its own, as torch.optim.Adam does
                t += 1
                self._ms_tables[gi][2] = t
                beta1, beta2 = group["betas"]
                bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
                step_size, bc2_sqrt = group["lr"] / bc1, math.sqrt(bc2)
                lib, dev = _lib.lib(), params[0].device
                stream = _lib.current_stream_ptr(dev)
                with _lib.on_device(dev):
                    for desc, bt, n_blocks, first, count in launches:
                        gp = (ctypes.c_void_p * count)(*[g.data_ptr() for g in grads[first:first + count]])
                        _lib.check(lib.ms_adam_multi(desc.data_ptr(), bt.data_ptr(), n_blocks, gp, count, step_size, bc2_sqrt,
                                                     1.0 - beta1, beta2, 1.0 - beta2, group["eps"], stream), "ms_adam_multi")
        return loss
