"""Small-batch inference: the eval() forward of a model of this package captured ONCE in a HIP graph and replayed.

The reference validates / predicts eagerly (train.py:82-95: `net.eval()`, `torch.no_grad()`, one `net(images)` per batch).  At small batch
that forward is ~250 kernel launches of a few microseconds each: the host, not the GPU, sets the latency (4.8 ms per MedMamba-T forward at
batch 1..16 on MI355X).  Every kernel of this package is launched on torch's CURRENT stream through the C ABI, so `torch.cuda.CUDAGraph`
(a hipGraph on ROCm) records them like torch's own; a replay costs the GPU time only -- 1.6 ms at batch 1 together with the segmented scan
(`MsScanParams.segments`, ss2d_fused._scan_segments), 2.9 ms at batch 16 (tools/bench_infer.py, profiles/r03_infer_bench.txt).

    fwd = GraphedForward(net, example_images)        # net.eval() is applied; shapes / dtypes are fixed by `example_images`
    logits = fwd(images)                             # a tensor that the NEXT call overwrites: clone it to keep it

Inference only: the graph holds the weights' addresses (in-place weight updates are seen; re-allocated parameters are not) and the bf16
working copies of the weights as they were at capture (call `recapture()` after the weights changed).
"""
import torch

__all__ = ["GraphedForward"]


class GraphedForward:
    def __init__(self, model, example, autocast_dtype=torch.bfloat16, warmup=2):
        if not (isinstance(example, torch.Tensor) and example.is_cuda):
            raise RuntimeError("GraphedForward: a CUDA (HIP) example input is required -- there is no CPU path")
        self.model = model.eval()
        self.autocast_dtype = autocast_dtype
        self.static_in = example.detach().clone()
        self.warmup = max(1, int(warmup))
        self.graph = self.static_out = None
        self.recapture()

    def _run(self):
        with torch.no_grad():
            if self.autocast_dtype is not None:
                with torch.autocast("cuda", dtype=self.autocast_dtype):
                    return self.model(self.static_in)
            return self.model(self.static_in)

    def recapture(self):
        """(Re)record the graph: after the weights changed (their bf16 working copies are part of the recording)."""
        dev = self.static_in.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):                 # lazy initialisations (library load, weight copies, allocator pools) happen outside the capture
                self._run()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out = self._run()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph, self.static_out = graph, out

    def __call__(self, x):
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype or x.device != self.static_in.device:
            raise RuntimeError(f"GraphedForward: the graph was captured for {tuple(self.static_in.shape)} {self.static_in.dtype} on "
                               f"{self.static_in.device}, got {tuple(x.shape)} {x.dtype} on {x.device}")
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out
