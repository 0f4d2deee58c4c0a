"""HIP-graph replay of the denoiser forward for launch-bound batch sizes.

One DiM-L/2 forward is ~2000 kernel launches (~18 ms of host time): below ~32 latents per GPU the host, not the GPU,
sets the pace of the NFE loop (bench.py: batch 8 -> 426 latents/s eager). `GraphedForward` captures model(x, t, y) once
per input signature into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replays it: every HIP kernel of libdimsum_hip.so
is launched on torch's current stream, which during capture is the capture stream, so the ctypes launches are recorded
like torch's own; the caching allocator gives the capture a private pool. Inputs are copied into static buffers, the
output is returned as a copy (2 small device copies per call). Inference only (no autograd through a replay)."""
import torch


class GraphedForward:
    def __init__(self, fn, warmup=3):
        self.fn, self.warmup, self.graphs = fn, warmup, {}

    @staticmethod
    def _sig(args, kwargs):
        def one(a):
            return (tuple(a.shape), a.dtype, a.device) if torch.is_tensor(a) else ("py", a)
        return tuple(one(a) for a in args) + tuple((k, one(v)) for k, v in sorted(kwargs.items()))

    @torch.no_grad()
    def __call__(self, *args, **kwargs):
        key = self._sig(args, kwargs)
        entry = self.graphs.get(key)
        if entry is None:
            s_args = [a.clone() if torch.is_tensor(a) else a for a in args]
            s_kwargs = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kwargs.items()}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # warm-up off the capture: lazy tables, hipBLASLt workspaces
                for _ in range(self.warmup):
                    self.fn(*s_args, **s_kwargs)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                s_out = self.fn(*s_args, **s_kwargs)
            entry = self.graphs[key] = (graph, s_args, s_kwargs, s_out)
        graph, s_args, s_kwargs, s_out = entry
        for dst, src in zip(s_args, args):
            if torch.is_tensor(dst):
                dst.copy_(src)
        for k, dst in s_kwargs.items():
            if torch.is_tensor(dst):
                dst.copy_(kwargs[k])
        graph.replay()
        return s_out.clone()
