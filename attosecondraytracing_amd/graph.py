"""HIP-graph capture of launch-bound work.

Every entry point of libart_hip.so is a plain asynchronous kernel launch on the caller's stream with its descriptors
passed by value, and the host shell allocates through PyTorch's caching allocator, so a whole step -- e.g.
`RayTracingCalculation` + `Detector.readout(sync=False)` -- can be captured once into a HIP graph and replayed.  For
small bundles, where a step is a few tens of microseconds of GPU work behind ~100 us of Python and launch overhead,
that is the difference between host-bound and GPU-bound (1e4 rays x 4 toroids + read-out: 250 -> 26 us per step;
from 1e6 rays on the step is GPU-bound either way).

What a replay re-executes is fixed at capture time: the kernels, the element and detector descriptors (poses baked
in), and the ADDRESSES of inputs and outputs.  To trace other rays through the same scene, overwrite the captured
source bundle's tensors in place (`src.data.copy_(...)`) and replay; the results appear in the tensors of the
returned outputs.  Anything that synchronises with the host (`len(bundle)`, `readout(sync=True)`, `.cpu()`) must stay
outside the captured function."""
import torch


class CapturedStep:
    """`CapturedStep(fn)` runs `fn()` a few times on a side stream (allocator warm-up), captures one more call into a
    graph, and keeps what it returned; `replay()` re-executes the captured launches and returns those same objects."""

    def __init__(self, fn, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs
