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


class SceneProgram:
    """A compiled scene: device-resident scene table (include/art_hip.h: art_scene_pack / art_trace_scene) +
    preallocated per-element output bundles + a captured HIP graph of the launch(es).

        prog = SceneProgram(sources, element_lists)      # one launch for all chains, captured
        outs = prog.run()                                 # replay -> [chain][element] bundles (always the same objects)
        prog.update(new_element_lists)                    # same optics, new poses: rewrites the table in place
        outs = prog.run()

    The launch reads nothing but the table and the bundles, so a re-trace of a modified scene costs one small
    host-to-device copy and one graph launch -- no descriptor marshalling, no allocation, no per-launch Python.  This is
    what a pose scan (the misalignment loop lists of ART/ModuleOpticalChain.py:371-657, an alignment optimiser) or a
    repeated trace of small bundles (1e4-1e6 rays, where an eager launch is host-bound) should use.  To trace other
    rays, overwrite the source bundles' tensors in place.  `detectors`: one placed Detector per chain whose read-out
    of the chain's last bundle is fused behind the trace (art_trace_chain_readout; results in `self.readouts`, and
    `detector.readout(outputs[c][-1])` returns them).  `post`: optional callable `post(outputs)` captured right behind
    the trace; its return value is `self.post_result`.

    `readout_lite=True`: the fused read-outs reduce only count, sum of paths, bounding box and path range (ArtChainReadout.lite).
    `readout_targets`: per chain None or (X, Y, opl) tensors of the caller the fused read-out writes into.
    `outputs`: [chain][element] bundles of the caller to trace into, instead of bundles of the program's own.
    `placement_tries` (default: ART_PLACEMENT_TRIES, else 1 = off): opt-in look at where the output bundles lie, see
    `_tune_placement` below.

    Results are bit-identical to `RayTracingCalculation`; the returned bundles are overwritten by the next `run()`."""

    def __init__(self, sources, element_lists, IgnoreDefects=True, post=None, capture=True, detectors=None, history=True,
                 placement_tries=None, readout_lite=False, readout_targets=None, outputs=None):
        from . import ModuleProcessing as mp
        from . import _abi
        from .bundle import RayBundle
        self._mp, self._abi = mp, _abi
        self.sources = [mp._as_bundle(s) for s in sources]
        self.c, self.m = len(self.sources), len(element_lists[0])
        self.n = self.sources[0].n_slots
        self.be = self.sources[0].backend
        self.IgnoreDefects = bool(IgnoreDefects)
        if self.c == 0 or self.m == 0 or self.n == 0:
            raise ValueError("SceneProgram needs at least one chain, one element and one ray")
        if any(len(e) != self.m for e in element_lists) or any(s.n_slots != self.n for s in self.sources):
            raise ValueError("all chains of a SceneProgram share the element count and the ray count")
        # history=False: only every chain's LAST bundle is written (the others are None) -- what a caller that analyses the
        # final bundle needs (ARTmain's lazy history); chains of at most 8 elements (one fused launch, no hand-over bundle)
        if not history and self.m > 8:
            raise ValueError("SceneProgram(history=False) covers chains of at most 8 elements")
        self._history = bool(history)
        self._views_in = [s.view() for s in self.sources]
        # outputs: [chain][element] bundles of the caller to trace into (None entries where history=False) instead of bundles
        # of the program's own -- e.g. ranges (RayBundle.slots) of one set of bundles shared by the tiles of a tiled step
        self._bind(outputs if outputs is not None else self._alloc_outputs())
        self._readout_lite = bool(readout_lite)
        # readout_targets: per chain None or (X, Y, opl) caller-owned tensors the fused read-out writes into -- e.g. the
        # dense sections of a survivor send buffer (sharding.SurvivorGather.acquire: zero-copy gather)
        self._ro_targets = list(readout_targets) if readout_targets is not None else None
        self.detectors, self.readouts = None, None
        if detectors is not None and self.n <= self.be.MAX_FUSED_READOUT_RAYS:
            if len(detectors) != self.c:
                raise ValueError("need one detector per chain")
            self._ro_scratch = self.be.chain_readout_scratch(self.n, self.c)
            self.set_detectors(detectors)
        self.host, self.dev = self.be.scene_alloc(self.c, self.m)
        self._uploaded = None
        self._signature = None
        self._stepwise = None
        self.post, self.post_result = post, None
        self.update(element_lists)
        self.placement = None
        if placement_tries is None:
            import os
            placement_tries = int(os.environ.get("ART_PLACEMENT_TRIES", "1"))
        if self.be.name == "hip" and placement_tries > 1 and not self._stepwise:
            self._tune_placement(element_lists, int(placement_tries))
        self.graph = None
        if capture and self.be.name == "hip" and not self._stepwise:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._launch()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            counted = getattr(self.be, "counted_launches", 0)
            with torch.cuda.graph(self.graph):
                self._launch()
            if hasattr(self.be, "counted_launches"):
                self.be.counted_launches = counted      # a captured launch has not run

    # ---- where the output bundles lie ------------------------------------------------------------------------------
    # The same launch into another allocation of the same size can take a few percent longer or shorter (what differs is
    # how the driver mapped the block, not the memory: profiles/HISTORY.md, "placement").  A program keeps its output
    # bundles for its lifetime, so it MAY look: `SceneProgram(..., placement_tries=N)` or ART_PLACEMENT_TRIES=N allocates
    # up to N candidate blocks, times its own launch into each and keeps the fastest.  OFF by default (N = 1): under the
    # driver's protocol the look chose the first block and the spread was 5 % (BENCH_r03).  When on, it is bounded: the
    # candidates and the spacers between them together stay below ART_PLACEMENT_MEM_CAP (bytes, or a fraction < 1 of the
    # free memory; default 0.25), an allocation that fails ends the look with the candidates found so far, a first pass
    # whose spread is below 3 % ends it without a second pass, and the library never empties the caching allocator: the
    # blocks of the candidates it drops go back to PyTorch's cache, where the caller's next allocation finds them.
    def _alloc_outputs(self):
        from .bundle import RayBundle
        if self._history:
            return RayBundle.allocate_grid(self.n, self.c, self.m, self.sources, self.be)
        return [[None] * (self.m - 1) + [RayBundle.allocate(self.n, like=s, backend=self.be)] for s in self.sources]

    def _bind(self, outputs):
        self.outputs = outputs
        for ci, outs in enumerate(outputs):
            prev = self.sources[ci]
            for b in outs:
                if b is not None:
                    b.parent = prev
                    prev = b
        self._views_out = [b.view() if b is not None else self._abi.ArtBundleView() for outs in outputs for b in outs]

    def _time_launch(self, reps=2):
        seg = -(-self.m // 8)
        self.be.trace_scene(self.dev, self.host, self.n, segments=seg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.be.trace_scene(self.dev, self.host, self.n, segments=seg)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    @staticmethod
    def _placement_budget(free):
        import os
        cap = float(os.environ.get("ART_PLACEMENT_MEM_CAP", "0.25"))
        return int(cap * free) if cap < 1.0 else int(min(cap, free))

    def _tune_placement(self, element_lists, tries):
        import time
        rows = sum(b is not None for outs in self.outputs for b in outs)
        nbytes = rows * 65 * self.n
        if nbytes < (64 << 20) or tries <= 1:
            return
        free, _ = torch.cuda.mem_get_info()
        budget = self._placement_budget(free)
        asked, tries = tries, max(1, min(tries, 1 + budget // max(nbytes, 1)))     # the candidates exist side by side
        if tries <= 1:
            self.placement = {"tries": 1, "asked": asked, "chosen": 0, "note": "memory cap: no room for a second candidate"}
            return
        # the clocks first: a launch takes a third longer on a device that has just been idle (the governor's ramp lasts
        # ~40 ms); without this the LAST candidate measured looks best
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.08:
            self._time_launch(reps=4)
        # Consecutive allocations lie in one region of physical memory more often than not: an untouched spacer in front
        # of every further candidate spreads them out -- inside the same budget
        spacer = max(0, min(8 << 30, (budget - (tries - 1) * nbytes) // (tries - 1)))
        spacer = spacer if spacer >= (256 << 20) else 0
        pool, spacers, oom = [self.outputs], [], False
        for _ in range(tries - 1):
            try:
                if spacer:
                    spacers.append(torch.empty(spacer, dtype=torch.uint8, device=self.be.device))
                pool.append(self._alloc_outputs())
            except torch.OutOfMemoryError:
                oom = True          # another tenant of the device got there first: look at what there is
                break
        del spacers
        times = [float("inf")] * len(pool)
        passes = 0
        for order in (range(len(pool)), reversed(range(len(pool)))):       # two passes, the second in reverse order
            for j in order:
                self._bind(pool[j])
                self.update(element_lists)
                times[j] = min(times[j], self._time_launch(reps=3))
            passes += 1
            if max(times) < 1.03 * min(times):
                break               # nothing to choose between: keep the first block
        best = min(range(len(pool)), key=times.__getitem__)
        if times[0] < 1.03 * times[best]:
            best = 0
        self._bind(pool[best])
        self.update(element_lists)
        self.placement = {"tries": len(pool), "asked": asked, "launch_ms": [round(t, 4) for t in times], "chosen": best,
                          "gain_vs_first": round(times[0] / times[best], 4), "passes": passes,
                          "spacer_bytes": spacer, "budget_bytes": budget, "allocation_failed": oom}
        del pool

    def set_detectors(self, detectors):
        """(Re)place the fused read-outs' detectors; takes effect with the next update().  Only for a program that was
        built with detectors (whether the launch carries a read-out is part of the captured graph)."""
        if self.readouts is None and getattr(self, "_signature", None) is not None:
            raise ValueError("this SceneProgram was built without detectors")
        self.detectors = list(detectors)
        if self.readouts is None:
            self.readouts = []
            for ci, (d, s, area) in enumerate(zip(self.detectors, self.sources, self._ro_scratch)):
                d._iscomplete()
                kw = {} if (self._ro_targets is None or self._ro_targets[ci] is None) else {"targets": self._ro_targets[ci]}
                self.readouts.append(self.be.new_chain_readout(d._desc(), s.intensity, self.n, scratch=area,
                                                               lite=self._readout_lite, **kw))
        else:
            for d, ro in zip(self.detectors, self.readouts):
                d._iscomplete()
                ro["struct"].det = d._desc()
                ro.pop("stats", None)

    def _structure(self, element_lists, descs):
        """What a captured launch has baked in: counts, optic kinds and whether defects are present."""
        return (len(element_lists), len(element_lists[0]), tuple(d.kind for d in descs),
                tuple((d.n_defects > 0 or d.n_grid > 0) for d in descs))

    def matches(self, sources, element_lists, kwargs=None):
        """Can this program re-trace the given scene (same bundles, same structure, same options)?"""
        kwargs = kwargs or {}
        if set(kwargs) - {"IgnoreDefects"} or bool(kwargs.get("IgnoreDefects", True)) != self.IgnoreDefects:
            return False
        if len(sources) != self.c or any(a is not b for a, b in zip(sources, self.sources)):
            return False
        if len(element_lists) != self.c or any(len(e) != self.m for e in element_lists):
            return False
        descs = [self._mp.element_descriptor(oe, self.IgnoreDefects, self.be)[0] for els in element_lists for oe in els]
        return (self._structure(element_lists, descs) == self._signature and not any(d.nonfinite for d in descs)
                and any(d.flags & self._abi.ART_FLAG_ZERN_RECURRENCE for d in descs) == self._stepwise)

    def update(self, element_lists):
        """New poses / parameters for the same optics: re-pack the table and copy it over the device image."""
        descs, keep = [], []
        for els in element_lists:
            for oe in els:
                d, k = self._mp.element_descriptor(oe, self.IgnoreDefects, self.be)
                if d.nonfinite:
                    raise ValueError("an element has non-finite parameters")
                descs.append(d)
                keep.append(k)
        sig = self._structure(element_lists, descs)
        if self._signature is not None and sig != self._signature:
            raise ValueError("SceneProgram.update: the optics changed (kinds / counts / defects); build a new program")
        self._signature = sig
        # A Zernike defect above order 16 runs the reference's recurrences per ray in a kernel of its own (art_trace_element
        # only: per-lane row storage): a program that contains one is STEPWISE -- one launch per element into the same
        # preallocated bundles, descriptors as kernel arguments (so: eager launches, no captured graph, no fused read-out)
        stepwise = any(d.flags & self._abi.ART_FLAG_ZERN_RECURRENCE for d in descs)
        if stepwise and (not self._history or self.readouts is not None):
            raise ValueError("a SceneProgram with a Zernike defect above order 16 is traced element by element: it needs "
                             "history=True and cannot carry fused read-outs (detectors=None; Detector.readout works)")
        if self._stepwise is not None and stepwise != self._stepwise:
            raise ValueError("SceneProgram.update: the optics changed (a Zernike order crossed 16); build a new program")
        self._stepwise = stepwise
        if stepwise:
            self._descs, self._keep = descs, keep
            for outs in self.outputs:
                for b in outs:
                    b.touch()
            return
        if self._uploaded is not None:
            self._uploaded.synchronize()       # the previous copy has read the pinned image
        self.flags = self.be.scene_pack(descs, self._views_in, self._views_out, self.c, self.m, self.host, self.readouts)
        self._uploaded = self.be.scene_upload(self.host, self.dev)
        self._keep = keep
        for outs in self.outputs:
            for b in outs:
                if b is not None:
                    b.touch()

    def _mark(self):
        """The output arrays have new contents: drop cached survivor lists, re-attach the fused read-outs."""
        for ci, outs in enumerate(self.outputs):
            for b in outs:
                if b is not None:
                    b.touch()
            if self.readouts is not None:
                self.readouts[ci].pop("stats", None)
                self._mp._attach_readout(outs[-1], self.detectors[ci], 0.0, self.readouts[ci])

    def _launch(self):
        if self._stepwise:
            for ci in range(self.c):
                vin = self._views_in[ci]
                for k in range(self.m):
                    vout = self._views_out[ci * self.m + k]
                    self.be.trace_element(self._descs[ci * self.m + k], vin, vout, self.n)
                    vin = vout
        else:
            self.be.trace_scene(self.dev, self.host, self.n, segments=-(-self.m // 8))
        self._mark()
        if self.post is not None:
            self.post_result = self.post(self.outputs)

    def run(self):
        if self.graph is not None:
            self.graph.replay()
            if hasattr(self.be, "note_launches"):
                self.be.note_launches(self.n, -(-self.m // 8))      # the captured launches, for measurement bookkeeping
            self._mark()
        else:
            self._launch()
        return self.outputs
