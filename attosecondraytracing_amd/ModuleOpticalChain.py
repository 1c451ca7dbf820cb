"""OpticalChain: source bundle + successive optical elements, API of ART/ModuleOpticalChain.py.

`get_output_rays()` is where the tracing is triggered (reference: ModuleOpticalChain.py:183-202); it calls
ModuleProcessing.RayTracingCalculation, i.e. the HIP kernels, and caches the result until the source bundle
or an element pose/parameter changes (pose hash + bundle version counter instead of hashing every Ray)."""
import copy

import numpy as np

from . import ModuleGeometry as mgeo
from . import ModuleOpticalElement as moe
from . import ModuleOpticalRay as mray
from . import ModuleProcessing as mp
from . import ModuleSource as msource
from .bundle import RayBundle


def trace_chain_list(chains, **kwargs):
    """Trace every chain of a list whose cached result is stale in ONE launch (mp.RayTracingCalculationMany) and fill
    the chains' caches, so that the `get_output_rays()` calls that follow (ARTmain.run_ART, one per chain,
    ART/ARTmain.py:304-342) find their result ready.  A list as `OEPlacement` returns it -- chains that differ only in
    poses -- shares one scene table; anything else falls back to one launch per chain."""
    with mgeo.frozen_hashes():        # (nothing below modifies an element: one hash per element serves every cache key)
        return _trace_chain_list(chains, kwargs)


def _trace_chain_list(chains, kwargs):
    keys = [ch._cache_key(kwargs) for ch in chains]          # once per chain: every step below compares or stores it
    stale = [(ch, k) for ch, k in zip(chains, keys) if k != ch._last_key and getattr(ch, "_program", None) is None]
    lazy = kwargs.get("history") == "lazy"
    if len(stale) > 1 and not (lazy and kwargs.get("want", -1) not in (-1, len(stale[0][0].optical_elements) - 1)):
        kw = {k: v for k, v in kwargs.items() if k not in ("history", "want")}
        # (lazy: what follows is the analysis of every chain's last bundle -- the launch forms its first pass)
        outs = mp.RayTracingCalculationMany([ch.source_rays for ch, _ in stale], [ch.optical_elements for ch, _ in stale],
                                            history=not lazy, sums=lazy, **kw)
        for (ch, key), o in zip(stale, outs):
            # lazy: only the last bundle of every chain was written; the rest appears on first access (mp.LazyHistory)
            ch._output_rays = mp.LazyHistory(ch.source_rays, ch.optical_elements, first=o[-1], **kw) if lazy else o
            ch._last_key = key
    return [ch._output_for(key, kwargs) for ch, key in zip(chains, keys)]


class OpticalChain:
    def __init__(self, source_rays, optical_elements, description="", loop_variable_name=None,
                 loop_variable_value=None, *, _placed=False):
        # _placed (OEPlacement's loop lists): this chain's source is a bundle object of its own over the SAME device arrays
        # as the other chains' (bundle.RayBundle.alias) instead of a private copy of 65 bytes per ray.  The elements are
        # deep-copied either way, as in the reference: the chains of a list share their optic objects until here.
        self.source_rays = source_rays.alias() if _placed else copy.deepcopy(source_rays)
        self.optical_elements = copy.deepcopy(optical_elements)
        self.description = description
        self.loop_variable_name = loop_variable_name
        self.loop_variable_value = loop_variable_value
        self._output_rays = None
        self._last_key = None
        self._program = None

    @classmethod
    def _adopt(cls, source_rays, optical_elements, description=""):
        """OEPlacement's constructor: `optical_elements` were built for THIS chain alone (own element objects, own copies of
        the optics: mp._placeChains), so they are adopted as they are -- the state `OpticalChain(source, elements)` reaches
        through its deep copy --, and the source is an alias of the list's one source bundle (a bundle object of its own
        over the SAME device arrays, bundle.RayBundle.alias, instead of a private copy of 65 bytes per ray)."""
        new = cls.__new__(cls)
        new.source_rays = source_rays.alias()
        new.optical_elements = optical_elements
        new.description = description
        new.loop_variable_name = None
        new.loop_variable_value = None
        new._output_rays = None
        new._last_key = None
        new._program = None
        return new

    # ------------------------------------------------------------------ properties
    @property
    def source_rays(self):
        return self._source_rays

    @source_rays.setter
    def source_rays(self, source_rays):
        if isinstance(source_rays, RayBundle):
            self._source_rays = source_rays
        elif type(source_rays) == list and all(isinstance(x, mray.Ray) for x in source_rays):
            self._source_rays = RayBundle.from_ray_list(source_rays)
        else:
            raise TypeError("Source_rays must be list of Ray-objects.")

    @property
    def optical_elements(self):
        return self._optical_elements

    @optical_elements.setter
    def optical_elements(self, optical_elements):
        if type(optical_elements) == list and all(isinstance(x, moe.OpticalElement) for x in optical_elements):
            self._optical_elements = optical_elements
        else:
            raise TypeError("Optical_elements must be list of OpticalElement-objects.")

    @property
    def loop_variable_name(self):
        return self._loop_variable_name

    @loop_variable_name.setter
    def loop_variable_name(self, loop_variable_name):
        if not (type(loop_variable_name) == str or loop_variable_name is None):
            raise TypeError("loop_variable_name must be a string.")
        self._loop_variable_name = loop_variable_name

    @property
    def loop_variable_value(self):
        return self._loop_variable_value

    @loop_variable_value.setter
    def loop_variable_value(self, loop_variable_value):
        if not (type(loop_variable_value) in [int, float, np.float64] or loop_variable_value is None):
            raise TypeError("loop_variable_value must be a number of types int or float.")
        self._loop_variable_value = loop_variable_value

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_program"] = None          # a captured HIP graph does not travel into an archive
        return st

    # ------------------------------------------------------------------ tracing
    def copy_chain(self):
        return OpticalChain(self.source_rays, self.optical_elements, self.description)

    def _cache_key(self, kwargs):
        # a lazy history holds the same content as a full one: `history="lazy"` / `want` are not part of the identity
        kw = {k: v for k, v in kwargs.items() if k != "want" and not (k == "history" and v in ("lazy", True))}
        return (hash(self.source_rays), mp._hash_list_of_objects(self.optical_elements), tuple(sorted(kw.items())))

    def get_output_rays(self, **kwargs):
        """List of ray bundles after each optical element; recomputed only when something changed.

        `history="lazy"` (+ `want=k`, default the last element): only bundle k is written by the trace, the others are
        materialised -- bit-identically, by ONE re-trace with the full history -- the first time they are accessed
        (mp.LazyHistory).  What ARTmain uses: it analyses one bundle per chain (ART/ARTmain.py:254-255).

        A chain that was `compile()`d re-traces by rewriting its device-resident scene table and replaying a captured
        HIP graph; the bundles it returns are then always the SAME objects (their arrays are overwritten by the next
        re-trace), which is what makes a pose scan on small bundles GPU-bound instead of launch-bound."""
        return self._output_for(self._cache_key(kwargs), kwargs)

    def _output_for(self, key, kwargs):
        """get_output_rays for a cache key the caller has computed already (trace_chain_list: once per chain)."""
        if key != self._last_key:
            prog = getattr(self, "_program", None)
            plain = {k: v for k, v in kwargs.items() if k != "want" and not (k == "history" and v in ("lazy", True))}
            if prog is not None and prog.matches([self.source_rays], [self.optical_elements], plain):
                prog.update([self.optical_elements])
                self._output_rays = prog.run()[0]
            else:
                self._program = None
                print("...ray-tracing...", end="", flush=True)
                if kwargs.get("history") == "lazy":
                    kw = {k: v for k, v in kwargs.items() if k != "history"}
                    self._output_rays = mp.LazyHistory(self.source_rays, self.optical_elements, **kw)
                else:
                    self._output_rays = mp.RayTracingCalculation(self.source_rays, self.optical_elements, **kwargs)
                print("\r\033[K", end="", flush=True)
            self._last_key = key
        return self._output_rays

    def compile(self, **kwargs):
        """Opt in to graph replay for repeated re-traces of this chain with changed poses / source contents (same
        optics, same ray count): see graph.SceneProgram.  Returns the program."""
        from .graph import SceneProgram
        self._program = SceneProgram([self.source_rays], [self.optical_elements], **kwargs)
        self._output_rays = self._program.run()[0]
        self._last_key = self._cache_key(kwargs)
        return self._program

    def render(self):
        from . import ModuleAnalysisAndPlots as mplots
        return mplots.RayRenderGraph(self, None, maxRays=300, OEpoints=3000, scale_spheres=5, draw_mesh=False,
                                     cycle_ray_colors=False)

    # ------------------------------------------------------------------ source (mis-)alignment
    def _incidence_plane_axes(self):
        """(central ray, normal of the first mirror hit at non-normal incidence) -- ModuleOpticalChain.py:255-269."""
        central = mp.FindCentralRay(self.source_rays).vector
        for OE in self.optical_elements:
            if "Mirror" in OE.type.type and np.linalg.norm(np.cross(central, OE.normal)) > 1e-10:
                return central, OE.normal
        raise Exception("There doesn't seem to be a non-normal-incidence mirror in this optical chain, "
                        "so you should rather give 'axis' as a numpy-array of length 3.")

    def shift_source(self, axis, distance: float):
        """Shift the source bundle by `distance` mm along a vector or "vert"/"horiz"/"random"
        (ModuleOpticalChain.py:219-292)."""
        if type(distance) not in [int, float, np.float64]:
            raise ValueError('The "distance"-argument must be an int or float number.')
        central, OEnormal = self._incidence_plane_axes()
        if type(axis) == np.ndarray and len(axis) == 3:
            t = axis
        else:
            perp = np.cross(central, OEnormal)
            horiz = np.cross(perp, central)
            if axis == "vert":
                t = perp
            elif axis == "horiz":
                t = horiz
            elif axis == "random":
                t = np.random.uniform(-1, 1, 1) * perp + np.random.uniform(-1, 1, 1) * horiz
            else:
                raise ValueError('The shift direction must be specified by "axis" as one of ["vert", "horiz", "random"].')
        self.source_rays = mgeo.TranslationRayList(self.source_rays, distance * mgeo.Normalize(t))

    def tilt_source(self, axis, angle: float):
        """Rotate the source directions by `angle` degrees about a vector or "in_plane"/"out_plane"/"random"
        (ModuleOpticalChain.py:294-369)."""
        if type(angle) not in [int, float, np.float64]:
            raise ValueError('The "angle"-argument must be an int or float number.')
        central, OEnormal = self._incidence_plane_axes()
        if type(axis) == np.ndarray and len(axis) == 3:
            r = axis
        else:
            r_in = np.cross(central, OEnormal)
            r_out = np.cross(r_in, central)
            if axis == "in_plane":
                r = r_in
            elif axis == "out_plane":
                r = r_out
            elif axis == "random":
                r = np.random.uniform(-1, 1, 1) * r_in + np.random.uniform(-1, 1, 1) * r_out
            else:
                raise ValueError('The tilt axis must be specified by as one of ["in_plane", "out_plane", "random"] '
                                 'or as a numpy-array of length 3.')
        self.source_rays = mgeo.RotationAroundAxisRayList(self.source_rays, r, np.deg2rad(angle))

    _SOURCE_LOOPS = {
        "tilt_in_plane": "source tilt in-plane (deg)",
        "tilt_out_plane": "source tilt out-of-plane (deg)",
        "tilt_random": "source tilt random axis (deg)",
        "shift_vert": "source shift vertical (mm)",
        "shift_horiz": "source shift horizontal (mm)",
        "shift_random": "source shift random-direction (mm)",
        "divergence": "point-source divergence half-angle (rad)",
    }

    def get_source_loop_list(self, axis: str, loop_variable_values):
        """Variations of this chain with a modified source (ModuleOpticalChain.py:371-447)."""
        if axis not in self._SOURCE_LOOPS:
            raise ValueError("For automatic loop-list generation, the axis must be one of "
                             + str(list(self._SOURCE_LOOPS)) + ".")
        if type(loop_variable_values) not in [list, np.ndarray]:
            raise ValueError("For automatic loop-list generation, the loop_variable_values must be a list or a numpy-array.")
        chains = []
        for x in loop_variable_values:
            ch = self.copy_chain()
            ch.loop_variable_name = self._SOURCE_LOOPS[axis]
            ch.loop_variable_value = x
            if axis.startswith("tilt_"):
                ch.tilt_source(axis[5:], x)
            elif axis.startswith("shift_"):
                ch.shift_source(axis[6:], x)
            else:
                first, last = self.source_rays[0], self.source_rays[-1]
                src = msource.PointSource(first.point, first.vector, x, len(self.source_rays), first.wavelength)
                ch.source_rays = msource.ApplyGaussianIntensityToRayList(src, last.intensity)
            chains.append(ch)
        return chains

    # ------------------------------------------------------------------ element (mis-)alignment
    def _check_index(self, OEindx):
        if abs(OEindx) > len(self.optical_elements):
            raise ValueError('The "OEnumber"-argument is out of range compared to the length of '
                             'OpticalChain.optical_elements.')

    def rotate_OE(self, OEindx: int, axis: str, angle: float):
        """ModuleOpticalChain.py:449-488."""
        self._check_index(OEindx)
        if type(angle) not in [int, float, np.float64]:
            raise ValueError('The "angle"-argument must be an int or float number.')
        names = {"pitch": "rotate_pitch_by", "roll": "rotate_roll_by", "yaw": "rotate_yaw_by",
                 "random": "rotate_random_by", "rotate_random": "rotate_random_by"}
        if axis not in names:
            raise ValueError('The "axis"-argument must be a string out of ["pitch", "roll", "yaw", "random"].')
        getattr(self.optical_elements[OEindx], names[axis])(angle)

    def shift_OE(self, OEindx: int, axis: str, distance: float):
        """ModuleOpticalChain.py:490-531."""
        self._check_index(OEindx)
        if type(distance) not in [int, float, np.float64]:
            raise ValueError('The "dist"-argument must be an int or float number.')
        if axis not in ("normal", "major", "cross", "random"):
            raise ValueError('The "axis"-argument must be a string out of ["normal", "major", "cross", "random"].')
        getattr(self.optical_elements[OEindx], "shift_along_" + axis)(distance)

    def get_OE_loop_list(self, OEindx: int, axis: str, loop_variable_values):
        """Variations of this chain moving one degree of freedom of one element (ModuleOpticalChain.py:533-614)."""
        self._check_index(OEindx)
        name = self.optical_elements[OEindx].type.type + "_idx_" + str(OEindx)
        labels = {
            "pitch": " pitch rotation (deg)", "roll": " roll rotation (deg)", "yaw": " yaw rotation (deg)",
            "rotate_random": " random rotation (deg)", "shift_normal": " shift along normal axis (mm)",
            "shift_major": " shift along major axis (mm)",
            "shift_cross": " shift along (normal x major)-direction (mm)",
            "shift_random": " shift along random axis (mm)",
        }
        if axis not in labels:
            raise ValueError("For automatic loop-list generation, the axis must be one of " + str(list(labels)) + ".")
        if type(loop_variable_values) not in [list, np.ndarray]:
            raise ValueError("For automatic loop-list generation, the loop_variable_values must be a list or a numpy-array.")
        chains = []
        for x in loop_variable_values:
            ch = self.copy_chain()
            ch.loop_variable_name = name + labels[axis]
            ch.loop_variable_value = x
            if axis.startswith("shift_"):
                ch.shift_OE(OEindx, axis[6:], x)
            else:
                ch.rotate_OE(OEindx, axis, x)
            chains.append(ch)
        return chains

    def get_OE_random_loop_list(self, rotate_std: float, shift_std: float, number_sims: int):
        """Random mis-alignments of all elements (ModuleOpticalChain.py:616-657)."""
        name = ("all optical elements randomly rotated with std=" + str(rotate_std)
                + "deg and and shifted with Std=" + str(shift_std) + "mm")
        chains = []
        for i in range(number_sims):
            ch = self.copy_chain()
            ch.loop_variable_name = name
            ch.loop_variable_value = i
            for j in range(len(self.optical_elements)):
                ch.rotate_OE(j, "random", np.random.normal(loc=0, scale=rotate_std))
                ch.shift_OE(j, "random", np.random.normal(loc=0, scale=shift_std))
            chains.append(ch)
        return chains
