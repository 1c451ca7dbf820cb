"""Launcher logic of ART (reference: ARTmain.py at the repository root): merge option dictionaries with the
defaults, trace each OpticalChain, place/optimise the detector, summarise, optionally archive."""
import importlib.util
import os
import sys

import numpy as np

from . import ModuleAnalysisAndPlots as mplots
from . import ModuleDetector as mdet
from . import ModuleOpticalChain as moc
from . import ModuleProcessing as mp

_NICELINE = "_" * 99


def print_banner(i=-1):
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "VERSION"), "r") as f:
        version = f.read().strip()
    print(_NICELINE)
    print("ART - Attosecond Ray Tracing, MI355X-native ray-bundle propagation")
    print(f"v{version}")
    print(_NICELINE, flush=True)


def load_config(config):
    """Pick the chain(s) and the three option dictionaries out of an imported CONFIG module (ARTmain.py:56-96)."""
    print("...setting up and importing optical chain(s)...", end="", flush=True)
    if hasattr(config, "OpticalChainList"):
        chains = config.OpticalChainList
    elif hasattr(config, "OpticalChain"):
        chains = config.OpticalChain
    else:
        raise ValueError("Could not import an optical-chain-object or list thereof with the name OpticalChain or "
                         "OpticalChainList.")
    opts = []
    for name in ("SourceProperties", "DetectorOptions", "AnalysisOptions"):
        if hasattr(config, name):
            opts.append(getattr(config, name))
        else:
            print(f"No {name}-dictionary provided, will use defaults.")
            opts.append({})
    print("\r\033[K", end="", flush=True)
    return (chains, *opts)


def complete_defaults(SourceProperties, DetectorOptions, AnalysisOptions):
    """User dictionaries on top of the defaults (ARTmain.py:99-110).  Like the reference this updates the
    module-level default dictionaries in place."""
    from .DefaultOptions import DefaultAnalysisOptions, DefaultDetectorOptions, DefaultSourceProperties
    DefaultSourceProperties.update(SourceProperties)
    DefaultDetectorOptions.update(DetectorOptions)
    DefaultAnalysisOptions.update(AnalysisOptions)
    return DefaultSourceProperties, DefaultDetectorOptions, DefaultAnalysisOptions


def setup_detector(OpticalChain, DetectorOptions, RayList=None):
    """ARTmain.py:113-144."""
    ref = OpticalChain.optical_elements[DetectorOptions["ReflectionNumber"]].position
    ref = np.asarray(ref, dtype=float)
    if DetectorOptions["ManualDetector"]:
        for key in ("DetectorCentre", "DetectorNormal"):
            if DetectorOptions[key] is None:
                raise RuntimeError(f'For manual detector placement you need to specify "{key}" in the '
                                   '"DetectorOptions"-dictionary.')
        return mdet.Detector(ref, DetectorOptions["DetectorCentre"], DetectorOptions["DetectorNormal"])
    if DetectorOptions["DistanceDetector"] is None:
        raise RuntimeError('For automatic detector placement you need to specify "DistanceDetector" in the '
                           '"DetectorOptions"-dictionary.')
    if RayList is None:
        raise RuntimeError("For automatic detector placement you need to add a RayList as an input.")
    det = mdet.Detector(ref)
    det.autoplace(RayList, DetectorOptions["DistanceDetector"])
    return det


def optimize_detector(RayListAnalysed, Detector, DetectorOptions, verbose=True, maxRaystoConsider=1000,
                      IntensityWeighted=False, Amplitude=None, Precision=3):
    """Autofocus (ARTmain.py:147-190).  The reference sub-samples `maxRaystoConsider` random rays to keep its
    Python loops affordable; that is still honoured when a number is given (index sampling replaces
    np.random.choice on a list of Ray objects), but `run_ART` passes None: the scan costs two passes over the bundle
    on the GPU, so all rays are used and the result is deterministic."""
    rays = RayListAnalysed
    if maxRaystoConsider is not None and len(rays) > maxRaystoConsider:
        rays = rays.subset(np.random.default_rng().choice(len(rays), maxRaystoConsider, replace=False))
    det, spot, dur = mp.FindOptimalDistance(Detector, rays, DetectorOptions["OptFor"], Amplitude, Precision,
                                            IntensityWeighted, verbose)
    if verbose:
        _report_optimum(det, spot, dur, DetectorOptions["OptFor"], IntensityWeighted)
    return det, spot, dur


def _report_optimum(det, spot, dur, OptFor, IntensityWeighted):
    s = f"The optimal detector distance is {det.get_distance():.3f} mm, with"
    if IntensityWeighted:
        s += " intensity-weighted"
    if OptFor in ["intensity", "spotsize"]:
        s += f" spatial std of {spot*1e3:.3g} μm"
    if OptFor in ["intensity", "duration"]:
        s += f" temporal std of {dur:.3g} fs."
    print(s, flush=True)


def make_plots(OpticalChain, RayListAnalysed, Detector, SourceProperties, DetectorOptions, AnalysisOptions):
    """ARTmain.py:193-244: the plots selected in AnalysisOptions."""
    A = AnalysisOptions
    if A["plot_Render"]:
        mplots.RayRenderGraph(OpticalChain)
    for kind in ("Delay", "Intensity", "Incidence"):
        if A[f"plot_{kind}MirrorProjection"]:
            mplots.MirrorProjection(OpticalChain, DetectorOptions["ReflectionNumber"], Detector, kind)
    if A["plot_SpotDiagram"]:
        mplots.SpotDiagram(RayListAnalysed, Detector, A["DrawAiryAndFourier"])
    for kind in ("Delay", "Intensity", "Incidence"):
        if A[f"plot_{kind}SpotDiagram"]:
            mplots.SpotDiagram(RayListAnalysed, Detector, A["DrawAiryAndFourier"], kind)
    for kind in ("Delay", "Intensity", "Incidence"):
        if A[f"plot_{kind}Graph"]:
            mplots.DelayGraph(RayListAnalysed, Detector, SourceProperties["DeltaFT"], A["DrawAiryAndFourier"], kind)


def analyse_chain_list(OpticalChainList, SourceProperties, DetectorOptions, AnalysisOptions, loop=True, announce=False):
    """`run_ART` for every chain of a list (ARTmain.py:248-300 per chain, :304-342 the loop) with the device work of ALL
    chains batched: one scene launch traces them (moc.trace_chain_list), then ONE device analysis (analysis.analyse:
    four launches, blockIdx.y = chain, one copy back) yields every chain's transmitted energy, its detector placed on
    the mean ray (ART/ModuleDetector.py:109-137) and the read-out moments from which the autofocus search
    (ART/ModuleProcessing.py:317-460) or the result summary (ART/ModuleAnalysisAndPlots.py:81-129) of every chain
    follows by arithmetic on the host.  Same numbers as calling run_ART chain by chain (which is this function with a
    list of one).  Returns [(OpticalChain, Detector, ETransmission, SpotSizeSD, DurationSD)]."""
    from . import ModuleGeometry as mgeo
    chains = list(OpticalChainList)
    with mgeo.frozen_hashes():        # (nothing below modifies an element: one hash per element serves every cache key)
        try:
            return _analyse_chain_list(chains, SourceProperties, DetectorOptions, AnalysisOptions, loop, announce)
        except Exception:
            if len(chains) <= 1:
                raise
        # One chain of the list cannot be analysed (all its rays lost, a zero search amplitude, ...).  The reference's loop
        # (ART/ARTmain.py:304-342) reports every chain in front of it before it fails: so does this second pass, chain by
        # chain (the traces are cached), which ends in the same exception.
        out = []
        for i, ch in enumerate(chains):
            if announce:
                print("Optical Chain " + str(i) + "/" + str(len(chains)) + " ", end="", flush=True)
            out.append(_analyse_chain_list([ch], SourceProperties, DetectorOptions, AnalysisOptions, loop, False)[0])
        return out


def _analyse_chain_list(OpticalChainList, SourceProperties, DetectorOptions, AnalysisOptions, loop=True, announce=False):
    from . import analysis
    chains = list(OpticalChainList)
    k_an = DetectorOptions["ReflectionNumber"]
    # ONE bundle of every history is analysed: trace it alone; any other entry of `output_rays` (a plot of another
    # element, Ray.path tuples, an archive) is materialised bit-identically on first access (mp.LazyHistory)
    outs = moc.trace_chain_list(chains, history="lazy", want=k_an)
    analysed = [o[k_an] for o in outs]
    # requests: every chain's analysed bundle (placement + moments) and every DISTINCT source (sum of intensities)
    requests, src_slot = [], {}
    if DetectorOptions["ManualDetector"] or DetectorOptions["DistanceDetector"] is not None:
        for ch, B in zip(chains, analysed):
            if DetectorOptions["ManualDetector"]:
                requests.append((B, "manual", setup_detector(ch, DetectorOptions)))
            else:
                requests.append((B, "autoplace", DetectorOptions["DistanceDetector"]))
    elif chains:
        setup_detector(chains[0], DetectorOptions, analysed[0])      # raises the reference's RuntimeError
    for ch in chains:
        key = ch.source_rays.content_key()
        if key not in src_slot:
            if ch.source_rays.intensity is None or analysed[0].intensity is None:
                raise TypeError("rays carry no intensity")
            src_slot[key] = len(requests)
            requests.append((ch.source_rays, "sums", None))
    res = analysis.analyse(requests)
    # every chain's detector, then (AutoDetectorDistance) the autofocus search of ALL chains at once: each scan level
    # evaluates the positions of every chain in one broadcast (mp._optimise_many, arithmetic on the analyses' 64 doubles)
    detectors = []
    for i, ch in enumerate(chains):
        if DetectorOptions["ManualDetector"]:
            Detector = requests[i][2]
            Detector._analysis = res[i]
        else:
            Detector = mdet.Detector(np.asarray(ch.optical_elements[k_an].position, dtype=float))
            Detector._adopt(res[i])
        detectors.append(Detector)
    optima = None
    if DetectorOptions["AutoDetectorDistance"]:
        # (the search itself is silent here: what the reference prints per chain -- the searched range, the "no minimum"
        # remark -- is printed below, under the chain it belongs to)
        optima = mp._optimise_many([(d, B, res[i]) for i, (d, B) in enumerate(zip(detectors, analysed))],
                                   DetectorOptions["OptFor"], None, 3, True, False, announce=False)
    results = []
    for i, (ch, B) in enumerate(zip(chains, analysed)):
        if announce:
            print("Optical Chain " + str(i) + "/" + str(len(chains)) + " ", end="", flush=True)
        ETransmission = 100 * float(res[i].sum_w) / float(res[src_slot[ch.source_rays.content_key()]].sum_w)
        if AnalysisOptions["verbose"]:
            print(_NICELINE, flush=True)
            if isinstance(ch.description, str) and len(ch.description) > 0:
                print("***" + ch.description + "*** :")
            if ch.loop_variable_name is not None and ch.loop_variable_value is not None:
                print("For " + ch.loop_variable_name + " = " + "{:f}".format(ch.loop_variable_value) + ":\n")
                print("The optical setup has an energy transmission of " + "{:.1f}".format(ETransmission) + "%.\n")
        if optima is not None:
            Detector, SpotSizeSD, DurationSD, (outside, lo_, hi_) = optima[i]
            if AnalysisOptions["verbose"]:     # (the reference's progress line, ART/ModuleProcessing.py:436-441, erased like there)
                print(f"Searching optimal detector position for *{DetectorOptions['OptFor']}* within [{lo_:.3f}, {hi_:.3f}] mm...",
                      end="", flush=True)
                print("\r\033[K", end="", flush=True)
            if outside:
                print("There`s no minimum-size/duration focus in the searched range.")
            if AnalysisOptions["verbose"]:
                _report_optimum(Detector, SpotSizeSD, DurationSD, DetectorOptions["OptFor"], True)
        else:
            Detector = detectors[i]
            SpotSizeSD, DurationSD = mplots.GetResultSummary(Detector, B, AnalysisOptions["verbose"])
        if AnalysisOptions["verbose"]:
            print(_NICELINE + "\n")
        if any(AnalysisOptions[k] for k in AnalysisOptions if k.startswith("plot_")):
            make_plots(ch, B, Detector, SourceProperties, DetectorOptions, AnalysisOptions)
        results.append((ch, Detector, ETransmission, SpotSizeSD, DurationSD))
    return results


def run_ART(OpticalChain, SourceProperties, DetectorOptions, AnalysisOptions, loop=False):
    """One chain: trace, transmission, detector, summary, plots (ARTmain.py:248-300)."""
    return analyse_chain_list([OpticalChain], SourceProperties, DetectorOptions, AnalysisOptions, loop)[0]


def main(OpticalChainList, SourceProperties, DetectorOptions, AnalysisOptions, save_file_name=None):
    """ARTmain.py:304-342."""
    SourceProperties, DetectorOptions, AnalysisOptions = complete_defaults(SourceProperties, DetectorOptions,
                                                                           AnalysisOptions)
    names = ["OpticalChain", "Detector", "ETransmission", "SpotSizeSD", "DurationSD"]
    kept_data = {n: [] for n in names}
    if isinstance(OpticalChainList, moc.OpticalChain):
        OpticalChainList = [OpticalChainList]
        loop = False
    elif not isinstance(OpticalChainList, list):
        raise ValueError("The supplied OpticalChain is neither an OpticalChain-object, nor a list of those, as it "
                         "should be.")
    else:
        loop = True
    # the whole loop list in ONE trace launch and ONE device analysis (chains that differ only in poses share a
    # device-resident scene table; blockIdx.y = chain in both): analyse_chain_list
    for results in analyse_chain_list(OpticalChainList, SourceProperties, DetectorOptions, AnalysisOptions, loop,
                                      announce=True):
        for n, v in zip(names, results):
            kept_data[n].append(v)
    if AnalysisOptions["save_results"]:
        print("...saving data...", end="", flush=True)
        mp.save_compressed(kept_data, save_file_name)
        print("\r\033[K", end="", flush=True)
    return kept_data


def cli(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print("Usage: python ARTmain.py CONFIG_FILE")
        return 2
    print_banner(1)
    config_file = argv[1]
    filename = os.path.basename(config_file)
    spec = importlib.util.spec_from_file_location(filename, config_file)
    module = importlib.util.module_from_spec(spec)
    sys.modules[filename] = module
    spec.loader.exec_module(module)
    chains, src, det, ana = load_config(module)
    main(chains, src, det, ana, save_file_name=config_file)
    return 0
