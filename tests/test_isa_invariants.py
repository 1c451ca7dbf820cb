"""CPU (no GPU needed): performance-critical properties of the gfx950 device code, read off the compiler's assembly
(hipcc cross-compiles here).  Each of them was worth 10-50 % when it was found:
  * the fused chain kernel fits 5 waves per SIMD (<= 96 VGPRs) without scratch;
  * after its first store the fused chain / scene kernel never waits on vmcnt again -- gfx9 counts loads and stores in
    one in-order queue, so a late load, a fence or a scratch reload makes every wave wait for the acknowledgement of all
    its outstanding stores instead of retiring;
  * the scene kernel fetches its descriptors with scalar loads only (no vector loads of table data);
  * the Zernike evaluators keep their coefficients scalar (no LDS traffic in the defect kernels; <= 128 VGPRs);
  * the vector-instruction count of the torus path and of the fused kernel stays where the round-2 work put it
    (compile-time constants as scalar operands, no IEEE sqrtf / division expansions, no select chains in the tail)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "attosecondraytracing_amd", "csrc", "art_kernels.hip")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "art.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC],
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        g = lambda key: int(re.search(r"\.amdhsa_%s (\d+)" % key, m.group(2)).group(1))
        meta[m.group(1)] = {"vgpr": g("next_free_vgpr"), "lds": g("group_segment_fixed_size"),
                            "scratch": g("private_segment_fixed_size")}
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
    res = {}
    for name, body in zip(parts[1::2], parts[2::2]):
        if name in meta:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = dem.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            res[dem] = dict(meta[name], code=body[:body.rfind("s_endpgm") + 8].splitlines())
    return res


def _after_first_store(code, pattern):
    first = next(i for i, l in enumerate(code) if "buffer_store" in l)
    return [l.strip() for l in code[first:] if re.search(pattern, l)]


def test_fused_chain_kernel_occupancy(kernels):
    for name in ("k_trace_chain<false, 5>", "k_trace_scene<false, 5>"):
        k = kernels[name]
        assert k["vgpr"] <= 96, (name, k["vgpr"])        # 5 waves per SIMD
        assert k["scratch"] == 0, (name, k["scratch"])    # a scratch reload is a VMEM load behind the stores
        assert k["lds"] <= 32 * 1024, (name, k["lds"])    # 5 workgroups per CU fit 160 KiB
    for name in ("k_trace_chain<true, 4>", "k_trace_scene<true, 4>"):
        k = kernels[name]
        assert k["vgpr"] <= 128, (name, k["vgpr"])
        # (the compiler may reserve a few bytes of private segment without using them: what counts is that no spill
        # or reload instruction exists)
        assert not [l for l in k["code"] if re.search(r"scratch_(load|store)|Folded (Spill|Reload)", l)], name


def test_two_rays_per_lane_body(kernels):
    """The body chains with a mask are traced by (two neighbouring slots per lane): 4 waves per SIMD without scratch, ONE
    EXECUTED barrier (the read-out tail's or the sums tail's), 16-byte accesses for all fp64 streams, and -- like the one-ray body -- no wait on vmcnt
    and no load after its first store."""
    for name in ("k_trace_chain2<false, 4>", "k_trace_scene2<false, 4>"):
        k = kernels[name]
        assert k["vgpr"] <= 128 and k["scratch"] == 0 and k["lds"] <= 32 * 1024, (name, k["vgpr"], k["scratch"], k["lds"])
        code = k["code"]
        # two barriers in the code, ONE executed: the read-out tail's or the sums tail's (ArtChainReadout.sums, round 5), a
        # wave-uniform choice
        assert sum("s_barrier" in l for l in code) == 2, name
        # 8 16-byte loads (7 ray streams + the weights); the scene kernel carries the 7 ray streams a second time with the
        # default cache policy (the shared input of a chain-interleaved launch, round 4), behind a wave-uniform branch
        assert sum("buffer_store_dwordx4" in l for l in code) == 11, name
        assert sum("buffer_load_dwordx4" in l for l in code) == (15 if "scene" in name else 8), name
        assert not _after_first_store(code, r"s_waitcnt.*vmcnt"), name
        assert not _after_first_store(code, r"\b(buffer_load|global_load|flat_load|scratch_load)\b"), name


def test_no_wait_for_store_acknowledgements(kernels):
    for name in ("k_trace_chain<false, 5>", "k_trace_scene<false, 5>"):
        waits = _after_first_store(kernels[name]["code"], r"s_waitcnt.*vmcnt")
        assert not waits, (name, waits[:5])
        # nothing that needs such a wait either: loads of any kind.  (The 16-byte store path has two s_barrier per
        # element; they order LDS only -- the check above proves that no vmcnt wait comes with them.)
        late = _after_first_store(kernels[name]["code"], r"\b(buffer_load|global_load|flat_load|scratch_load)\b")
        assert not late, (name, late[:5])


def test_scene_descriptors_are_scalar_loads(kernels):
    for name in ("k_trace_scene<false, 5>", "k_trace_scene<true, 4>"):
        code = kernels[name]["code"]
        assert sum("s_load" in l for l in code) > 100, name
        assert not [l for l in code if re.search(r"\b(global_load|flat_load)\b", l) and "grid" in name], name
    # without gridded defects there is no vector load besides the ray's own 9 streams (7 ray streams + alive + weight) --
    # and, since round 4, a second copy of the 8 ray / alive loads with the default cache policy (the shared input of a
    # chain-interleaved launch), behind a wave-uniform branch
    code = kernels["k_trace_scene<false, 5>"]["code"]
    assert not [l for l in code if re.search(r"\b(global_load|flat_load)\b", l)]
    assert sum("buffer_load" in l for l in code) == 17
    assert sum("buffer_load" in l for l in kernels["k_trace_chain<false, 5>"]["code"]) == 9


def test_zernike_coefficients_stay_scalar(kernels):
    for kind in range(6):
        k = kernels[f"k_trace_element<{kind}, true>"]
        assert k["vgpr"] <= 104 and k["lds"] == 0 and k["scratch"] == 0, (kind, k["vgpr"], k["lds"], k["scratch"])
        code = k["code"]
        # coefficients arrive through s_load and are consumed as the scalar operand of v_fma_f64 ...
        assert sum(bool(re.search(r"v_fma_f64 v\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\], s\[\d+:\d+\]", l)) for l in code) > 300, kind
        # ... not copied into VGPRs first (the compiler's two-address v_fmac form needs 2 v_mov per coefficient)
        assert sum("v_mov_b32" in l for l in code) < 400, kind


def _valu(code):
    return [l.split()[0] for l in (x.strip() for x in code) if l.startswith("v_")]


def test_vector_instruction_budget(kernels):
    """Static VALU counts (all paths of the kernel, lemon and retry blocks included): a quarter below the mid-round build.
    A constant that slips back into a VGPR pair, an IEEE expansion or a select chain shows up here before it shows up
    as milliseconds (tools/valu_count.py)."""
    torus = _valu(kernels["k_trace_element<3, false>"]["code"])
    chain = _valu(kernels["k_trace_chain<false, 5>"]["code"])
    assert len(torus) <= 730, len(torus)            # 694 (mid-round: 838 with one solver path less)
    # round 4: 2203 = 2066 + the LITE tail's block (ArtChainReadout.lite), which the default path branches around -- the
    # EXECUTED count per wave is unchanged (SQ_INSTS_VALU 1324 per wave in profiles/r04_relay4_sq.md, 1321 in round 3)
    # round 5: 2272 = 2203 + the SUMS tail's block (ArtChainReadout.sums: one LDS transpose of 8 sums), again behind a
    # wave-uniform branch the read-out path never enters
    assert len(chain) <= 2330, len(chain)           # 2272 (round 4: 2203; round 3: 2066; mid-round-2: 2558)
    # no IEEE division / sqrt expansions on the torus path (the quadrics keep ONE IEEE division on purpose: q / a with a
    # leading coefficient that may be 1e-34)
    assert not [i for i in torus if i.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_sqrt_f32"))]
    # the atan polynomial's 10 coefficients are scalar operands: few literal moves into VGPRs remain in the whole kernel
    lit = [l for l in kernels["k_trace_element<3, false>"]["code"] if re.search(r"v_mov_b32_e32 v\d+, 0x[0-9a-f]{6,}", l)]
    assert len(lit) <= 24, len(lit)
