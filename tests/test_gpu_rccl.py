"""GPU (-m gpu): the N > 1 code path of bench.py through RCCL itself.  A test box has ONE GPU, so the process group has
one rank (ART_FORCE_DIST=1) -- but it is a real `nccl` group on the device: init_process_group, the per-step all-gather
(statistics + sample, pipelined), the header all-gather nobody waits for and the ONE gather of the survivor records, the stream ordering
between torch's stream and the communicator's, all run as they do at N = 8; only the peers are missing.  Each run is a
FRESH child process (never a re-exec of the pytest process).  Plus the survivor-record kernel against torch."""
import json
import os
import subprocess
import sys

import pytest

from conftest import report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, **extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ART_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION")
    env.update(extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]          # the contract: ONE JSON line on stdout (RCCL's banner goes to stderr)
    return json.loads(lines[0]), p.stderr


def _check_line(j, name):
    c = j["config"]
    assert j["n_gpus"] == 1 and c["world_size_seen"] == 1 and c["dist_backend"] == "nccl"
    # round 5: the headline step CARRIES the gather; the statistics-only exchange is the secondary number
    assert j["value"] > 0 and j["value_full_gather"] == j["value"] and j["value_stats_exchange"] > 0
    assert "SURVIVING" in c["step"] and "all-gather" in c["step_stats_exchange"]
    # the survivor records: 28 B per surviving ray (24 B when the shard lost nothing), header included
    s = c["gather_survivors"]
    dense = s == c["rays_per_gpu"]
    exact = (16 + (24 if dense else 28) * s + 15) // 16 * 16
    roomy = exact if dense else (16 + 28 * min(c["rays_per_gpu"], s + max(1024, s // 16 + 1)) + 15) // 16 * 16
    # sized from the previous steps' counts with a margin -- never below what was packed; only the first step of the run
    # read its headers synchronously, nothing was short, nothing dropped; a dense shard went zero-copy
    assert exact <= c["gather_bytes_per_rank"] <= roomy, (exact, c["gather_bytes_per_rank"], roomy)
    assert c["gather_host_syncs"] == 1 and c["gather_overflows"] == 0 and c["gather_dropped"] == 0
    assert c["gather_zero_copy"] is dense
    assert abs(c["gather_floor_ms"] - c["gather_bytes_per_rank"] / 153e9 * 1e3) < 1e-9
    assert c["gather_link_gbs_measured"] is None and c["gather_only_ms"] >= 0        # one rank: nothing crosses a link
    # parity stays on the N > 1 line
    par = j["parity"]
    assert par["survivor_indices_equal"] and par["delay_max_rel_err"] <= 1e-10 and par["position_max_rel_err"] <= 1e-10
    assert par["path_max_rel_err"] <= 1e-10
    report(f"[rccl {name}] backend {c['dist_backend']}, world {c['world_size_seen']}: step with the survivor gather "
           f"{j['ms_per_step']:.3f} ms ({c['gather_bytes_per_rank']} B for {s} survivors of {c['rays_per_gpu']}, "
           f"{'zero-copy' if c['gather_zero_copy'] else 'packed'}), with the statistics exchange instead "
           f"{j['ms_per_step_stats_exchange']:.3f} ms; parity delay {par['delay_max_rel_err']:.1e} pos {par['position_max_rel_err']:.1e}")


def test_bench_n_gt_1_path_through_rccl_relay4():
    j, err = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--rays", "200000", "--cpu-sample", "20000"])
    _check_line(j, "relay4")
    assert j["config"]["gather_survivors"] == 200000          # every ray survives the relay: the dense form, 24 B/ray
    assert "NCCL version" in err or "RCCL version" in err, err[-1500:]


def test_bench_n_gt_1_path_through_rccl_c2_strided():
    j, _ = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "C2", "--shard", "strided", "--rays", "200000",
                   "--cpu-sample", "20000"])
    _check_line(j, "C2 strided")
    c = j["config"]
    assert c["chains"] == 11 and c["shard_layout"] == "strided"
    assert 0.4 * c["rays_per_gpu"] < c["gather_survivors"] < 0.6 * c["rays_per_gpu"]        # the mask stops half of the rays


@pytest.fixture(scope="module")
def hip():
    import torch
    import __graft_entry__
    from attosecondraytracing_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    __graft_entry__.ensure_built()
    _lib._BACKEND = None
    return _lib.get_backend()


@pytest.mark.parametrize("n", [0, 1, 63, 2048, 2049, 100_003])
def test_gpu_survivor_records_match_torch(hip, n):
    """art_pack_survivors against torch.nonzero + index_select: masked shards (explicit and implicit numbers), a shard
    with every slot alive (dense, 24 B/ray), a shard with nothing alive, an empty shard."""
    import torch
    from attosecondraytracing_amd import sharding
    g = torch.Generator(device="cpu").manual_seed(1234 + n)
    X, Y, O = (torch.randn(n, dtype=torch.float64, generator=g).to(hip.device) for _ in range(3))
    number = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64).to(hip.device)
    for case in ("masked", "dense", "none", "numbers"):
        if case == "masked" or case == "numbers":
            alive = (torch.rand(n, generator=g) < 0.37).to(torch.uint8).to(hip.device)
        elif case == "dense":
            alive = torch.ones(n, dtype=torch.uint8, device=hip.device)
        else:
            alive = torch.zeros(n, dtype=torch.uint8, device=hip.device)
        first, step = 7, 3
        sg = sharding.SurvivorGather(hip, n, 1, 0, specs=[(first, step, n)])
        nb = sg.start(0, X, Y, O, alive, number=number if case == "numbers" else None)
        num, gx, gy, go = sg.result(0)[0]
        idx = torch.nonzero(alive).reshape(-1)
        c, flags = sg.headers[0][0]
        assert c == idx.numel() and flags == (1 if (case != "numbers" and c == n and n > 0) else 0)     # an empty shard: header 0, 0
        assert nb == hip.survivor_bytes(c, bool(flags)) == (16 + (24 if flags else 28) * c + 15) // 16 * 16
        assert torch.equal(num, number[idx] if case == "numbers" else first + step * idx)
        for got, ref in ((gx, X), (gy, Y), (go, O)):
            assert torch.equal(got.view(torch.int64), ref[idx].view(torch.int64))


def test_gpu_survivor_records_refuse_numbers_beyond_int32(hip):
    import torch
    from attosecondraytracing_amd import _lib
    n = 10
    z = torch.zeros(n, dtype=torch.float64, device=hip.device)
    alive = torch.ones(n, dtype=torch.uint8, device=hip.device)
    send = torch.empty(hip.survivor_bytes(n), dtype=torch.uint8, device=hip.device)
    with pytest.raises(_lib.ArtError, match="int32"):
        hip.pack_survivors(alive, z, z, z, None, 2 ** 31 - 5, 1, send)
    with pytest.raises(_lib.ArtError, match="smaller"):
        hip.pack_survivors(alive, z, z, z, None, 0, 1, send[:64])


def test_bench_line_contract_single_gpu():
    """The driver's call (`python bench.py --gpus 1 --steps K --warmup W`), small: ONE JSON line with every field of the
    contract -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline /
    dtype / data / config.workload, `roofline` (counted or compulsory bytes, stated), `cpu_baseline` (the oracle, one
    thread, bounded sample), `parity` within 1e-10, the lazy-history rate beside the headline, the box state."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "ART_FORCE_DIST")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--rays",
                        "400000", "--cpu-sample", "20000"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["metric"] == "ray-surface intersections/s" and j["unit"] == "intersections/s" and j["value"] > 1e9
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (1, 5, 2) and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["dtype"] == "f64" and j["data"] == "synthetic"
    assert abs(j["value"] - 400000 * 4 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    assert "relay4" in j["config"]["workload"] and j["config"]["hip_graph"] is True
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["frac_basis"] in ("counted", "compulsory")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac_compulsory"] < 1.0
    assert r["compulsory_bytes"] == 400000 * (65 + 4 * 65 + 24) + 176 * ((400000 + 255) // 256)     # every ray survives the relay
    assert r["kernel"].startswith("k_trace_scene<false") and r["kernel_ms"] > 0 and len(r["timed_region_launches"]) == 2
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 1e4 and "oracle" in c["sample"]
    par = j["parity"]
    assert par["survivor_indices_equal"] and max(par["delay_max_rel_err"], par["position_max_rel_err"], par["path_max_rel_err"]) <= 1e-10
    assert j["value_lazy_history"] > j["value"] * 0.9 and j["value_full_gather"] is None
    assert j["box"] is None or "idle_before_run" in j["box"]
