"""CPU: `bench.py --gpus N` as the driver calls it (no torchrun, no WORLD_SIZE) must start N ranks by itself, run the
N > 1 step on every rank and report the world size the process group really had.  The ranks here are CPU processes over
gloo with the test-only twin backend installed through bench.py's test hook; the launcher, the rendezvous, the
per-step all-gather, the full read-out gather and the JSON contract are the code the GPU run executes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ART_DIST_BACKEND="gloo", ART_BENCH_BACKEND_HOOK="twin_backend:install", OMP_NUM_THREADS="2",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""))
    env.update(extra)
    return env


def _run(args, **extra):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(**extra), capture_output=True,
                          text=True, timeout=600)


def test_bench_gpus_2_spawns_two_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rays", "5000", "--cpu-sample", "0"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["world_size_seen"] == 2 and j["scaling"] == "weak"
    assert j["steps"] == 3 and j["warmup"] == 1 and j["unit"] == "intersections/s"
    # both N > 1 numbers are on the line: the statistics + sample step and the full-gather step
    assert j["value"] > 0 and j["value_full_gather"] > 0 and j["ms_per_step_full_gather"] > 0
    assert "all-gather" in j["config"]["step"] and "gather" in j["config"]["step_full_gather"]
    # round 5: the headline step carries the survivor gather (zero-copy: every ray survives the relay, the read-out writes
    # straight into the send buffers), the statistics-only exchange is the secondary number; one peer link was measured
    c = j["config"]
    assert "SURVIVING" in c["step"] and j["value_full_gather"] == j["value"] and j["value_stats_exchange"] > 0
    assert c["gather_zero_copy"] is True and c["gather_overflows"] == 0 and c["gather_dropped"] == 0 and c["gather_host_syncs"] == 1
    assert c["gather_bytes_per_peer"] == [16 + 24 * 5000] and c["gather_link_gbs_measured"] > 0
    # whole-job aggregate: 2 ranks x 5000 rays x 4 mirrors per step
    inter = 2 * 5000 * 4
    assert abs(j["value"] - inter * 3 / (j["ms_per_step"] * 3e-3)) <= 1e-6 * j["value"]


def test_bench_gather_in_tiles_two_ranks():
    """--gather-tiles T: the step of a zero-copy shard is traced as T launches over consecutive slot ranges of the same
    bundles and each range's records leave behind it; from the third step on (the ranks tile once the headers of two steps
    earlier say `dense`) -- same records on rank 0 (the worker asserts them), nothing short."""
    p = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--rays", "5000", "--cpu-sample", "0", "--gather-tiles", "3"])
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    c = j["config"]
    assert c["gather_tiles"] == 3 and c["gather_tiled_steps"] >= 4 and c["gather_zero_copy"] is True
    assert c["gather_overflows"] == 0 and c["gather_dropped"] == 0 and j["value"] > 0


def test_bench_multichain_config_two_ranks():
    """A loop-list configuration (C2: 11 chains in one scene-table launch) through the same launcher."""
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "C2", "--rays", "2000", "--cpu-sample", "0"])
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert j["n_gpus"] == 2 and j["config"]["chains"] == 11 and j["config"]["name"] == "C2"


def test_bench_strided_shards_balance_a_masked_scene():
    """C2 on two ranks: with contiguous shards rank 1 (the outer half of the Vogel spiral) is stopped by the mask almost
    entirely; with strided shards both ranks carry the same load."""
    import re
    share = {}
    for layout in ("blocks", "strided"):
        p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "C2", "--rays", "2000", "--cpu-sample", "0",
                  "--shard", layout])
        assert p.returncode == 0, p.stderr[-3000:]
        j = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
        assert j["config"]["shard_layout"] == layout
        m = re.search(r"= (\d+) intersections/step on rank 0, (\d+) on all", j["config"]["workload"])
        share[layout] = int(m.group(1)) / int(m.group(2))
    assert abs(share["strided"] - 0.5) < 0.01 and share["blocks"] > 0.6, share


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    # as a torchrun worker (WORLD_SIZE set) whose group has 1 rank while --gpus says 2: must fail, not report n_gpus 1
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--rays", "1000", "--cpu-sample", "0"],
             WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert "process group has 1 rank" in p.stderr


def test_bench_single_rank_line_on_cpu_twin():
    p = _run(["--steps", "2", "--warmup", "1", "--rays", "4000", "--cpu-sample", "0"])
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert j["n_gpus"] == 1 and j["value_full_gather"] is None and j["config"]["name"] == "relay4"


def test_launcher_ends_all_ranks_when_one_fails():
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "2000", "--cpu-sample", "0"],
             ART_BENCH_BACKEND_HOOK="twin_backend:install_failing_on_rank_1")
    assert p.returncode != 0 and time.time() - t0 < 120
    assert "rank 1 fails on purpose" in p.stderr and "worker exit codes" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_profile_of_another_build_is_dropped(monkeypatch):
    """bench.py quotes counted HBM bytes only from a profile taken on THIS build (source hash of csrc/* + the C header);
    otherwise the line falls back to the compulsory bytes it computes itself and names the dropped profile."""
    import glob
    sys.path.insert(0, ROOT)
    import bench
    import tools.source_hash as sh
    profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_relay4.json")))
    assert profs, "no committed relay4 profile"
    newest = json.load(open(profs[-1]))
    kernel = next(k for k in newest["per_launch"] if k.startswith("k_trace_"))
    prefix = kernel.split("<")[0] + "<"
    monkeypatch.setattr(sh, "source_hash", lambda root=None: newest["source_hash"])
    hit, note = bench.profiled_traffic("relay4", prefix, newest["rays_per_gpu"])
    assert hit is not None and hit[1].endswith(os.path.basename(profs[-1])) and note is None
    monkeypatch.setattr(sh, "source_hash", lambda root=None: "0123456789abcdef")
    hit, note = bench.profiled_traffic("relay4", prefix, newest["rays_per_gpu"])
    assert hit is None and "dropped" in note and os.path.basename(profs[-1]) in note
    # the tree's own hash is a pure function of the four source files
    assert len(sh.__dict__["FILES"]) == 4 and all(os.path.exists(os.path.join(ROOT, f)) for f in sh.FILES)


def test_masked_configurations_find_their_profile(monkeypatch):
    """C2 / C3 contain a mask, so the library launches the two-rays-per-lane body (k_trace_scene2<false, 4>): the bench
    line's kernel pattern must match that name in the committed profile of the same build."""
    import glob
    sys.path.insert(0, ROOT)
    import bench
    import tools.source_hash as sh
    for cfg in ("C2", "C3"):
        profs = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{cfg}.json")))
        assert profs, f"no committed {cfg} profile"
        newest = json.load(open(profs[-1]))
        assert any(k.startswith("k_trace_scene2<false") for k in newest["per_launch"]), list(newest["per_launch"])
        monkeypatch.setattr(sh, "source_hash", lambda root=None, h=newest["source_hash"]: h)
        hit, note = bench.profiled_traffic(cfg, r"k_trace_scene2?<false", newest["rays_per_gpu"])
        assert hit is not None and hit[2].startswith("k_trace_scene2<false") and hit[0] > 1e8 and note is None


def test_bench_gpus_8_over_gloo():
    """The driver's largest case, rehearsed with CPU ranks: eight ranks (an odd ray count per rank, strided shards of a
    masked loop-list scene), every rank in the all-gather and in the ONE survivor gather."""
    p = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--rays", "1501", "--config", "C3", "--shard", "strided",
              "--cpu-sample", "0"], OMP_NUM_THREADS="1")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["world_size_seen"] == 8 and j["config"]["shard_layout"] == "strided"
    assert j["value"] > 0 and j["value_full_gather"] > 0
    # the gathered records are the survivors of the whole job: about two thirds of 8 x 1501 rays pass C3's mask
    assert 0.5 * 8 * 1501 < j["config"]["gather_survivors"] < 0.8 * 8 * 1501
    # which of the two numbers the scaling target is judged on, and the link-rate bound of the other one
    c = j["config"]
    assert abs(c["gather_floor_ms"] - c["gather_bytes_per_rank"] / 153e9 * 1e3) <= 1e-9 and "xGMI" in c["gather_floor_note"]
    assert "the number the >= 6x scaling target is judged on" in p.stderr and "gather_floor_ms" in p.stderr


def test_launcher_wall_clock_limit_kills_all_workers():
    """A rank that hangs (here: before the rendezvous) must not hold the job: the launcher's wall-clock limit ends every
    worker -- fresh child processes, nothing re-exec'ed -- and the exit code says so."""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "2000", "--cpu-sample", "0", "--time-limit", "20"],
             ART_BENCH_BACKEND_HOOK="twin_backend:install_hanging_on_rank_1")
    assert p.returncode == 4 and time.time() - t0 < 90, (p.returncode, p.stderr[-2000:])
    assert "exceeded its wall-clock limit" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_process_group_timeout_ends_a_rank_that_waits_alone():
    """The collectives' own timeout (--pg-timeout): rank 0 waits in its first collective for a rank that never joins it
    and gives up with an error instead of waiting for ever."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--rays", "1000", "--cpu-sample", "0", "--pg-timeout", "5",
              "--time-limit", "120"], ART_BENCH_BACKEND_HOOK="twin_backend:install_sleeping_on_rank_1")
    assert p.returncode == 1, (p.returncode, p.stderr[-2000:])
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_as_torchrun_workers():
    """The driver's N > 1 launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- bench.py is then a WORKER (RANK / WORLD_SIZE set), not the launcher: one JSON
    line from rank 0, the world size the group really had, the step with the survivor gather as `value`."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--rays", "3000", "--cpu-sample", "0"],
                       env=_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["world_size_seen"] == 2 and j["scaling"] == "weak"
    assert j["steps"] == 3 and j["warmup"] == 1
    assert j["value"] == j["value_full_gather"] > 0 and j["value_stats_exchange"] > 0
