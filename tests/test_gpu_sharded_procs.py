"""GPU (-m gpu): two worker processes share the one GPU of the test box, each traces its index-range shard of a
point source with the HIP kernels (source generated on the device from the global ray index), the read-out is
staged to the host and exchanged over gloo (RCCL refuses two ranks on one device); the assembled result must equal
the single-process run bit for bit.  Exercises the one-process-per-GPU code path end to end except RCCL itself."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port nobody listens on right now (the rendezvous of the worker processes)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _trace_shard(rank, world, n_total):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.cuda.set_device(0)
    from attosecondraytracing_amd import _lib, sharding, ModuleGeometry as mgeo
    from attosecondraytracing_amd.bundle import RayBundle
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    from test_sharding_gloo import _scene
    be = _lib.get_backend()
    chain = _scene(n_total)
    lo, hi = sharding.shard_range(n_total, rank, world)
    src = RayBundle.allocate(hi - lo, backend=be)
    rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    be.make_source(0, 0.025, rot, np.zeros(3), lo, hi - lo, n_total, src.view())
    src.intensity = torch.ones(hi - lo, dtype=torch.float64, device=be.device)
    out = mp.RayTracingCalculation(src, chain.optical_elements)
    det = mdet.Detector(np.zeros(3), np.array([1900.0, 30.0, 0.0]), np.array([-0.9, -0.1, 0.2]))
    r = det.readout(out[-1], sync=False)
    # the survivor records of this shard, packed by the HIP kernels (art_pack_survivors); they travel over gloo below
    send = torch.empty(be.survivor_bytes(hi - lo), dtype=torch.uint8, device=be.device)
    be.pack_survivors(out[-1].alive, r["X"], r["Y"], r["opl"], None, lo, 1, send)
    hdr = send[:16].view(torch.int64).cpu()
    used = be.survivor_bytes(int(hdr[0]), bool(int(hdr[1]) & 1))
    return (torch.stack([r["X"], r["Y"], r["opl"]]).cpu(), out[-1].alive.cpu(), r["stats_dev"].cpu(), hdr, send[:used].cpu())


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from attosecondraytracing_amd import sharding
        XYO, alive, stats, hdr, rec = _trace_shard(rank, world, n_total)
        g = sharding.allreduce_stats(stats, torch.device("cpu"))
        full, al = sharding.gather_readout(XYO[0], XYO[1], XYO[2], alive, 0)
        # the survivor-only gather with the device-packed records: header all-gather, ONE gather of max bytes
        hdrs = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(hdrs, hdr)
        nb = max((16 + (24 if int(h[1]) & 1 else 28) * int(h[0]) + 15) // 16 * 16 for h in hdrs)
        pad = torch.zeros(nb, dtype=torch.uint8)
        pad[:rec.numel()] = rec
        recv = [torch.zeros(nb, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, recv, dst=0)
        if rank == 0:
            specs = [sharding.shard_spec(n_total, k, world) for k in range(world)]
            parts = [sharding.decode_survivors(recv[k], int(hdrs[k][0]), int(hdrs[k][1]), specs[k]) for k in range(world)]
            surv = [torch.cat([p[j] for p in parts]).numpy() for j in range(4)]
            q.put((g.numpy(), full.numpy(), al.numpy(), surv, [int(h[1]) for h in hdrs]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu_match_single_process():
    n_total = 200_003
    ref_xyo, ref_alive, ref_stats, _, _ = _trace_shard(0, 1, n_total)
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, 2, port, n_total, q)) for rk in range(2)]
    for p in procs:
        p.start()
    stats, XYO, alive, surv, dense = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(alive, ref_alive.numpy())
    m = alive.astype(bool)
    assert 0 < m.sum() < n_total
    assert np.array_equal(XYO[:, m], ref_xyo.numpy()[:, m])
    # the survivor records, packed on the device by each rank: the single-process read-out of the survivors, numbers included;
    # rank 0's inner shard passes the mask entirely (dense: no number section), rank 1's does not
    assert np.array_equal(surv[0], np.nonzero(m)[0]) and np.array_equal(np.stack(surv[1:]), ref_xyo.numpy()[:, m])
    assert dense == [1, 0], dense
    rs = ref_stats.numpy()
    assert stats[0] == rs[0]
    for k in (2, 3, 4, 5, 12, 13):
        assert stats[k] == rs[k]
    for k in (1, 6, 7, 8, 11):
        assert abs(stats[k] - rs[k]) <= 1e-12 * abs(rs[k])
