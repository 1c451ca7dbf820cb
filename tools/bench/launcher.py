"""The launcher half of bench.py: starts one worker process per GPU and relays rank 0's JSON line; the box-state helper.
Nothing here imports torch or touches the GPU."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)

# =========================================================================================== launcher (no GPU, no torch)
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def multi_process_env(env):
    """What every process of a multi-process GPU job needs in its environment on this pool.
    HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts' kernel driver supports only dmabuf-based IPC.  RCCL opens its peers' buffers
    through hipIpcGetMemHandle / hipIpcOpenMemHandle (and so does any CUDA-tensor sharing between processes); with the
    runtime's LEGACY IPC mode (the default of some ROCr builds) those calls fail with `hipIpcGetMemHandle: invalid
    argument` as soon as two ranks on one node set up their xGMI / P2P transport -- a one-rank group never gets there.
    The image exports the variable already; it is set here too (setdefault: an explicit choice of the caller wins) so that
    a worker started from a scrubbed environment behaves the same.  examples/sharded_trace.py does the same."""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def launch_workers(n, argv, time_limit=1500.0):
    """Start n workers of this script, one per GPU; relay rank 0's stdout (the JSON line); fail if any worker fails or
    the job exceeds `time_limit` seconds of wall clock (all workers are killed, exit code 4).  The workers are fresh
    child processes: nothing that has touched the GPU is ever re-exec'ed.
    Runs before any torch.cuda / HIP call of this process: nothing here initialises the GPU."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = multi_process_env(dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                                     MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
        out = subprocess.PIPE if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=out))
    # rank 0's stdout is drained while the workers run (a reader thread): a rank 0 that printed more than the pipe holds
    # would otherwise block in write() while this loop waits for it to exit
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # watch all workers: if one dies, the others would sit in a collective until RCCL's own timeout -- end them at once
    failed = timed_out = False
    t_start = time.monotonic()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        if time.monotonic() - t_start > time_limit:
            failed = timed_out = True
            log(f"[bench] the {n}-rank job exceeded its wall-clock limit of {time_limit:.0f} s: killing all workers")
            break
        time.sleep(0.2)
    if failed:
        time.sleep(1.0)                      # let the failing rank's traceback reach stderr first
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10.0)
    line = b"".join(chunks)
    sys.stdout.write(line.decode(errors="replace"))
    sys.stdout.flush()
    if any(rc != 0 for rc in rcs):
        log(f"[bench] worker exit codes {rcs}: failing")
        return 4 if timed_out else 1
    return 0


# =========================================================================================== box state (read-only queries)
SMI_ARGS = ["rocm-smi", "--showclocks", "--showperflevel", "--showpower", "--showmaxpower", "--showmemorypartition",
            "--showcomputepartition", "--showtemp", "--json"]
_SMI_HELPER = r"""
import subprocess, sys
for line in sys.stdin:
    try:
        out = subprocess.run(%r, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=30).stdout.decode(errors="replace")
    except Exception as e:
        out = "{}"
    sys.stdout.write(out.replace("\n", " ") + "\n")
    sys.stdout.flush()
""" % (SMI_ARGS,)


class BoxState:
    """rocm-smi queries (clocks, power, power cap, partition modes: sysfs reads, no queue touched) through a helper
    process that is started BEFORE this process initialises the GPU: nothing is ever exec'ed from a GPU-initialised
    process (a rule of the pool).  Under rocprofv3 the profiler's preloaded library has initialised the GPU before main()
    runs, so no helper is started there: tools/prof.sh records the box state beside the passes itself."""

    def __init__(self):
        self.p = None
        if "rocprof" in os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in os.environ:
            return
        try:
            self.p = subprocess.Popen([sys.executable, "-c", _SMI_HELPER], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL)
        except OSError:
            self.p = None

    def ask(self):
        """Start one query; returns immediately (read it with `answer`)."""
        if self.p is None:
            return False
        try:
            self.p.stdin.write(b"q\n")
            self.p.stdin.flush()
            return True
        except OSError:
            self.p = None
            return False

    def ready(self):
        import select
        return self.p is None or bool(select.select([self.p.stdout], [], [], 0)[0])

    def answer(self, card):
        """Compact dict of the pending query's fields for device `card` (clock levels, power, partitions), or None."""
        if self.p is None:
            return None
        try:
            out = self.p.stdout.readline().decode(errors="replace")
            j = json.loads(out[out.index("{"):])
            c = j.get(f"card{card}", next(iter(j.values())))
            return {k: v for k, v in c.items()
                    if any(t in k.lower() for t in ("clock", "power", "partition", "performance", "temperature (sensor junction)",
                                                    "temperature (sensor memory)"))}
        except Exception as e:    # noqa: BLE001 -- a diagnostic must never cost the result line
            return {"error": repr(e)[:200]}

    def close(self):
        if self.p is not None:
            try:
                self.p.stdin.close()
                self.p.wait(timeout=5)
            except Exception:     # noqa: BLE001
                pass
            self.p = None


