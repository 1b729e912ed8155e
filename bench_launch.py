"""`python3 bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.

The parent process never touches the GPU (no HIP call, no torch.cuda.is_available(), no gpk.Context, no RCCL): it only counts
the devices (torch.cuda.device_count(), which does not initialise the runtime on this image), starts N fresh child processes
of the SAME command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- children are new processes
started with subprocess.Popen, nothing is exec'd from a GPU-initialised process anywhere -- relays their stderr (inherited),
keeps rank 0's stdout, and prints rank 0's compact JSON line as its own LAST stdout line.  Exit code = the worst child's.

The children run bench.main() exactly as if `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` had
started them (that launch keeps working: main() only comes here when WORLD_SIZE is absent from the environment).

Pure Python, importable anywhere (tests/test_bench_flow.py runs the bare command on CPU over gloo).
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_devices():
    """Number of GPUs this process could use, WITHOUT initialising the HIP runtime (device_count() reads the topology only);
    0 when torch is missing or there is no GPU."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:                                             # noqa: BLE001 -- no torch / no driver: the children will say so
        return 0


def local_rank_of(rank, n_ranks, n_dev, backend):
    """RCCL needs one device per rank; the gloo stand-in flow (GPK_BENCH_BACKEND=gloo, collectives staged through the host:
    tools/bench_two_ranks_one_gpu.sh, the -m gpu test of the bare command on a 1-GPU box) may put several ranks on one device."""
    if backend == 'gloo' and n_dev < n_ranks:
        return rank % max(n_dev, 1)
    return rank


def launch(n_ranks, argv=None, script=None, env=None, timeout=None, out=None, err=None):
    """Start n_ranks children of `python <script> <argv...>`; returns the exit code to leave with.
    rank 0's stdout is kept and re-printed (its last '{' line LAST); the other ranks' stdout goes to stderr with a rank prefix."""
    out = out or sys.stdout
    err = err or sys.stderr
    argv = list(sys.argv[1:] if argv is None else argv)
    script = script or os.path.abspath(sys.argv[0])
    base = dict(os.environ if env is None else env)
    backend = base.get('GPK_BENCH_BACKEND', 'nccl')
    n_dev = visible_devices()
    if backend != 'gloo' and n_dev < n_ranks:
        print(f'bench.py: --gpus {n_ranks} needs {n_ranks} visible GPUs, torch.cuda.device_count() = {n_dev} '
              f'(one rank per GPU over RCCL; nothing was launched)', file=err, flush=True)
        return 2
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    base.update(WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR='127.0.0.1',
                MASTER_PORT=str(base.get('GPK_BENCH_MASTER_PORT') or free_port()), GPK_BENCH_SELF_LAUNCHED='1')
    timeout = float(base.get('GPK_BENCH_LAUNCH_TIMEOUT', '3300')) if timeout is None else timeout
    procs, rank0_lines, relays = [], [], []

    def relay(stream, rank):
        for line in stream:
            if rank == 0:
                rank0_lines.append(line)
            else:                                                 # a peer has nothing to say on stdout (RCCL banners at most)
                err.write(f'[rank {rank} stdout] {line}')
        stream.close()

    try:
        for rank in range(n_ranks):
            e = dict(base, RANK=str(rank), LOCAL_RANK=str(local_rank_of(rank, n_ranks, n_dev, backend)))
            # own process group per child: a stuck child is ended by ITS pid / group, never by a pattern
            p = subprocess.Popen([sys.executable, script] + argv, env=e, stdout=subprocess.PIPE, stderr=None, text=True,
                                 start_new_session=True)
            procs.append(p)
            t = threading.Thread(target=relay, args=(p.stdout, rank), daemon=True)
            t.start()
            relays.append(t)
        deadline = time.monotonic() + timeout
        grace = None                                              # once one rank has left with an error the others get a minute
        while any(p.poll() is None for p in procs):
            now = time.monotonic()
            if grace is None and any(p.poll() not in (None, 0) for p in procs):
                grace = now + float(base.get('GPK_BENCH_PEER_GRACE', '60'))
            if now > deadline or (grace is not None and now > grace):
                why = 'launch timeout' if now > deadline else 'a rank failed and its peers did not leave'
                print(f'bench.py: {why}: ending the remaining ranks', file=err, flush=True)
                _end(procs)
                break
            time.sleep(0.2)
    except KeyboardInterrupt:
        _end(procs)
        raise
    for t in relays:
        t.join(timeout=10)
    codes = [p.returncode if p.returncode is not None else 124 for p in procs]
    json_lines = [l for l in rank0_lines if l.lstrip().startswith('{')]
    for l in rank0_lines:                                         # whatever else rank 0 printed (library banners), before the line
        if not (json_lines and l is json_lines[-1]):
            out.write(l)
    if json_lines:
        out.write(json_lines[-1] if json_lines[-1].endswith('\n') else json_lines[-1] + '\n')
    out.flush()
    worst = 0
    for c in codes:
        c = 128 - c if c < 0 else c                               # ended by a signal
        worst = max(worst, c)
    if worst:
        print(f'bench.py: rank exit codes {codes}', file=err, flush=True)
    if not json_lines and worst == 0:
        print('bench.py: rank 0 printed no JSON line', file=err, flush=True)
        worst = 5
    return worst


def _end(procs):
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGTERM)                  # the child's own session (start_new_session): exactly what we started
            except (ProcessLookupError, PermissionError):
                pass
    t0 = time.monotonic()
    while any(p.poll() is None for p in procs) and time.monotonic() - t0 < 10:
        time.sleep(0.2)
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()
