#!/usr/bin/env python3
"""Benchmark of the hot path: Gauss-Newton steps/sec (+ L2 error) of the GP solver for NonLinElliptic2d on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU -- bench_launch.py; the parent never
                                                            touches the GPU and relays rank 0's line as its last stdout line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          (the same ranks started by torchrun: WORLD_SIZE is set, nothing is re-launched)

`value` at N = 1 is measured on BASELINE config 2 (N_domain=4000, N_boundary=400: the configuration the metric is quoted on, it
fits one GPU); with N > 1 ranks `value` is the SHARDED BASELINE config 5 (N_domain=16000, Theta of order 34000: panel-sharded
Cholesky + column-sharded Gauss-Newton step over RCCL, strong scaling) -- `value_workload` ("c2" / "c5") says which, machine-readably,
and a sharded run that does not complete leaves `value: null`, never another workload's figure.  Config 5 on ONE GPU (the 1-GPU
point of that series) is in every N = 1 line under `sharded_config`; N independent config-2 replicas in every N > 1 line under
`replicas_c2`.

The LAST stdout line is compact (bench_line.py, ~5 KB: contract fields, roofline, cpu_baseline, parity, one small object per
secondary workload); the full result object goes to bench_detail.json and stderr.

A "step" is one Gauss-Newton step of the reference's GN_method (src/PDEs.py:117-127): Hessian_GN + grad_loss + linear
solve + update + one loss evaluation, executed as ONE gpk_gn_step call (TRSM with n_z+1 right-hand sides + SYRK + Cholesky of H + triangular solve + update, and the loss of
the iterate it starts from by true substitution), all operands resident in HBM -- the product's default sequence.  Nothing is cached across steps (the dense "F1" formulation of SURVEY 8d).
Rank 0 prints ONE JSON line.  `roofline.achieved` is measured live with HIP events recorded inside the timed steps on
the stream the kernels run on; `roofline.traffic` is read from the newest stored PMC pass under profiles/ and says so
(`traffic_source`); `cpu_baseline` times the CPU oracle (reference operation sequence) on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X datasheet fp64 matrix rate (= 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz); the local
                                  # guides state no fp64 peak.  v_mfma_f64_16x16x4_f64 issue-rate ubench: ~74.
HBM_PEAK_GBS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md:35 (6290 GB/s measured achievable)

WORKLOADS = {
    # name: (N_domain, N_boundary, GNsteps of the reference config, description)
    'c1': (900, 124, 4, 'NonLinElliptic2d Gaussian sigma=0.2 N_domain=900 N_boundary=124 (BASELINE config 1)'),
    'c2': (4000, 400, 4, 'NonLinElliptic2d Gaussian sigma=0.2 N_domain=4000 N_boundary=400 (BASELINE config 2)'),
    'c5': (16000, 2000, 4, 'NonLinElliptic2d Gaussian sigma=0.2 N_domain=16000 N_boundary=2000 (BASELINE config 5)'),
    'n10k': (10000, 1000, 4, 'NonLinElliptic2d Gaussian sigma=0.2 N_domain=10000 N_boundary=1000 (north-star target size)'),
    # the two non-elliptic BASELINE configurations (run_system below): same step, other Gram layouts / Gauss-Newton systems
    'c3': (2000, 400, 8, 'Burgers1d anisotropic_Gaussian sigma=[0.3,0.05] N_domain=2000 N_boundary=400 (->399) nugget 1e-5 seed 0 (BASELINE config 3)'),
    'c4': (1600, 200, 8, 'DarcyFlow2d inverse problem Gaussian sigma=0.2 N_domain=1600 N_boundary=200 N_data=60 noise=1e-3 nugget 1e-8 seed 9999 (BASELINE config 4)'),
}
TIMED_SEQUENCE = {False: 'gpk_gn_step per step, exact in-step loss (the product default: src/PDEs.py _gn_iterate)',
                  True: 'gpk_gn_step + gpk_gn_loss per step (GPK_SEPARATE_LOSS=1, the round-4 sequence)'}
PARITY_TOL = 1e-6                 # north star: device iterate within 1e-6 relative of the reference path on the same points
SIGMA, ALPHA, M_EXP = 0.2, 1.0, 3.0


def u_true(x1, x2):
    return np.sin(np.pi * x1) * np.sin(np.pi * x2) + 2 * np.sin(4 * np.pi * x1) * np.sin(4 * np.pi * x2)


def rhs(x1, x2):        # -Laplace(u*) + alpha u*^m, main_NonLinElliptic2d.py:60-64 of the reference
    return (2 * np.pi ** 2 * np.sin(np.pi * x1) * np.sin(np.pi * x2)
            + 64 * np.pi ** 2 * np.sin(4 * np.pi * x1) * np.sin(4 * np.pi * x2) + ALPHA * u_true(x1, x2) ** M_EXP)


def synthetic_problem(Nd, Nb):
    """numpy.random.seed(0), the reference sampler's draw order, then the N(0,1) initial guess."""
    from src.sample_points import sampled_pts_rdm
    np.random.seed(0)
    Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]), time_dependent=False)
    z0 = np.random.normal(0.0, 1.0, Nd)
    return Xd, Xb, rhs(Xd[:, 0], Xd[:, 1]), u_true(Xb[:, 0], Xb[:, 1]), z0


def test_grid(n=60):
    xx = np.linspace(0, 1, n)
    XX, YY = np.meshgrid(xx, xx)
    return np.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)


def f1_flops(N, nz):
    return float(N) * N * nz + float(N) * nz * nz + nz ** 3 / 3.0


def syrk_executed_flops(N, nz, tile=64, bk=16):
    """Flops the SYRK launch Hb = S^T S actually executes (lower tiles only).  gn_step stores unknown j in column
    nz-1-j, so column c < nz of S is zero above row nz-1-c and the K loop of the tile with columns [n0, n0+tile) starts at
    floor(max(0, nz - (n0+tile)) / bk) * bk  (csrc/gpk_gemm.hip, GemmArgs::lead).  Dense count: N (nz+1)^2."""
    nc = nz + 1
    nt = (nc + tile - 1) // tile
    total = 0.0
    for tn in range(nt):
        n0 = tn * tile
        bn = min(tile, nc - n0)
        k0 = (max(0, nz - (n0 + tile)) // bk) * bk
        rows_m = nc - n0                      # all tile rows tm >= tn: columns n0 .. nc-1 of the lower triangle
        total += 2.0 * bn * rows_m * (N - k0)
    return total


def trsm_dinv_executed_flops(N, nz, db=1024, num_cu=256, nb=64, bk=16, c0=0, c1=None):
    """(flops, launches) executed by the solve phase S = L^{-1}[A | F] of one step (gpk_i_trsm_left_dinv with lead = n_z, nrhs =
    n_z + 1) -- mirrors csrc/gpk_factor.hip / gpk_gemm.hip: recursion split at multiples of the inverted-block size db, active
    column range rounded down to 64, per 64-wide column tile the K loop starts at floor(max(0, lz - (n0 + 64)) / 16) * 16, the
    triangular leaf products stop at each row tile's last row.  All of it is ONE kernel, gemm_f64_kernel<.., NN>.
    c0, c1: the column shard [c0, c1) of the n_z + 1 right-hand sides a rank of the sharded step solves (gpk_trsm_dinv is then called
    with nrhs = c1 - c0 and lead = n_z - c0); default: all columns."""
    c1 = nz + 1 if c1 is None else c1
    nrhs, lead = c1 - c0, max(nz - c0, 0)
    acc = [0.0, 0]

    def gemm(m, n, k, lz, tri):
        if m <= 0 or n <= 0:
            return
        acc[1] += 1
        t64 = ((m + 63) // 64) * ((n + 63) // 64)
        if tri:
            t64 //= 2
        bm = 32 if (t64 < 2 * num_cu and m >= 64) else 64
        for n0 in range(0, n, 64):
            bn = min(64, n - n0)
            k0 = (max(0, lz - (n0 + 64)) // bk) * bk if lz > 0 else 0
            if tri:
                for m0 in range(0, m, bm):
                    acc[0] += 2.0 * min(bm, m - m0) * bn * max(0, min(k, m0 + bm) - k0)
            else:
                acc[0] += 2.0 * m * bn * max(0, k - min(k0, k))

    def rec(n, row0):
        if n <= 0:
            return
        clo = lead - (row0 + n)
        clo = (clo // nb) * nb if clo > 0 else 0
        if clo >= nrhs:
            return
        if n <= db:
            gemm(n, nrhs - clo, n, max(lead - row0 - clo, 0), True)
            return
        n1 = ((n // 2 + db - 1) // db) * db
        if n1 >= n:
            n1 = db
        rec(n1, row0)
        cu = lead - (row0 + n1)
        cu = (cu // nb) * nb if cu > 0 else 0
        if cu < nrhs:
            gemm(n - n1, nrhs - cu, n1, max(lead - row0 - cu, 0), False)
        rec(n - n1, row0 + n1)

    rec(N, 0)
    return acc[0], acc[1]


def stored_pmc_traffic(which='syrk', workload='c2'):
    """`roofline.traffic` cannot be measured inside this process (PMC counters need a rocprofv3 --pmc pass of their own,
    tools/profile_round.sh): it is READ from the newest committed PMC pass OF THE SAME WORKLOAD -- profiles/rNN_pmc_<which>.json for
    BASELINE config 2 (the historical name), profiles/rNN_pmc_<which>_<workload>.json for every other one; the file's own `workload`
    field must agree -- HBM-side bytes per Gauss-Newton step of that kernel family (FETCH_SIZE x2 + WRITE_SIZE,
    MI355X_MICROARCH.md).  No file for the workload: traffic is null (never another workload's bytes)."""
    import glob
    suffix = '' if workload == 'c2' else f'_{workload}'
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_pmc_{which}{suffix}.json')))
    for path in reversed(files):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get('workload', 'c2') != workload:
            continue
        v = d.get('hbm_bytes_per_step', d.get('hbm_bytes_per_launch'))
        if v is not None:
            per = 'step (sum over the launches of this kernel in one Gauss-Newton step)' if 'hbm_bytes_per_step' in d else 'launch'
            return v, (f'stored PMC pass profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of the same '
                       f'kernel at workload {workload}), bytes per {per}; not measured in this run')
    return None, f'no stored PMC pass for workload {workload}'


def emit(out):
    """The full result object -> bench_detail.json (repo root; also gpurun_out/ when that exists, so that it travels back from a GPU
    box) and stderr; the LAST stdout line is its compact form (bench_line.compact: <= ~5 KB, the driver keeps an 8 KB tail)."""
    import bench_line
    text = json.dumps(out, indent=1, default=str)
    for d in ([os.environ['GPK_BENCH_DETAIL_DIR']] if os.environ.get('GPK_BENCH_DETAIL_DIR') else [ROOT, os.path.join(ROOT, 'gpurun_out')]):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, 'bench_detail.json'), 'w') as fh:
                    fh.write(text + '\n')
            except OSError:
                pass
    print('bench.py: full result object (also in bench_detail.json):\n' + json.dumps(out, default=str), file=sys.stderr, flush=True)
    try:                                                          # RCCL prints a version banner through C stdio (fully buffered when stdout is
        C.CDLL(None).fflush(None)                                 # a pipe): flush it NOW so that the JSON line is the last thing on stdout
    except Exception:                                             # noqa: BLE001
        pass
    print(bench_line.compact(out), flush=True)


class AbortWatch:
    """Keeps the primary result alive while the SECONDARY (sharded) run is in flight.
    * N > 1 ranks: a rank that fails in the middle of that run cannot tell peers that are blocked inside a collective -- they
      would sit there until the RCCL timeout and the primary value would be lost.  Every rank therefore polls a key of the
      rendezvous store from a daemon thread; the failing rank sets it, and on seeing it rank 0 prints the one JSON line
      (primary value + the error) and every rank leaves the process without touching the process group again.
    * any N: a deadline (GPK_SHARDED_TIMEOUT seconds, default 420; the run takes 10-20 s) -- a hang of the secondary run (a
      collective that never completes on some fabric) ends the same way instead of taking the whole benchmark with it."""
    KEY = 'gpk_bench_abort'

    def __init__(self, rank, primary_out, use_store=True):
        import threading
        self.rank, self.out = rank, primary_out
        self.store = None
        if use_store:
            import torch.distributed as dist
            self.store = dist.distributed_c10d._get_default_store()
        self.deadline = time.monotonic() + float(os.environ.get('GPK_SHARDED_TIMEOUT', '600'))
        self.done = threading.Event()
        self.lock = threading.Lock()                              # main thread and watcher may both get here: ONE line only
        self.thread = threading.Thread(target=self._poll, daemon=True)
        self.thread.start()

    def _finish(self, msg):
        with self.lock:                                           # never released: the process ends inside
            if self.rank == 0 and self.out is not None:
                self.out['sharded_config'] = {'error': msg}
                emit(self.out)
            sys.stdout.flush(); sys.stderr.flush()
            # exit code: 0 keeps a launcher (torchrun tears the job down on the first non-zero exit) from discarding the line rank 0
            # has just printed; GPK_BENCH_STRICT_EXIT=1 makes every rank that abandons the secondary run leave with 3 instead
            os._exit(3 if os.environ.get('GPK_BENCH_STRICT_EXIT') == '1' else 0)

    def _poll(self):
        while not self.done.wait(0.5):
            if time.monotonic() > self.deadline:
                msg = f'rank {self.rank}: the sharded run did not finish within its deadline (GPK_SHARDED_TIMEOUT); abandoned'
                try:
                    if self.store is not None and not self.store.check([self.KEY]):
                        self.store.set(self.KEY, msg)
                except Exception:
                    pass
                self._finish(msg)
            try:
                if self.store is not None and self.store.check([self.KEY]):
                    self._finish(self.store.get(self.KEY).decode(errors='replace'))
            except Exception:                                     # store gone: the job is being torn down anyway
                return

    def fail(self, msg):
        """called by the rank that caught the exception; if a peer reported first, its message is the cause (this rank most
        likely only saw the peer's connection close)"""
        try:
            if self.store is not None:
                if self.store.check([self.KEY]):
                    msg = self.store.get(self.KEY).decode(errors='replace')
                else:
                    self.store.set(self.KEY, msg)
        except Exception:
            pass
        self._finish(msg)

    def stop(self):
        self.done.set()
        self.thread.join(timeout=5)


def syrk_pipelined_flops(N, nz, tile=64, bk=16, ob=512):
    """Flops LAUNCHED for Hb = S^T S in the pipelined form (csrc/gpk_factor.hip, gpk_i_syrk_potrf): one product per
    512-column block over the rows from the block's first row down; tiles entirely above the diagonal are skipped, so this
    equals the lower-tile count of the single-launch form."""
    nc = nz + 1
    total = 0.0
    for j0 in range(0, nc, ob):
        for n0 in range(j0, min(j0 + ob, nc), tile):
            bn = min(tile, nc - n0)
            k0 = (max(0, nz - (n0 + tile)) // bk) * bk
            total += 2.0 * bn * (nc - n0) * (N - k0)              # tile rows from the diagonal tile down
    return total


# ------------------------------------------------------------------------------------------------------ single GPU
def run_single(args, workload, comm=None, secondary=False, steps=None, warmup=None):
    """One independent solve on this rank's GPU.  With `comm` (N > 1 ranks): the timed region is bracketed by barriers and
    the slowest rank's time counts; value is the aggregate over the N replicas."""
    import torch
    import gpk
    if steps is not None or warmup is not None:
        args = argparse.Namespace(**vars(args))
        args.steps = args.steps if steps is None else steps
        args.warmup = args.warmup if warmup is None else warmup
    Nd, Nb, _, desc = WORKLOADS[workload]
    N, nz = 2 * Nd + Nb, Nd
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    local = int(os.environ.get('LOCAL_RANK', '0')) if world > 1 else 0
    torch.cuda.set_device(local)
    ctx = gpk.Context(local)
    Xd, Xb, f, g, z0 = synthetic_problem(Nd, Nb)

    # one-time phases (reported, not part of the metric): assembly (HBM-write bound) and Cholesky of Theta
    dXd, dXb = ctx.points(Xd), ctx.points(Xb)
    T = ctx.empty(N, N)
    kp = gpk.device.kernel_params('Gaussian', SIGMA)
    ratios = (C.c_double * 3)()
    nugget, info = 1e-13, -1
    asm_ms = chol_ms = None
    while True:
        ctx.prof_enable(True)                                     # (HIP events around the evaluator launch itself: gpk_prof_read_assembly)
        asm_kernel_ms = None
        for rep in range(3):                                      # warm + 2 timed
            ctx.timer_start()
            ctx._chk(ctx.lib.gpk_assemble(ctx.h, 0, 0, kp, dXd.ptr, Nd, dXb.ptr, Nb, nugget, 2, T.ptr, T.ld, ratios))
            ms = ctx.timer_stop()
            asm_ms = ms if rep == 1 else min(asm_ms or ms, ms)
            if rep >= 1:
                k_ms = ctx.prof_read_assembly()
                asm_kernel_ms = k_ms if asm_kernel_ms is None else min(asm_kernel_ms, k_ms)
        ctx.prof_enable(False)
        ctx.timer_start()
        info = ctx.potrf(T)
        chol_ms = ctx.timer_stop()
        if info == 0 or nugget >= 1e-8:
            break
        nugget *= 10.0                                            # SURVEY 7 hard part 1: report the nugget actually used
    # the first factorisation of a process also pays the first launch of every kernel it uses (code-object load): time it once
    # more on a freshly assembled matrix; both figures are reported, the roofline uses the second
    chol_first_ms = chol_ms
    ctx._chk(ctx.lib.gpk_assemble(ctx.h, 0, 0, kp, dXd.ptr, Nd, dXb.ptr, Nb, nugget, 2, T.ptr, T.ld, ratios))
    ctx.timer_start()
    info2 = ctx.potrf(T)
    chol_ms = ctx.timer_stop()
    assert info2 == info
    ctx.synchronize(); t0 = time.perf_counter()
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=ALPHA, p1=M_EXP)   # + gpk_trtri_diag of the factor (once per factor)
    ctx.synchronize(); dinv_ms = 1e3 * (time.perf_counter() - t0)
    dinv_block = prob.struct.dinv_block
    z = ctx.array(z0)
    prob.workspace()
    dev_first = first_step_on_device(ctx, prob, z0)               # for `parity` (also the first, code-object-loading step)
    # A timed step is what the product's GN_method executes per iteration (nonlinpdes-gpsolver_amd/src/PDEs.py, _gn_iterate; the
    # reference's src/PDEs.py:117-127 = Hessian + gradient + solve + update + one loss evaluation): gpk_gn_step, which since round 5
    # returns the loss of the iterate it starts from by true substitution (exact; one vector, its chain beside the end of the step),
    # so that one call IS one reference iteration.  GPK_SEPARATE_LOSS=1 -- in the product and here -- is round 4's sequence:
    # gpk_gn_step followed by a gpk_gn_loss call of its own.
    with_loss = os.environ.get('GPK_SEPARATE_LOSS', '0') == '1'
    losses = [ctx.gn_loss(prob, z)] if with_loss else []          # (the in-step value of the first step already IS J(z_0): no double entry)
    loss_s = 0.0

    def product_step(timed=False):
        nonlocal loss_s
        l_in = ctx.gn_step(prob, z)[0]
        if not with_loss:
            return l_in
        t1 = time.perf_counter()
        l_new = ctx.gn_loss(prob, z)                               # (both calls end with a host synchronisation: host clocks are exact)
        if timed:
            loss_s += time.perf_counter() - t1
        return l_new
    for _ in range(args.warmup):
        losses.append(product_step())
    ctx.prof_enable(True)
    ctx.synchronize(); torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(product_step(True))
    ctx.synchronize(); torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max_float(elapsed, torch.device('cuda', local))
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    if not with_loss:
        losses.append(ctx.gn_loss(prob, z))

    # The one-time phases once more, WARM (round 6): the measurements above are the first launches of the process (clocks still ramping,
    # pages and TLBs cold -- tools/assembly_store_ab.py: the same evaluator launch takes 0.112 ms there and 0.088 ms in steady state).  A
    # roofline fraction is about the kernel, so it is taken from launches issued AFTER the timed steps: 8 assemblies (mean launch duration; min and
    # median beside it) and 3 factorisations into a scratch matrix; the first-call figures stay in one_time_ms.
    asm_cold_ms, chol_cold_ms = asm_kernel_ms, chol_ms
    T2 = ctx.empty(N, N)
    ctx.prof_enable(True)
    asm_warm = []
    for _ in range(8):                                            # 8 launches, each alone on the chip (host round trip in between)
        ctx._chk(ctx.lib.gpk_assemble(ctx.h, 0, 0, kp, dXd.ptr, Nd, dXb.ptr, Nb, nugget, 2, T2.ptr, T2.ld, ratios))
        ctx.synchronize()
        asm_warm.append(ctx.prof_read_assembly())
    ctx.prof_enable(False)
    # ... and 8 calls back to back (point packing + evaluator each): the SUSTAINED write rate.  A launch that runs alone ends while up to
    # 256 MB of its 8 N^2 bytes still sit in the Infinity Cache; behind another write stream it has to wait for that one's write-back
    # (config 2: 0.088-0.100 ms alone, 0.113 ms in a row; at config 5, 9.2 GB, the two agree: 5.6 TB/s)
    ctx.synchronize(); ctx.timer_start()
    for _ in range(8):
        ctx._chk(ctx.lib.gpk_assemble(ctx.h, 0, 0, kp, dXd.ptr, Nd, dXb.ptr, Nb, nugget, 2, T2.ptr, T2.ld, ratios))
    asm_row_ms = ctx.timer_stop() / 8.0
    chol_warm = []
    for _ in range(3):
        ctx._chk(ctx.lib.gpk_assemble(ctx.h, 0, 0, kp, dXd.ptr, Nd, dXb.ptr, Nb, nugget, 2, T2.ptr, T2.ld, ratios))
        ctx.timer_start(); ctx.potrf(T2); chol_warm.append(ctx.timer_stop())
    T2.free()
    asm_kernel_ms, chol_ms = float(np.mean(asm_warm)), min(chol_warm)     # (assembly: the AVERAGE launch duration, as rocprofv3 --stats reports it)

    # accuracy half of the metric
    sol = z.download()
    pts_l2 = float(np.sqrt(np.sum((u_true(Xd[:, 0], Xd[:, 1]) - sol) ** 2) / Nd))
    Xt = test_grid(60)
    coeff = ctx.array(np.concatenate([ALPHA * sol ** M_EXP - f, sol, g]))
    ctx.potrs(T, coeff, nrhs=1)
    ext = ctx.extend('Nonlinear_elliptic', 'Gaussian', SIGMA, Xt, Xd, Xb, coeff).download()
    test_l2 = float(np.sqrt(np.sum((u_true(Xt[:, 0], Xt[:, 1]) - ext) ** 2) / Xt.shape[0]))

    steps = max(prof['steps'], 1)
    # the SYRK launch(es) Hb = S^T S, timed by HIP events on the stream they run on, inside the timed steps.  Pipelined mode
    # (default): the product is issued as one launch per 512-column block on the GEMM partition (256 - chain_cus CUs) while the
    # panel chains of the factorisation run on the other CUs; syrk_ms is the SUM of those launches per step.
    syrk_ms = prof['syrk_launch_ms'] / steps
    phase_ms = prof['syrk_ms'] / steps                            # product + factorisation of Hb (one phase since round 2)
    pipelined = prof['pipelined']
    gemm_cus = 256 - prof['chain_cus'] if pipelined else 256
    syrk_flops = syrk_executed_flops(N, nz)                      # lower tiles, leading zeros skipped (the useful work)
    syrk_launched = syrk_pipelined_flops(N, nz) if pipelined else syrk_flops
    syrk_dense = float(N) * (nz + 1) ** 2                        # dense symmetric count, SURVEY 8d ("SYRK N n_z^2")
    achieved = syrk_flops / (syrk_ms * 1e-3) / 1e12
    traffic, traffic_source = stored_pmc_traffic('syrk', workload)
    trsm_traffic, trsm_traffic_source = stored_pmc_traffic('trsm_gemm', workload)
    # dominant kernel of the step: gemm_f64_kernel<NN> = the whole solve phase S = L^{-1}[A | F] (GEMMs only since round 2)
    trsm_ms = prof['trsm_ms'] / steps
    uses_dinv = prob.Dinv is not None and os.environ.get('GPK_DEBUG_SET', '').find('10=0') < 0
    trsm_flops, trsm_launches = trsm_dinv_executed_flops(N, nz, gpk.device.dinv_block_for(N)) if uses_dinv else (None, None)
    trsm_achieved = trsm_flops / (trsm_ms * 1e-3) / 1e12 if trsm_flops else None
    # the same quantities counted by the launch logic itself while the steps ran (gpk_prof_read_flops): must agree with the models
    counted = {'solve_flops_per_step': prof['solve_flops'] / steps, 'solve_launches_per_step': prof['solve_launches'] / steps,
               'product_flops_per_step': prof['product_flops'] / steps, 'product_launches_per_step': prof['product_launches'] / steps,
               'cholesky_H_update_flops_per_step': prof['potrf_update_flops'] / steps,
               'note': 'flops executed by the matrix-product launches of the timed steps, accumulated inside libgpk from the K range of every tile '
                       '(gpk_prof_read_flops); the host-side models used for the rooflines (flops_per_step / flops_per_launch) must match: '
                       'solve model %.6e, product model (launched tiles) %.6e' % (trsm_flops or float('nan'), syrk_launched)}
    out = {
        'metric': 'Gauss-Newton steps/sec + L2 error, NonLinElliptic2d at N_domain points',
        'value': world * args.steps / elapsed, 'unit': 'GN steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': desc, 'N_domain': Nd, 'N_boundary': Nb, 'theta_order': N, 'unknowns': nz,
                   'parallelism': ('one GPU' if world == 1 else
                                   f'{world} independent replicas of the workload, one per GPU (no data-path collective); the '
                                   f'sharded north-star configuration is reported under sharded_config'),
                   'kernel': 'Gaussian', 'kernel_parameter': SIGMA, 'nugget': nugget, 'nugget_type': 'adaptive',
                   'timed_sequence': TIMED_SEQUENCE[with_loss],
                   'formulation': 'TRSM(n_z+1 rhs) + SYRK + POTRF(H) + TRSV every step.  Nothing that depends on the iterate is cached '
                                  'across steps; what IS computed once per factor and reused by every step (like the factor L itself) are the '
                                  f'inverses of the {dinv_block}-row diagonal blocks of L (gpk_trtri_diag, block size gpk.device.dinv_block_for(N); '
                                  'its cost is one_time_ms.diagonal_block_inverses) -- the solve then runs as GEMMs only -- and the zeroed region '
                                  'of the solve workspace left of the leading-zero boundary.  The structural zeros of A(z) (column j zero above '
                                  'row j) are skipped inside TRSM and SYRK; f1_tflops is the DENSE F1 flop count / time (an equivalent rate, '
                                  'not executed flops)', 'seed': 0},
        'l2_error': {'pts_L2_err': pts_l2, 'test_L2_err': test_l2, 'gn_steps_run': args.warmup + args.steps,
                     'loss_first': losses[0], 'loss_last': losses[-1], 'chol_info': info},
        'f1_tflops': world * f1_flops(N, nz) * args.steps / elapsed / 1e12,
        'phases_ms_per_step': {'trsm': prof['trsm_ms'] / steps, 'syrk_and_potrf_H': phase_ms,
                               'syrk_launches_sum': syrk_ms, 'trsv_update': prof['trsv_update_ms'] / steps,
                               'loss_call': 1e3 * loss_s / args.steps if with_loss else 0.0,
                               'pipelined': bool(pipelined), 'chain_partition_cus': prof['chain_cus'] if pipelined else 0},
        'one_time_ms': {'assembly': asm_ms, 'cholesky_theta': chol_cold_ms, 'cholesky_theta_warm': chol_ms, 'cholesky_theta_first_call': chol_first_ms, 'diagonal_block_inverses': dinv_ms,
                        'diagonal_block_rows': dinv_block},
        'roofline': ({'bound': 'mfma',
                      'kernel': 'gemm_f64_kernel<.., NN> = the solve phase S = L^{-1}[A | F]: update products of the recursion and '
                                'triangular products with the inverted diagonal blocks of the factor, all fp64 MFMA',
                      'achieved': trsm_achieved, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': trsm_achieved / FP64_MFMA_PEAK_TFLOPS,
                      'traffic': trsm_traffic, 'traffic_source': trsm_traffic_source,
                      'flops_per_step': trsm_flops, 'launches_per_step': trsm_launches, 'phase_ms_per_step': trsm_ms,
                      'avg_launch_ms': trsm_ms / trsm_launches,
                      'dense_flops_per_step': float(N) * N * (nz + 1), 'dense_equivalent_tflops': float(N) * N * (nz + 1) / (trsm_ms * 1e-3) / 1e12,
                      'peak_source': 'datasheet fp64 matrix rate (2.4 GHz); v_mfma_f64_16x16x4_f64 issue-rate ubench on this chip 71-74 '
                                     '(the chip sustains ~2.1-2.2 GHz under fp64 MFMA load)',
                      'note': 'achieved = flops EXECUTED by the phase (structural zeros and the zero halves of the triangular blocks '
                              'skipped; 60.8 % of the dense count N^2 (n_z+1)) / phase time from HIP events on the launch stream inside '
                              'the timed steps (the events also bracket a memset and the O(N) build kernel, ~40 us)'}
                     if trsm_flops else None),
        'roofline_syrk': {'bound': 'mfma', 'kernel': 'gemm_f64_kernel<TN, lower tiles> = SYRK Hb = S^T S',
                     'achieved': achieved, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / FP64_MFMA_PEAK_TFLOPS,
                     'traffic': traffic, 'traffic_source': traffic_source, 'flops_per_launch': syrk_flops, 'dense_flops_per_launch': syrk_dense,
                     'dense_equivalent_tflops': syrk_dense / (syrk_ms * 1e-3) / 1e12, 'avg_launch_ms': syrk_ms,
                     'launches_per_step': (nz + 1 + 511) // 512 if pipelined else 1,
                     'launch_shape': ('block 0 (512 columns) as one launch on the whole chip, then one split-K launch per 512-column block on '
                                      'the GEMM partition (K cut so that a launch has ~1000 workgroups)' if pipelined else 'one launch, lower tiles'),
                     'cus_available_to_kernel': gemm_cus,
                     'frac_of_partition_peak': achieved / (FP64_MFMA_PEAK_TFLOPS * gemm_cus / 256.0),
                     'launched_flops_per_step': syrk_launched,
                     'note': 'flops_per_launch = useful flops of the product per step (lower tiles, structural zeros skipped); avg_launch_ms = '
                             'sum of the product launches per step; peak is the FULL chip although the launches only get '
                             'cus_available_to_kernel CUs when pipelined (the rest runs the Cholesky panel chain concurrently)',
                     'peak_source': 'datasheet fp64 matrix rate; v_mfma_f64_16x16x4_f64 issue-rate ubench on this chip ~74'},
        'roofline_assembly': {'bound': 'hbm', 'kernel': 'assemble2_kernel<elliptic> (the Gram evaluator launch itself)',
                              'achieved': 8.0 * N * N / (asm_kernel_ms * 1e-3) / 1e9,
                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': 8.0 * N * N / (asm_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              'bytes_per_launch': 8.0 * N * N, 'kernel_ms': asm_kernel_ms, 'kernel_ms_min': min(asm_warm), 'kernel_ms_median': float(np.median(asm_warm)),
                              'sustained_call_ms': asm_row_ms, 'sustained_gbs': 8.0 * N * N / (asm_row_ms * 1e-3) / 1e9,
                              'kernel_ms_first_launches': asm_cold_ms, 'launches_timed': len(asm_warm),
                              'store_policy': 'plain global_store_dwordx4 (A/B of nt / sc0 sc1 / sc0 sc1 nt: profiles/r06_assembly_store_ab.json -- nt is 15-20 % slower)',
                              'call_ms': asm_ms, 'call_gbs': 8.0 * N * N / (asm_ms * 1e-3) / 1e9,
                              'note': 'achieved = 8 N^2 bytes written / duration of the evaluator launch (HIP events around that launch on its '
                                      'stream), MEAN over launches_timed launches issued one at a time after the timed steps (kernel_ms_min / _median beside it; sustained_gbs = 8 calls back to back incl. point packing; '
                                      'kernel_ms_first_launches = the same at process start, clocks and pages cold); call_ms = the whole gpk_assemble call (point packing kernel + launch overheads) by events around the call'},
        # the factorisation of Theta (north star: "MFMA fp64 utilisation for the factorisation"): N^3/3 flops are what a Cholesky
        # executes (nothing structural to skip) over the whole gpk_potrf call -- panel kernels, rank-64 updates and the trailing
        # GEMM updates together, HIP events on the handle's stream
        'roofline_cholesky_theta': {'bound': 'mfma', 'kernel': 'gpk_potrf(Theta): potrf_panel_mfma_kernel chain + gemm_k64_kernel + gemm_f64_kernel<NT> trailing updates',
                                    'achieved': N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 'flops': N ** 3 / 3.0, 'ms': chol_ms,
                                    'ms_median': float(np.median(chol_warm)), 'ms_before_the_timed_steps': chol_cold_ms, 'calls_timed': len(chol_warm)},
        'flops_counted_by_library': counted,
        'step_executed': {'flops_per_step': (trsm_flops or 0.0) + syrk_flops + (nz + 1) ** 3 / 3.0,
                          'tflops': ((trsm_flops or 0.0) + syrk_flops + (nz + 1) ** 3 / 3.0) * args.steps / elapsed / 1e12,
                          'frac_of_peak': ((trsm_flops or 0.0) + syrk_flops + (nz + 1) ** 3 / 3.0) * args.steps / elapsed / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                          'note': 'flops EXECUTED by one step (solve with structural zeros skipped + lower-tile product + Cholesky of H) / wall time per step'},
    }
    if out['roofline'] is None:                                   # substitution schedule (debug): the SYRK is the measured GEMM then
        out['roofline'] = out['roofline_syrk']
    if not args.no_structured and world == 1 and not secondary:
        out['structured_step'] = structured_step(args, ctx, gpk, prob, T, Nd, Nb, f, g, z0, sol, Xd)
    if not args.no_cpu_baseline and world == 1:
        from oracle import gp_oracle as O
        L = tril_inplace(T.download())
        attach_cpu_baseline(out, O.EllipticSystem(ALPHA, M_EXP, f, g), [L], z0, f'N={N}, n_z={Nd}', dev_first, elapsed / args.steps)
        del L
    ctx.close()
    return out if rank == 0 else None


def structured_step(args, ctx, gpk, prob, T, Nd, Nb, f, g, z0, sol_default, Xd):
    """SECONDARY figures, never `value`: the same Gauss-Newton iteration with the OPTIONAL structured modes.
    solve_level (gpk_gn_structured_prepare): W = L^{-1}[unit columns] is computed once, then every step forms
    [L^{-1}A(z) | L^{-1}F(z)] from it in one memory-bound pass instead of the triangular solve; product, factorisation, update unchanged.
    gram_level (gpk_gn_gram_prepare, on top): the Gram blocks of W once, then the bordered matrix is assembled in O(n_z^2) per step --
    neither the solve nor the product S^T S is executed, only the Cholesky factorisation of H, the solve with it and the update.
    Same start and number of steps as the main measurement; setup_ms is the one-time cost of the mode (incl. the inverted blocks)."""
    prob.release_workspace()
    out = {'note': 'optional modes (GNProblem(structured=True / 2)); NOT the reference operation sequence per step: z-independent '
                   'operators are computed once per factor (setup_ms) and the step uses the linearity of the solve in its right-hand '
                   'sides; reported next to, never instead of, `value`'}
    for key, mode in (('solve_level', True), ('gram_level', 2)):
        ctx.synchronize(); t0 = time.perf_counter()
        sp = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=ALPHA, p1=M_EXP, structured=mode)
        ctx.synchronize(); setup = time.perf_counter() - t0
        z = ctx.array(z0)
        for _ in range(args.warmup):
            ctx.gn_step(sp, z)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.gn_step(sp, z)
        ctx.synchronize(); elapsed = time.perf_counter() - t0
        sol = z.download()
        out[key] = {'value': args.steps / elapsed, 'unit': 'GN steps/s', 'ms_per_step': 1e3 * elapsed / args.steps, 'setup_ms': 1e3 * setup,
                    'pts_L2_err': float(np.sqrt(np.sum((u_true(Xd[:, 0], Xd[:, 1]) - sol) ** 2) / Nd)),
                    'iterate_rel_diff_vs_default': float(np.linalg.norm(sol - sol_default) / np.linalg.norm(sol_default))}
        z.free(); sp.release_workspace()
        for a in (sp.W1, sp.W2, sp.v0, sp.G, sp.pvec, sp.Dinv):
            if a is not None:
                a.free()
    return out


def host_mem_available_gb():
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable:'):
                return int(line.split()[1]) / 1048576.0
    except Exception:                                             # noqa: BLE001
        pass
    return 0.0


def tril_inplace(a):
    """zero the strict upper triangle of a square host array without a second copy (9.2 GB at BASELINE config 5)"""
    n = a.shape[0]
    for i0 in range(0, n, 2048):
        i1 = min(i0 + 2048, n)
        a[i0:i1, i1:] = 0.0
        a[i0:i1, i0:i1] = np.tril(a[i0:i1, i0:i1])
    return a


def cpu_baseline(sysm, Ls, z0, sample, dev=None, with_b1=True, with_mkl=True):
    """The CPU oracle on this box's host cores, full workload size, ONE Gauss-Newton step from the benchmark's start z0 on the
    DEVICE's factor(s) Ls (downloaded):
    B1 = the reference's operation sequence (general LU solves of the triangular L for Hessian, gradient and loss, LU
    solve of H: src/PDEs.py:86,97,118,295-307, src/InverseProblems.py:126-166) -- the stand-in for 'reference JAX on CPU', which
    cannot be installed here;
    B2 = the triangular formulation the GPU path uses (TRSM + SYRK + Cholesky of H), timed twice: on numpy/scipy (OpenBLAS)
    and on torch CPU (MKL, SURVEY 8d "prefer MKL via torch"); the faster one is reported as the triangular figure.
    dev = (z1, loss0, loss1) of the device for the same step: the iterates the oracle computes ANYWAY are compared with it and
    reported as `parity` (round 4; they used to be thrown away) -- relative deviation of the first iterate from B2 and from B1,
    of the loss at the start and after the step; `ok` = iterate deviations <= 1e-6 (the north star's bound)."""
    from oracle import gp_oracle as O
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))
    t_b1 = z1_b1 = None
    if with_b1:
        t0 = time.perf_counter()
        H, grad = O.gn_quantities(sysm, Ls, z0, faithful=True)
        z1_b1 = z0 - np.linalg.solve(H, grad)
        O.loss(sysm, Ls, z1_b1, faithful=True)
        t_b1 = time.perf_counter() - t0
        del H
    t0 = time.perf_counter()
    z1_b2, hist_b2 = O.gn_method(sysm, Ls, z0, 1, 1, faithful=False)
    t_b2_np = time.perf_counter() - t0
    t_b2_mkl, mkl_threads, mkl_err = None, None, None
    if with_mkl:
        try:
            import torch
            mkl_threads = torch.get_num_threads()
            Lt = [None if L is None else torch.from_numpy(L) for L in Ls]
            nz = sysm.nz
            def mkl_step():
                Hb = None
                for L, A, F in zip(Lt, sysm.A(z0), sysm.F(z0)):
                    AF = torch.from_numpy(np.ascontiguousarray(np.concatenate([A, F[:, None]], axis=1)))
                    S = AF if L is None else torch.linalg.solve_triangular(L, AF, upper=False)
                    Hb = S.T @ S if Hb is None else Hb + S.T @ S
                _, ge, he = sysm.extra(z0)
                if ge is not None:                                # data misfit (Darcy): diagonal Hessian term and gradient, halved like Hb
                    Hb[torch.arange(nz), torch.arange(nz)] += torch.from_numpy(0.5 * he)
                    Hb[:nz, nz] += torch.from_numpy(0.5 * ge)
                Lh = torch.linalg.cholesky(Hb[:nz, :nz])
                d = torch.cholesky_solve(Hb[:nz, nz:nz + 1], Lh)
                z1 = z0 - d[:, 0].numpy()
                tot = sysm.extra(z1)[0]
                for L, F in zip(Lt, sysm.F(z1)):
                    w = torch.from_numpy(np.ascontiguousarray(F))[:, None]
                    if L is not None:
                        w = torch.linalg.solve_triangular(L, w, upper=False)
                    tot += float((w * w).sum())
                return tot
            t0 = time.perf_counter()
            mkl_step()
            t_b2_mkl = time.perf_counter() - t0
        except Exception as e:                                    # noqa: BLE001 -- reported
            mkl_err = f'{type(e).__name__}: {e}'
    t_b2 = min(t for t in (t_b2_np, t_b2_mkl) if t is not None)
    try:
        import threadpoolctl
        threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    parity = None
    if dev is not None:
        z1_dev, loss0_dev, loss1_dev = dev
        parity = {'z1_rel_dev_vs_B2': rel(z1_dev, z1_b2), 'z1_rel_dev_vs_B1': rel(z1_dev, z1_b1) if z1_b1 is not None else None,
                  'loss0_rel_dev': abs(loss0_dev / hist_b2[0] - 1.0), 'loss1_rel_dev': abs(loss1_dev / hist_b2[1] - 1.0),
                  'loss0_device': loss0_dev, 'loss0_oracle': hist_b2[0], 'loss1_device': loss1_dev, 'loss1_oracle': hist_b2[1],
                  'tol': PARITY_TOL,
                  'what': 'first Gauss-Newton iterate z1 from the seeded N(0,1) start z0, device (gpk_gn_step) vs the CPU oracle on the '
                          "device's factor: B2 = triangular formulation, B1 = reference operation sequence (LU solves); relative 2-norm "
                          'deviations; loss0 / loss1 = loss at z0 / z1 (device: in-step value / gpk_gn_loss)'}
        devs = [parity['z1_rel_dev_vs_B2']] + ([parity['z1_rel_dev_vs_B1']] if z1_b1 is not None else [])
        parity['ok'] = bool(all(np.isfinite(d) and d <= PARITY_TOL for d in devs))
    out = {'value': 1.0 / (t_b1 if t_b1 is not None else t_b2), 'unit': 'GN steps/s', 'cores': threads, 'kind': 'port',
           'sample': f'1 Gauss-Newton step at the full workload size ({sample}), '
                     + ('reference operation sequence (general LU solves of L for Hessian, gradient and loss + LU solve of H)' if with_b1 else
                        'triangular formulation only (the reference operation sequence would take minutes at this size)')
                     + ' with numpy/scipy BLAS; factor L taken from the device',
           'seconds_per_step': t_b1 if t_b1 is not None else t_b2, 'reference_sequence_timed': bool(with_b1),
           'triangular_formulation_value': 1.0 / t_b2, 'triangular_seconds_per_step': t_b2,
           'triangular_seconds_per_step_openblas': t_b2_np, 'triangular_seconds_per_step_torch_mkl': t_b2_mkl,
           'torch_threads': mkl_threads, 'torch_mkl_error': mkl_err, 'host_cpus': os.cpu_count()}
    return out, parity


def attach_cpu_baseline(out, sysm, Ls, z0, sample, dev, sec_per_step, **kw):
    cb, parity = cpu_baseline(sysm, Ls, z0, sample, dev, **kw)
    cb['gpu_speedup_vs_reference_sequence'] = cb['seconds_per_step'] / sec_per_step if cb['reference_sequence_timed'] else None
    cb['gpu_speedup_vs_triangular_best_cpu'] = cb['triangular_seconds_per_step'] / sec_per_step
    out['cpu_baseline'] = cb
    out['parity'] = parity


def first_step_on_device(ctx, prob, z0):
    """(z1, loss at z0 as the step reports it, loss at z1 by gpk_gn_loss) of ONE Gauss-Newton step from z0 -- the device half of `parity`"""
    zp = ctx.array(z0)
    loss0, info = ctx.gn_step(prob, zp)
    z1 = zp.download()
    loss1 = ctx.gn_loss(prob, zp)
    zp.free()
    return z1, loss0, loss1


# ------------------------------------------------------------------------------------------------------ configs 3 and 4
def system_problem(workload):
    """The reference drivers' own set-up of BASELINE configs 3 and 4 (main_Burgers1d.py:27-60, main_DarcyFlow2d.py:56-100 of the
    reference): seed, sampler draw order, N(0,1) initial guess; for Darcy the observations come from the finite-difference truth."""
    from src.sample_points import sampled_pts_rdm
    Nd0, Nb0, gn_steps, desc = WORKLOADS[workload]
    if workload == 'c3':
        import main_Burgers1d as drv
        np.random.seed(0)
        Xd, Xb = sampled_pts_rdm(Nd0, Nb0, np.array(drv.SPACE_TIME), time_dependent=True)
        z0 = np.random.normal(0.0, 1.0, 3 * Xd.shape[0])
        truth = drv.cole_hopf_truth(0.02)
        Xt = _grid(60, *drv.SPACE_TIME)
        return dict(system='Burgers', layouts=['Burgers'], kernel='anisotropic_Gaussian', kp=[0.3, 0.05], nugget=1e-5, Xd=Xd, Xb=Xb,
                    z0=z0, f=np.zeros(Xd.shape[0]), g=drv.initial_and_lateral(Xb[:, 0], Xb[:, 1]), p0=1.0, p1=0.02, data=None,
                    Xt=Xt, truth_pts=truth(Xd[:, 0], Xd[:, 1]), truth_test=truth(Xt[:, 0], Xt[:, 1]), desc=desc, gn_steps=gn_steps)
    import main_DarcyFlow2d as drv
    from scipy.interpolate import griddata
    from reference_solver.FD_for_Darcy_flow import FD_Darcy_flow_2d
    np.random.seed(9999)
    Xd, Xb = sampled_pts_rdm(Nd0, Nb0, np.array(drv.UNIT_SQUARE))
    ndata, noise = 60, 1e-3
    u_grid = FD_Darcy_flow_2d(drv.GRID - 2, drv.permeability, drv.source)
    xx = np.linspace(0, 1, drv.GRID)
    XX, YY = np.meshgrid(xx, xx)
    obs = griddata((XX.flatten(), YY.flatten()), u_grid.reshape(-1, 1), (Xd[:ndata, 0], Xd[:ndata, 1]), method='linear')[:, 0]
    data = obs + noise * np.random.normal(0, 1.0, ndata)          # Darcy_flow2d.get_observation, src/InverseProblems.py:60-64 of the reference
    z0 = np.random.normal(0.0, 1.0, 6 * Nd0)
    Xt = np.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)
    return dict(system='Darcy_flow2d', layouts=['Darcy_u', 'Darcy_a'], kernel='Gaussian', kp=SIGMA, nugget=1e-8, Xd=Xd, Xb=Xb, z0=z0,
                f=np.ones(Nd0), g=np.zeros(Xb.shape[0]), p0=noise, p1=0.0, data=data, Xt=Xt, truth_pts=None,
                truth_test=u_grid.reshape(-1), truth_a=drv.permeability(Xt[:, 0], Xt[:, 1]), desc=desc, gn_steps=gn_steps)


def _grid(n, r1, r2):
    XX, YY = np.meshgrid(np.linspace(r1[0], r1[1], n), np.linspace(r2[0], r2[1], n))
    return np.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)


def run_system(args, workload, steps=None, warmup=None):
    """BASELINE config 3 (Burgers) or 4 (Darcy) on one GPU under the same clock as the primary workload: K timed Gauss-Newton steps
    after W warm-up steps from the seeded start, per-phase times and executed flops from the library's own counters, L2 errors
    against the Cole-Hopf / finite-difference truths, CPU baselines B1 / B2 and the `parity` object.  Past convergence (8 steps) a
    step still does exactly the same work: nothing in it depends on the size of the update."""
    import torch
    import gpk
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    torch.cuda.set_device(0)
    ctx = gpk.Context(0)
    P = system_problem(workload)
    Xd, Xb, z0 = P['Xd'], P['Xb'], P['z0']
    Nd, Nb = Xd.shape[0], Xb.shape[0]
    factors, one_time, asm_bytes, asm_kernel_s = [], {}, 0.0, 0.0
    for lay in P['layouts']:
        ctx.prof_enable(True)
        T = None
        for rep in range(2):
            ctx.timer_start()
            T, _ = ctx.assemble(lay, P['kernel'], P['kp'], Xd, Xb, P['nugget'], 'adaptive', out=T)
            call_ms = ctx.timer_stop()                            # (includes the upload of the points: host arrays here)
            k_ms = ctx.prof_read_assembly()
        ctx.prof_enable(False)
        n = T.rows
        ctx.timer_start(); info = ctx.potrf(T); first_ms = ctx.timer_stop()
        ctx.assemble(lay, P['kernel'], P['kp'], Xd, Xb, P['nugget'], 'adaptive', out=T)
        ctx.timer_start(); info = ctx.potrf(T); chol_ms = ctx.timer_stop()
        if info != 0:
            raise RuntimeError(f'{workload}: Cholesky of the {lay} Gram matrix failed (info {info}) at the reference nugget')
        one_time[lay] = {'order': n, 'assembly_kernel_ms': k_ms, 'assembly_call_ms': call_ms, 'assembly_gbs': 8.0 * n * n / (k_ms * 1e-3) / 1e9,
                         'cholesky_ms': chol_ms, 'cholesky_first_call_ms': first_ms, 'cholesky_tflops': n ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12}
        asm_bytes += 8.0 * n * n; asm_kernel_s += k_ms * 1e-3
        factors.append(T)
    ctx.synchronize(); t0 = time.perf_counter()
    prob = gpk.GNProblem(ctx, P['system'], Nd, Nb, P['f'], P['g'], factors[0], p0=P['p0'], p1=P['p1'], data_u=P['data'],
                         L2=factors[1] if len(factors) > 1 else None)
    ctx.synchronize(); dinv_ms = 1e3 * (time.perf_counter() - t0)
    nz, rows = prob.nz, prob.rows
    prob.workspace()
    dev_first = first_step_on_device(ctx, prob, z0)
    z = ctx.array(z0)
    with_loss = os.environ.get('GPK_SEPARATE_LOSS', '0') == '1'       # (as in run_single: the product's per-iteration sequence)
    losses = [ctx.gn_loss(prob, z)] if with_loss else []          # (the in-step value of the first step already IS J(z_0): no double entry)
    loss_s = 0.0

    def product_step(timed=False):
        nonlocal loss_s
        l_in = ctx.gn_step(prob, z)[0]
        if not with_loss:
            return l_in
        t1 = time.perf_counter()
        l_new = ctx.gn_loss(prob, z)
        if timed:
            loss_s += time.perf_counter() - t1
        return l_new
    for _ in range(warmup):
        losses.append(product_step())
    ctx.prof_enable(True)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(product_step(True))
    ctx.synchronize(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    if not with_loss:
        losses.append(ctx.gn_loss(prob, z))
    n = max(prof['steps'], 1)
    sol = z.download()
    # ---- accuracy against the independent truths (the drivers' test grids)
    meas = ctx.gn_measurement(prob, z)                            # [sol_vec_a (Darcy) | sol_vec]: the right-hand side of the extension
    err = {}
    if workload == 'c3':
        coeff = ctx.array(meas[:4 * Nd + Nb]); ctx.potrs(factors[0], coeff, nrhs=1)
        ext = ctx.extend('Burgers', P['kernel'], P['kp'], P['Xt'], Xd, Xb, coeff).download()
        err = {'pts_L2_err': float(np.sqrt(np.mean((sol[:Nd] - P['truth_pts']) ** 2))),
               'test_L2_err': float(np.sqrt(np.mean((ext - P['truth_test']) ** 2))), 'truth': 'Cole-Hopf transform, 80-node Gauss-Hermite quadrature'}
    else:
        cu = ctx.array(meas[3 * Nd:7 * Nd + Nb]); ctx.potrs(factors[0], cu, nrhs=1)
        ca = ctx.array(meas[:3 * Nd]); ctx.potrs(factors[1], ca, nrhs=1)
        eu = ctx.extend('Darcy_u', P['kernel'], P['kp'], P['Xt'], Xd, Xb, cu).download()
        ea = ctx.extend('Darcy_a', P['kernel'], P['kp'], P['Xt'], Xd, Xb, ca).download()
        err = {'u_test_L2_err': float(np.sqrt(np.mean((eu - P['truth_test']) ** 2))),
               'a_test_L2_err': float(np.sqrt(np.mean((np.exp(ea) - P['truth_a']) ** 2))),
               'data_misfit_rms': float(np.sqrt(np.mean((sol[3 * Nd:3 * Nd + 60] - P['data']) ** 2))),
               'truth': 'flux-form finite differences on the 80 x 80 grid of the driver'}
    solve_ms, phase_ms, prod_ms, tail_ms = prof['trsm_ms'] / n, prof['syrk_ms'] / n, prof['syrk_launch_ms'] / n, prof['trsv_update_ms'] / n
    solve_fl, prod_fl, upd_fl = prof['solve_flops'] / n, prof['product_flops'] / n, prof['potrf_update_flops'] / n
    chol_fl = (nz + 1) ** 3 / 3.0
    dense = sum(float(T.rows) ** 2 * (nz + 1) for T in factors) + float(rows) * (nz + 1) ** 2 + chol_fl   # F1 of SURVEY 8d for this system
    executed = solve_fl + prod_fl + chol_fl
    tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12
    out = {'value': steps / elapsed, 'unit': 'GN steps/s', 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * elapsed / steps,
           'config': {'workload': P['desc'], 'N_domain': Nd, 'N_boundary': Nb, 'theta_orders': [T.rows for T in factors], 'unknowns': nz,
                      'stacked_rows': rows, 'kernel': P['kernel'], 'kernel_parameter': P['kp'], 'nugget': P['nugget'], 'nugget_type': 'adaptive',
                      'reference_gn_steps': P['gn_steps'], 'timed_sequence': TIMED_SEQUENCE[with_loss],
                      **({'iteration_independent': 'a-part rows [w1; w2; w0] (reference src/InverseProblems.py:137-146 do not involve z_old): W_a = L_a^{-1} A_a and '
                                                   'W_a^T W_a computed ONCE per factor with the step\'s own launches (one_time_ms.darcy_a_part_prepare), bit-identical '
                                                   'iterates; GPK_DARCY_CACHE=0 recomputes them every step.  f1_tflops stays the dense F1 count over the measured time '
                                                   '(a dense-equivalent rate, not a utilisation); roofline / step_executed count what is executed'}
                         if prob.Wa is not None else {}),
                      'schedule': ('leading-zero layout (unknowns interleaved by collocation point: staircase of slope 1/3)' if workload == 'c3' else
                                   'Darcy: a-part, u-part and data rows stacked; leading-zero layout with the unknowns ordered v1, v2, w1, w2, w0, v0 -- a '
                                   'piecewise staircase for the u-part, a slope-1 staircase on a column sub-range for the a-part (DESIGN section 4)')},
           'l2_error': dict(err, gn_steps_run=warmup + steps, loss_first=losses[0], loss_last=losses[-1]),
           'f1_tflops': dense * steps / elapsed / 1e12,
           'phases_ms_per_step': {'trsm': solve_ms, 'syrk_and_potrf_H': phase_ms, 'syrk_launches_sum': prod_ms, 'trsv_update': tail_ms,
                                  'loss_call': 1e3 * loss_s / steps if with_loss else 0.0, 'pipelined': bool(prof['pipelined'])},
           'one_time_ms': dict(one_time, diagonal_block_inverses=dinv_ms - (prob.darcy_prepare_ms or 0.0), diagonal_block_rows=prob.struct.dinv_block,
                               **({'darcy_a_part_prepare': prob.darcy_prepare_ms} if prob.darcy_prepare_ms is not None else {})),
           'roofline': {'bound': 'mfma', 'kernel': 'gemm_f64_kernel<.., NN> = the solve phase S = L^{-1}[A | F] of every factor',
                        'achieved': tf(solve_fl, solve_ms), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf(solve_fl, solve_ms) / FP64_MFMA_PEAK_TFLOPS,
                        'traffic': stored_pmc_traffic('trsm_gemm', workload)[0], 'traffic_source': stored_pmc_traffic('trsm_gemm', workload)[1],
                        'flops_per_step': solve_fl, 'launches_per_step': prof['solve_launches'] / n, 'phase_ms_per_step': solve_ms,
                        'avg_launch_ms': solve_ms / max(prof['solve_launches'] / n, 1),
                        'note': 'flops EXECUTED, counted by the launch logic (gpk_prof_read_flops) / phase time from HIP events inside the timed steps'},
           'roofline_syrk': {'bound': 'mfma', 'kernel': 'gemm_f64_kernel<TN> = the product Hb = S^T S', 'achieved': tf(prod_fl, prod_ms),
                             'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf(prod_fl, prod_ms) / FP64_MFMA_PEAK_TFLOPS,
                             'traffic': stored_pmc_traffic('syrk', workload)[0], 'flops_per_step': prod_fl,
                             'launches_per_step': prof['product_launches'] / n, 'avg_launch_ms': prod_ms,
                             'note': 'LAUNCHED flops of the product (lower tiles; when pipelined also the upper halves of the diagonal 512-blocks) / sum of its launches'},
           'roofline_assembly': {'bound': 'hbm', 'kernel': 'assemble kernels of this system', 'achieved': asm_bytes / asm_kernel_s / 1e9,
                                 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': asm_bytes / asm_kernel_s / 1e9 / HBM_PEAK_GBS, 'bytes': asm_bytes},
           'step_executed': {'flops_per_step': executed, 'tflops': executed * steps / elapsed / 1e12,
                             'frac_of_peak': executed * steps / elapsed / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             'cholesky_H_update_flops_counted': upd_fl,
                             'note': 'solve + product (counted) + (n_z+1)^3/3 for the Cholesky of H, over wall time per step'}}
    if not args.no_structured:
        # OPTIONAL structured solve of this system (round 6, gpk_gn_structured_prepare; GPK_STRUCTURED=1 in the drivers): A(z) = A1 diag(d(z)) + A2,
        # W1 = L^{-1}A1 and W2 = L^{-1}A2 once per factor, the step forms L^{-1}A(z) in one memory-bound pass and solves only the F column.
        # Not the reference's per-step sequence (it changes the rounding of the solve): reported NEXT TO the workload's value, never as it --
        # with its one-time cost and the flops it executes
        try:
            prob.release_workspace()
            sp = gpk.GNProblem(ctx, P['system'], Nd, Nb, P['f'], P['g'], factors[0], p0=P['p0'], p1=P['p1'], data_u=P['data'],
                               L2=factors[1] if len(factors) > 1 else None, structured=True)
            zs = ctx.array(z0)
            for _ in range(warmup):
                ctx.gn_step(sp, zs)
            ctx.prof_enable(True)
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                ctx.gn_step(sp, zs)
            ctx.synchronize(); el_s = time.perf_counter() - t0
            pr = ctx.prof_read(); ctx.prof_enable(False)
            m = max(pr['steps'], 1)
            sol_s = zs.download()
            ex_s = pr['solve_flops'] / m + pr['product_flops'] / m + chol_fl
            out['structured_step'] = {'solve_level': {
                'value': steps / el_s, 'unit': 'GN steps/s', 'ms_per_step': 1e3 * el_s / steps, 'setup_ms': sp.structured_prepare_ms,
                'phases_ms_per_step': {'form_and_F_column': pr['trsm_ms'] / m, 'syrk_and_potrf_H': pr['syrk_ms'] / m, 'trsv_update': pr['trsv_update_ms'] / m},
                'executed_flops_per_step': ex_s, 'executed_tflops': ex_s * steps / el_s / 1e12,
                'f1_equivalent_steps_per_s': steps / el_s, 'f1_flops_per_step': dense,
                'iterate_rel_diff_vs_default': float(np.linalg.norm(sol_s - sol) / np.linalg.norm(sol))},
                'note': 'optional mode, not the reference operation sequence per step; next to, never instead of, the value above'}
            zs.free()
            # second level (gpk_gn_gram_prepare): the bordered matrix assembled from the Gram blocks of W1, W2 in O(n_z^2) per step --
            # neither the solve of A(z) nor the product S^T S runs; the F column is still solved, the Cholesky of H unchanged
            ctx.synchronize(); t0 = time.perf_counter()
            sp.prepare_gram()
            ctx.synchronize(); gram_setup = 1e3 * (time.perf_counter() - t0)
            zs = ctx.array(z0)
            for _ in range(warmup):
                ctx.gn_step(sp, zs)
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                ctx.gn_step(sp, zs)
            ctx.synchronize(); el_g = time.perf_counter() - t0
            sol_g = zs.download()
            out['structured_step']['gram_level'] = {
                'value': steps / el_g, 'unit': 'GN steps/s', 'ms_per_step': 1e3 * el_g / steps, 'setup_ms': sp.structured_prepare_ms + gram_setup,
                'iterate_rel_diff_vs_default': float(np.linalg.norm(sol_g - sol) / np.linalg.norm(sol))}
            zs.free(); sp.release_workspace()
            for a_ in (sp.W1, sp.W2, sp.v0, sp.G, sp.pvec, sp.Dinv, sp.Dinv2, sp.Wa, sp.Ha):
                if a_ is not None:
                    a_.free()
        except Exception as e:                                    # noqa: BLE001 -- reported, the workload's value survives
            out['structured_step'] = {'error': f'{type(e).__name__}: {e}'}
    if not args.no_cpu_baseline:
        from oracle import gp_oracle as O
        Ls = [tril_inplace(T.download()) for T in factors]
        if workload == 'c3':
            sysm, Ls_o = O.BurgersSystem(P['p0'], P['p1'], P['f'], P['g']), Ls
        else:
            sysm, Ls_o = O.DarcySystem(P['f'], P['g'], P['data'], P['p0']), [Ls[1], Ls[0]]   # oracle order: [L_a, L_u]
        attach_cpu_baseline(out, sysm, Ls_o, z0, f'orders {[T.rows for T in factors]}, n_z={nz}', dev_first, elapsed / steps)
    ctx.close()
    return out


# ------------------------------------------------------------------------------------------------------ sharded
def run_sharded(args, workload, steps=None, warmup=None, solo=False):
    """solo: this process alone (world 1, no collective, whatever the job's size) -- the 1-GPU point of the strong-scaling series,
    measured by rank 0 inside a multi-rank job while the other ranks wait"""
    import torch
    import torch.distributed as dist
    import gpk
    from gpk._lib import GNProblemStruct
    from gpk.sharded import Comm, GpuBlockOps, ShardedFactorSolve

    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    world = 1 if solo else int(os.environ.get('WORLD_SIZE', '1'))
    rank = 0 if solo else int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    Nd, Nb, _, desc = WORKLOADS[workload]
    N, nz = 2 * Nd + Nb, Nd
    ctx = gpk.Context(local)
    ops = GpuBlockOps(ctx)                                        # libgpk now runs on torch's current stream
    comm = Comm()
    if solo:
        comm.on, comm.rank, comm.world = False, 0, 1
    solver = ShardedFactorSolve(ops, comm, nb=args.panel)
    # Two executors of the same schedule (the plan of gpk_mg_plan_potrf): 'native' (default) = gpk_mg_* behind the C ABI, HIP streams
    # and events, RCCL called from C on a communicator of its own (unique id shipped through the process group); 'python' =
    # gpk/sharded.py over torch.distributed collectives (GPK_BENCH_SHARDED=python)
    engine = os.environ.get('GPK_BENCH_SHARDED', 'native')
    mgpu = None
    engine_note = ''
    if engine == 'native':
        from gpk.mg import MultiGpu
        err = ''
        try:
            mgpu = MultiGpu(ctx, rank, world, panel=args.panel, comm=os.environ.get('GPK_BENCH_COMM', 'rccl') if world > 1 else None)   # ('staged': several ranks on ONE GPU, tools/bench_two_ranks_one_gpu.sh)
        except Exception as e:                                    # noqa: BLE001 -- e.g. the RCCL library cannot be bound: every rank must take the same branch
            err = f'{type(e).__name__}: {e}'
        if world > 1 and comm.max_int(1 if err else 0, dev):
            if mgpu is not None:
                mgpu.close()
            mgpu = None
            engine_note = f' (native executor unavailable on some rank{": " + err if err else ""}; fell back)'
        elif err:
            raise RuntimeError(err)
    # First contact with the fabric (N > 1, native executor): bandwidth of the BOUND ncclBroadcast / ncclAllGather on one panel-sized
    # buffer (34000 x 512 doubles = 139 MB) per root, and how many ranks the communicator delivers -- so that a slow curve can be
    # read from the line (GB/s per link against ~153 GB/s of one xGMI link) instead of guessed.
    preflight = None
    wall = {}
    t_run0 = time.perf_counter()
    if mgpu and world > 1 and os.environ.get('GPK_BENCH_PREFLIGHT', '1') == '1':
        try:
            preflight = mgpu.preflight(int(os.environ.get('GPK_PREFLIGHT_BYTES', str(139 * 2 ** 20))), 2)
        except Exception as e:                                    # noqa: BLE001 -- reported; a failing collective fails the run below anyway
            preflight = {'error': f'{type(e).__name__}: {e}'}
        wall['preflight_s'] = time.perf_counter() - t_run0
    # the A/B probes below share ONE wall budget (GPK_PROBE_BUDGET_S, default 90 s): before every probe all ranks agree (max over ranks)
    # on the time spent so far; past the budget the remaining probes are skipped and the defaults kept -- replicas + sharded run + solo
    # run + CPU parity then fit inside GPK_SHARDED_TIMEOUT by construction
    probe_budget = float(os.environ.get('GPK_PROBE_BUDGET_S', '90'))
    t_probe = [0.0]

    def probe_allowed(name):
        spent = comm.max_float(t_probe[0], dev) if world > 1 else t_probe[0]
        if spent > probe_budget:
            mode_probe.setdefault('skipped_over_budget', []).append(name)
            return False
        return True
    Xd, Xb, f, g, z0 = synthetic_problem(Nd, Nb)                  # identical on every rank (seeded)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    tXd, tXb, tf, tg, z = t(Xd), t(Xb), t(f), t(g), t(z0)
    ld = ((N + 15) // 16) * 16
    Theta = torch.empty((N, ld), dtype=torch.float64, device=dev)
    kp = gpk.device.kernel_params('Gaussian', SIGMA)
    ratios = (C.c_double * 3)()
    nugget = 1e-13
    mode_probe = {}

    def set_mode(lookahead=None, shard_hb=None, overlap_s=None):
        if overlap_s is not None and mgpu:
            mgpu.set_option('overlap_s', int(overlap_s))
        if lookahead is not None:
            if mgpu:
                mgpu.set_option('lookahead', int(lookahead))
            solver.lookahead = bool(lookahead)
        if shard_hb is not None:
            if mgpu:
                mgpu.set_option('shard_hb', int(shard_hb))
            solver.shard_hb = bool(shard_hb)

    def factor_once():
        # assembly shards trivially (every entry depends on two points); at 9.2 GB / 5 TB/s it is cheaper to let
        # every rank write the whole matrix than to communicate anything
        torch.cuda.synchronize(); comm.barrier(); t0 = time.perf_counter()
        ctx._chk(ctx.lib.gpk_assemble(ctx.h, 0, 0, kp, tXd.data_ptr(), Nd, tXb.data_ptr(), Nb, nugget, 2, Theta.data_ptr(), ld, ratios))
        torch.cuda.synchronize(); a_ms = 1e3 * (time.perf_counter() - t0)
        comm.barrier(); t0 = time.perf_counter()
        inf = mgpu.potrf(Theta.data_ptr(), N, ld) if mgpu else solver.potrf(Theta, N)
        torch.cuda.synchronize(); c_ms = 1e3 * (time.perf_counter() - t0)
        return a_ms, comm.max_float(c_ms, dev), inf

    while True:
        asm_ms, chol_ms, info = factor_once()
        if info == 0 or nugget >= 1e-8:
            break
        nugget *= 10.0
    chol_first_ms = chol_ms                                       # (first factorisation of the process: code objects, buffers)
    if world == 1:
        _, chol_ms, _ = factor_once()
    if world > 1 and os.environ.get('GPK_BENCH_MODE_PROBE', '1') == '1' and probe_allowed('cholesky_theta'):
        # More than one rank: the factorisation is timed with BOTH plans -- look-ahead (default) and the strictly sequential
        # factor -> broadcast -> update -- and the faster one (max over ranks, so every rank decides alike) is kept for what
        # follows; both times are reported.  (Look-ahead hides broadcasts behind updates but its panel kernels share the CUs
        # with the update GEMMs of the same GPU; which effect wins depends on the fabric.)
        # (the factorisation above was the first of the process -- side streams, events, transfer buffers, the first broadcast -- and
        # is reported as one_time_ms.cholesky_theta_first_call; both plans are timed WARM here: one untimed run of the sequential plan,
        # then look-ahead and sequential once each)
        chol_first_ms = chol_ms
        tp0 = time.perf_counter()
        set_mode(lookahead=0); factor_once()
        set_mode(lookahead=1); _, c1, info1 = factor_once()
        set_mode(lookahead=0); _, c2, info2 = factor_once()
        t_probe[0] += time.perf_counter() - tp0
        mode_probe['cholesky_theta_ms'] = {'lookahead': c1, 'sequential': c2, 'first_call_lookahead': chol_first_ms}
        if c2 < c1 and info2 == info:
            chol_ms = c2
        else:
            chol_ms = c1
            set_mode(lookahead=1)
        mode_probe['lookahead_kept'] = bool(solver.lookahead)
    ps = GNProblemStruct()
    ps.system, ps.Nd, ps.Nb, ps.Ndata = 0, Nd, Nb, 0
    ps.p0, ps.p1, ps.pen_lambda = ALPHA, M_EXP, 0.0
    ps.rhs_f, ps.bdy_g, ps.data_u = tf.data_ptr(), tg.data_ptr(), None
    ps.L, ps.ldl, ps.L2, ps.ldl2 = Theta.data_ptr(), ld, None, 0
    lds = ((nz + 1 + 15) // 16) * 16
    S = torch.empty((N, lds), dtype=torch.float64, device=dev)
    Hb = torch.empty((nz + 1, lds), dtype=torch.float64, device=dev)
    delta = torch.empty(nz, dtype=torch.float64, device=dev)
    # one-time companion of the factor: inverses of its 1024-row diagonal blocks (the column solves then are GEMMs only);
    # S2 receives the solved block out of place and is zero where the solve never writes
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Dinv = ops.trtri_diag(Theta, N, block=gpk.device.dinv_block_for(N))
    torch.cuda.synchronize(); dinv_ms = 1e3 * (time.perf_counter() - t0)
    S2 = torch.zeros((N, lds), dtype=torch.float64, device=dev)
    ps.Dinv, ps.Dinv2, ps.dinv_block = Dinv.data_ptr(), None, gpk.device.dinv_block_for(N)
    if mgpu:
        step_only = lambda: mgpu.gn_step(ps, z.data_ptr(), 1.0, S.data_ptr(), lds, S2.data_ptr(), Hb.data_ptr(), lds, delta.data_ptr())[0]
    else:
        step_only = lambda: solver.gn_step(ps, nz, N, Theta, z, S, Hb, delta, 1.0, rev=True, Dinv=Dinv, S2=S2)[0]
    # the timed step is the product's per-iteration sequence here too: the (collective) step, which reports the loss of the iterate it
    # starts from by true substitution (replicated: one vector); GPK_SEPARATE_LOSS=1 adds a loss call of its own per step
    with_loss = os.environ.get('GPK_SEPARATE_LOSS', '0') == '1'

    def loss_of_iterate():
        w = torch.cat([ALPHA * z ** M_EXP - tf, z, tg])           # F(z), src/PDEs.py:84-85 of the reference
        ops.trsv(Theta, N, w, False)
        return float((w * w).sum().item())

    def step():
        l_in = step_only()
        return loss_of_iterate() if with_loss else l_in
    # first step from z0 for `parity` (every rank takes part: the step is collective; it is also the untimed first step that pays
    # the one-time allocations of the executor); the iterate is then reset to z0
    loss0_dev = step_only()
    z1_dev = z.cpu().numpy().copy()
    loss1_dev = loss_of_iterate()                                 # loss(z1) = || L^{-1} F(z1) ||^2 by true substitution
    z.copy_(t(z0))
    losses = []
    warmup_run = 0
    if world > 1 and os.environ.get('GPK_BENCH_MODE_PROBE', '1') == '1':
        # the same for the Cholesky of the bordered Gauss-Newton matrix: ONE A/B pair, each variant run twice (the first run of a variant
        # pays its one-time allocations and is not the one compared), the faster one kept (all ranks alike); these steps count as warm-up
        def ab(names_flags, setter):
            nonlocal warmup_run
            times = {}
            tp0 = time.perf_counter()
            for name, flag in names_flags * 2:
                setter(flag)
                comm.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
                losses.append(step_only())
                torch.cuda.synchronize()
                times[name] = comm.max_float(1e3 * (time.perf_counter() - t0), dev)
                warmup_run += 1
            t_probe[0] += time.perf_counter() - tp0
            return times
        if probe_allowed('cholesky_of_Hb'):
            times = ab((('replicated', 0), ('panel_sharded', 1)), lambda f: set_mode(shard_hb=f))
            set_mode(shard_hb=int(times['panel_sharded'] < times['replicated']))
            mode_probe['step_ms_by_cholesky_of_Hb'] = times
        mode_probe['shard_hb_kept'] = bool(solver.shard_hb)
        if mgpu and probe_allowed('exchange_of_S'):
            # and for the exchange of the column shards of S: one all-gather, or one broadcast per shard on the communication stream with
            # the block-row products of Hb issued behind the arrivals (native executor only)
            # (round 6: third form where the point-to-point entry points are bound -- the DIRECT exchange, one transfer per peer inside one
            # ncclGroupStart / ncclGroupEnd: on xGMI every peer has a link of its own)
            forms = (('all_gather', 0), ('broadcasts_chased_by_products', 1)) + ((('direct_p2p', 2),) if mgpu.has_p2p() else ())
            xt = ab(forms, lambda f: set_mode(overlap_s=f))
            keep = min(forms, key=lambda nf: xt[nf[0]])
            set_mode(overlap_s=keep[1])
            mode_probe['step_ms_by_exchange_of_S'] = xt
            mode_probe['overlap_s_kept'] = keep[0]
        mode_probe['probe_wall_s'] = t_probe[0]
    for _ in range(max(warmup - warmup_run, 0)):
        losses.append(step())
        warmup_run += 1
    wall['setup_and_probes_s'] = time.perf_counter() - t_run0
    comm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(step())
    torch.cuda.synchronize(); comm.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    # accuracy (replicated work, tiny)
    sol = z.cpu().numpy()
    pts_l2 = float(np.sqrt(np.sum((u_true(Xd[:, 0], Xd[:, 1]) - sol) ** 2) / Nd))
    coeff = t(np.concatenate([ALPHA * sol ** M_EXP - f, sol, g]))
    ops.trsv(Theta, N, coeff, False); ops.trsv(Theta, N, coeff, True)
    Xt = test_grid(60)
    tXt = t(Xt)
    ext = torch.empty(Xt.shape[0], dtype=torch.float64, device=dev)
    ctx._chk(ctx.lib.gpk_extend(ctx.h, 0, 0, kp, tXt.data_ptr(), Xt.shape[0], tXd.data_ptr(), Nd, tXb.data_ptr(), Nb,
                                coeff.data_ptr(), ext.data_ptr()))
    torch.cuda.synchronize()
    test_l2 = float(np.sqrt(np.sum((u_true(Xt[:, 0], Xt[:, 1]) - ext.cpu().numpy()) ** 2) / Xt.shape[0]))
    # the Cholesky of the bordered matrix alone (one rank only: an input of the multi-GPU model, bench_model.py): its time does not depend
    # on the data, so it is taken on the identity of that order in Hb (free after the last step)
    potrf_hb_ms = None
    if world == 1:
        Hb.zero_(); Hb.diagonal().fill_(1.0)
        ops.potrf(Hb, 0, nz + 1)                                  # (warm)
        Hb.zero_(); Hb.diagonal().fill_(1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.potrf(Hb, 0, nz + 1)
        torch.cuda.synchronize(); potrf_hb_ms = 1e3 * (time.perf_counter() - t0)
    if not with_loss:                                             # close the history: the in-step value is the loss of the iterate a step STARTS from
        losses.append(loss_of_iterate())
    out = None
    if rank == 0:
        rate = f1_flops(N, nz) * steps / elapsed / 1e12
        # flops the step EXECUTES, summed over the ranks (never dense F1 flops over the shortened time): every rank's column
        # shard of the solve (same model as the one-GPU roofline, per shard), the lower-tile product with the structural zeros skipped
        # (row-block sharded: the same tiles, distributed) and the Cholesky of Hb counted ONCE -- where it is replicated the copies are
        # waste, not work, and lower the fraction
        db = gpk.device.dinv_block_for(N)
        bounds = solver.column_ranges_lz(nz + 1, nz, N)
        solve_fl = sum(trsm_dinv_executed_flops(N, nz, db, c0=bounds[r], c1=bounds[r + 1])[0] for r in range(world) if bounds[r + 1] > bounds[r])
        prod_fl = syrk_executed_flops(N, nz)
        chol_fl = (nz + 1) ** 3 / 3.0
        executed = solve_fl + prod_fl + chol_fl
        ex_rate = executed * steps / elapsed / 1e12
        # HBM-side bytes per step of the step's two product families (solve-phase launches + the product S^T S) from the stored PMC passes of
        # THIS workload on one GPU (profiles/rNN_pmc_*_<workload>.json); null with more than one rank (no pass exists) or without a stored pass
        t1, s1 = stored_pmc_traffic('trsm_gemm', workload)
        t2, s2 = stored_pmc_traffic('syrk', workload)
        step_traffic = (t1 + t2) if (world == 1 and t1 is not None and t2 is not None) else None
        step_traffic_source = (f'solve launches: {s1}; product: {s2}' if step_traffic is not None else
                               ('no stored PMC pass for more than one rank' if world > 1 else f'{s1}; {s2}'))
        out = {
            'metric': 'Gauss-Newton steps/sec + L2 error, NonLinElliptic2d at N_domain points',
            'value': steps / elapsed, 'unit': 'GN steps/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup_run,
            'ms_per_step': 1e3 * elapsed / steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': desc, 'N_domain': Nd, 'N_boundary': Nb, 'theta_order': N, 'unknowns': nz, 'kernel': 'Gaussian',
                       'kernel_parameter': SIGMA, 'nugget': nugget, 'nugget_type': 'adaptive', 'seed': 0,
                       'timed_sequence': TIMED_SEQUENCE[with_loss].replace('gpk_gn_step', 'gpk_mg_gn_step' if mgpu else 'sharded gn_step'),
                       'parallelism': solver.describe(world, gpk.device.dinv_block_for(N)),
                       'executor': (f'native: gpk_mg_potrf / gpk_mg_gn_step (C ABI), collectives = {getattr(mgpu, "comm_kind", "none (one rank)")}'
                                    if mgpu else 'python: gpk/sharded.py over torch.distributed' + engine_note),
                       'formulation': 'F1 (TRSM + SYRK + POTRF(H) + TRSV every step; only the factor of Theta and the inverses of its diagonal '
                                      'blocks are reused across steps); structural zeros of A(z) skipped as on one GPU, column shards cut by '
                                      'work; f1_tflops is the dense-equivalent rate, roofline.achieved the executed one'},
            'l2_error': {'pts_L2_err': pts_l2, 'test_L2_err': test_l2, 'gn_steps_run': warmup_run + steps,
                         'loss_first': losses[0], 'loss_last': losses[-1], 'chol_info': info},
            'f1_tflops': rate,
            'one_time_ms': {'assembly_per_rank': asm_ms, 'cholesky_theta_sharded': chol_ms, 'cholesky_theta_first_call': chol_first_ms, 'diagonal_block_inverses': dinv_ms},
            'cholesky_hb_alone_ms': potrf_hb_ms,
            'mode_probe': mode_probe or None, 'preflight': preflight, 'wall_s': wall,
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_f64_kernel (whole step: solve + product + Cholesky of Hb), flops EXECUTED summed over the ranks',
                         'achieved': ex_rate, 'peak': FP64_MFMA_PEAK_TFLOPS * world, 'unit': 'TFLOP/s',
                         'frac': ex_rate / (FP64_MFMA_PEAK_TFLOPS * world), 'traffic': step_traffic, 'traffic_source': step_traffic_source,
                         'flops_per_step': {'solve': solve_fl, 'product': prod_fl, 'cholesky_H': chol_fl},
                         'dense_equivalent_tflops': rate},
            'roofline_cholesky_theta': {'bound': 'mfma', 'kernel': 'sharded Cholesky of Theta (panel kernels + trailing GEMM updates; broadcasts at N > 1)',
                                        'achieved': N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12, 'peak': FP64_MFMA_PEAK_TFLOPS * world, 'unit': 'TFLOP/s',
                                        'frac': N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world), 'ms': chol_ms},
            'cpu_baseline': None, 'parity': None,
        }
        # CPU oracle on the device's factor (B2 only: the reference operation sequence needs minutes at order 34000), `parity` beside it
        need_gb = 3.2 * 8.0 * N * N / 1e9 + 4.0 * 8.0 * N * (nz + 1) / 1e9
        if not args.no_cpu_baseline and not args.no_sharded_parity and not solo:   # (solo = the 1-GPU point inside an N-rank job: the N-rank run carries the parity)
            if host_mem_available_gb() > need_gb + 16:
                from oracle import gp_oracle as O
                Lh = np.ascontiguousarray(tril_inplace(Theta.cpu().numpy()[:, :N]))
                attach_cpu_baseline(out, O.EllipticSystem(ALPHA, M_EXP, f, g), [Lh], z0,
                                    f'N={N}, n_z={nz}', (z1_dev, loss0_dev, loss1_dev), elapsed / steps, with_b1=False, with_mkl=False)
                del Lh
            else:
                out['parity'] = {'skipped': f'host memory available {host_mem_available_gb():.0f} GB < {need_gb + 16:.0f} GB needed by the CPU oracle at this size'}
    if mgpu:
        mgpu.close()
    del S, S2, Hb, Theta, Dinv
    torch.cuda.empty_cache()
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=['auto', 'c1', 'c2', 'c3', 'c4', 'c5', 'n10k'], default='auto')
    ap.add_argument('--panel', type=int, default=512, help='panel width of the sharded Cholesky')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-structured', action='store_true', help='skip the secondary measurement of the optional structured solve')
    ap.add_argument('--no-sharded-config', action='store_true', help='skip the BASELINE config 5 run reported under sharded_config')
    ap.add_argument('--no-n10k', action='store_true', help='skip the north-star target size (N_domain = 10000 on one GPU) reported under n10k')
    ap.add_argument('--no-c3c4', action='store_true', help='skip BASELINE configs 3 (Burgers) and 4 (Darcy) reported under c3 / c4')
    ap.add_argument('--no-sharded-parity', action='store_true', help='skip the CPU oracle step (about a minute, ~40 GB of host memory) behind sharded_config.parity')
    ap.add_argument('--no-replicas', action='store_true', help='N > 1: skip the secondary measurement of N independent config-2 replicas')
    ap.add_argument('--sharded-path', action='store_true', help='use the multi-rank schedule for the primary workload')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # the bare command `python3 bench.py --gpus N ...`: start the N ranks ourselves (bench_launch.py).  This process has made
        # no GPU call and makes none: the ranks are fresh child processes, rank 0's compact line is relayed as our last stdout line.
        import bench_launch
        sys.exit(bench_launch.launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch either the bare command (it starts its own ranks) or '
                 f'python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...')
    if os.environ.get('GPK_BENCH_STUBS'):
        # CPU flow tests only (tests/test_bench_flow.py): a file that REPLACES the measuring functions of this module with stand-ins
        # so that the launch / rendezvous / one-line logic runs without a GPU.  The line says so: `data` is not "synthetic".
        with open(os.environ['GPK_BENCH_STUBS']) as fh:
            exec(compile(fh.read(), os.environ['GPK_BENCH_STUBS'], 'exec'), globals())
    use_pg = world > 1 or os.environ.get('GPK_FORCE_PG') == '1'
    if use_pg:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(os.environ.get('GPK_BENCH_BACKEND', 'nccl'))   # (gloo only in the CPU flow test)
        dist.barrier()                                            # creates the communicator NOW (RCCL prints its version banner to stdout
                                                                  # when it does: it must not come after the JSON line)
    rank = int(os.environ.get('RANK', '0'))
    SECONDARY_KEYS = ('value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'config', 'l2_error', 'f1_tflops', 'phases_ms_per_step',
                      'one_time_ms', 'roofline', 'roofline_syrk', 'roofline_assembly', 'roofline_cholesky_theta', 'flops_counted_by_library',
                      'step_executed', 'mode_probe', 'cpu_baseline', 'parity', 'cholesky_hb_alone_ms', 'structured_step')
    pick = lambda d: {k: d[k] for k in SECONDARY_KEYS if k in d}

    def prediction(one, preflight, ranks):
        """bench_model.table from THIS job's 1-GPU point of config 5 (`one`: a run_sharded result on one rank) and the preflight of the bound
        collectives (None at N = 1: one xGMI link x 0.8 assumed) -- what the multi-GPU choices should take, printed before / beside what they do"""
        import bench_model
        one_gpu = None
        try:
            if isinstance(one, dict) and one.get('ms_per_step'):
                one_gpu = {'step_ms': one['ms_per_step'], 'cholesky_theta_ms': one['one_time_ms']['cholesky_theta_sharded'],
                           'source': 'the 1-GPU point of config 5 measured in this job'}
                if one.get('cholesky_hb_alone_ms'):
                    one_gpu['cholesky_hb_ms'] = one['cholesky_hb_alone_ms']
                fl = one['roofline'].get('flops_per_step') if isinstance(one.get('roofline'), dict) else None
                if isinstance(fl, dict):
                    one_gpu.update(solve_flops=fl['solve'], product_flops=fl['product'])
            return bench_model.table(one_gpu=one_gpu, fabric=bench_model.fabric_from_preflight(preflight), ranks=ranks)
        except Exception as e:                                    # noqa: BLE001 -- a model, never a reason to lose the measurement
            return {'error': f'{type(e).__name__}: {e}'}

    def contract_line(obj, n_gpus, scaling):
        """a run_system result dressed as the driver's line"""
        return dict({'metric': 'Gauss-Newton steps/sec + L2 error at N_domain points', 'n_gpus': n_gpus, 'higher_is_better': True,
                     'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic'}, **obj)

    if world > 1 and args.workload in ('auto', 'c5') and not args.sharded_path:
        # ---- N > 1 ranks: `value` IS the sharded BASELINE config 5 (the north star's scaling series: fixed total work, strong scaling).
        # The same job also times config 5 on rank 0 ALONE (vs_1gpu: a self-contained strong-scaling point per job) and, as a secondary
        # object, N independent replicas of config 2 (what `value` used to be up to round 3).
        # Order: the replicas first (no data-path collective: robust); their line is what survives if the sharded run fails or hangs
        # (AbortWatch prints it with the error attached and `value` then IS the replica figure, labelled as such).
        from gpk.sharded import Comm
        import torch.distributed as dist
        rep = None
        if not args.no_replicas:
            rep = run_single(args, 'c2', Comm())
        # what survives if the sharded run fails or hangs: a line whose `value` is NULL (the key means "sharded config 5" at N > 1 --
        # `value_workload` says so machine-readably -- and is never filled with another workload's figure); the replicas' aggregate
        # stays readable under replicas_c2
        fb = None
        if rank == 0:
            fb = {'metric': 'Gauss-Newton steps/sec + L2 error, NonLinElliptic2d at N_domain points', 'value': None, 'unit': 'GN steps/s',
                  'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'strong',
                  'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'value_workload': 'c5',
                  'config': {'workload': WORKLOADS['c5'][3], 'parallelism': f'sharded over {world} ranks (did not complete)'},
                  'fallback': 'the sharded BASELINE config 5 run did not complete (see sharded_config.error): value is null; the aggregate of the '
                              'N independent config-2 replicas measured before it is under replicas_c2'}
            if rep is not None:
                fb['replicas_c2'] = {k: rep[k] for k in ('value', 'unit', 'n_gpus', 'ms_per_step', 'scaling', 'config', 'l2_error', 'roofline') if k in rep}
        watch = AbortWatch(rank, fb, use_store=True)
        solo = None
        try:
            out = run_sharded(args, 'c5')
            if rank == 0:
                try:
                    solo = run_sharded(args, 'c5', steps=min(args.steps, 3), warmup=1, solo=True)
                except Exception as e:                            # noqa: BLE001 -- reported
                    solo = {'error': f'{type(e).__name__}: {e}'}
            dist.barrier()
        except Exception as e:                                    # noqa: BLE001 -- reported, not swallowed
            watch.fail(f'rank {rank}: {type(e).__name__}: {e}')   # does not return (peers may be blocked in a collective)
        watch.stop()
        if out is not None:
            if solo and 'value' in solo:
                out['one_gpu_same_job'] = pick(solo)
                out['vs_1gpu'] = out['value'] / solo['value']
                out['parallel_efficiency'] = out['vs_1gpu'] / world
            else:
                out['one_gpu_same_job'] = solo
                out['vs_1gpu'] = None
            # the host model beside the measurement (bench_model.py): what this rank count should have taken, per choice
            out['predicted'] = prediction(solo, out.get('preflight'), tuple(sorted({2, 4, 8, world})))
            mine = out['predicted'].get(str(world)) if isinstance(out['predicted'], dict) else None
            if isinstance(mine, dict):
                out['predicted_vs_1gpu'] = mine.get('predicted_vs_1gpu')
                mp = out.get('mode_probe')
                if isinstance(mp, dict):
                    for key, src in (('cholesky_theta_ms', 'cholesky_theta_ms'), ('step_ms_by_cholesky_of_Hb', 'cholesky_of_Hb_ms'),
                                     ('step_ms_by_exchange_of_S', 'exchange_of_S_ms')):
                        if isinstance(mp.get(key), dict):
                            mp[key]['expected_ms'] = mine.get(src)
            out['value_workload'] = 'c5'
            out['scaling_series'] = ('BASELINE config 5, strong scaling: this line\'s `value` at n_gpus > 1; at n_gpus = 1 the default line reports config 2 as '
                                     '`value` (the configuration the metric is quoted on, it fits one GPU) and config 5 on one GPU under `sharded_config`')
            if rep is not None and rep.get('value') is not None:
                out['replicas_c2'] = {k: rep[k] for k in ('value', 'unit', 'n_gpus', 'ms_per_step', 'scaling', 'config', 'l2_error') if k in rep}
    elif args.sharded_path or args.workload == 'c5':
        out = run_sharded(args, args.workload if args.workload != 'auto' else 'c2')
        if out is not None:
            out['value_workload'] = args.workload if args.workload != 'auto' else 'c2'
    elif args.workload in ('c3', 'c4'):
        out = contract_line(run_system(args, args.workload), 1, 'weak')
        out['value_workload'] = args.workload
    else:
        workload = args.workload if args.workload != 'auto' else 'c2'
        from gpk.sharded import Comm
        out = run_single(args, workload, Comm() if world > 1 else None)
        if out is not None:
            out['value_workload'] = workload
        # GPK_BENCH_SECONDARY_CPU=0: the CPU oracle legs (cpu_baseline + parity) of the SECONDARY workloads are skipped -- the
        # primary workload keeps its own; used by the -m gpu test of the default command, which checks the line, not the CPU
        sec_args = args
        if os.environ.get('GPK_BENCH_SECONDARY_CPU', '1') == '0':
            sec_args = argparse.Namespace(**vars(args)); sec_args.no_cpu_baseline = True
        if args.workload == 'auto' and world == 1 and not args.no_n10k:
            # north-star target size (N_domain = 10^4 on ONE GPU, >= 10x vs the CPU reference sequence): a secondary object of the
            # same line, its own CPU baselines beside it (B1 = reference operation sequence, ~1 min on the host cores; B2 = triangular)
            try:
                nk = run_single(sec_args, 'n10k', None, secondary=True, steps=min(args.steps, 4), warmup=1)
                out['n10k'] = pick(nk)
            except Exception as e:                                # noqa: BLE001 -- reported, the primary value survives
                out['n10k'] = {'error': f'{type(e).__name__}: {e}'}
        if args.workload == 'auto' and world == 1 and not args.no_c3c4:
            for name in ('c3', 'c4'):                             # BASELINE configs 3 and 4 under the same clock
                try:
                    out[name] = run_system(sec_args, name)
                except Exception as e:                            # noqa: BLE001 -- reported, the primary value survives
                    out[name] = {'error': f'{type(e).__name__}: {e}'}
        if args.workload == 'auto' and not args.no_sharded_config:
            watch = AbortWatch(rank, out, use_store=use_pg)
            try:                                                  # the value above must survive a failure of the secondary run
                sh = run_sharded(sec_args, 'c5', steps=min(args.steps, 3), warmup=1)
                if out is not None and sh is not None:
                    out['sharded_config'] = pick(sh)
                    if world == 1:                                # what 2 / 4 / 8 ranks should take from this 1-GPU point (one xGMI link x 0.8 assumed)
                        out['predicted'] = prediction(sh, None, (2, 4, 8))
            except Exception as e:                                # noqa: BLE001 -- reported, not swallowed
                msg = f'{type(e).__name__}: {e}'
                if use_pg:
                    watch.fail(f"rank {rank}: {msg}")             # does not return (peers may be blocked in a collective)
                if out is not None:
                    out['sharded_config'] = {'error': msg}
            watch.stop()
    # parity is a gate, not a remark: any object of the line whose device iterate is further than 1e-6 from the oracle's fails the run
    bad = []
    if out is not None:
        for key, obj in [('value', out)] + [(k, out.get(k)) for k in ('n10k', 'c3', 'c4', 'sharded_config', 'one_gpu_same_job')]:
            par = obj.get('parity') if isinstance(obj, dict) else None
            if isinstance(par, dict) and par.get('ok') is False:
                bad.append(key)
        out['parity_failed'] = bad or None
    # the ONE line goes out before anything that can still block (a peer that died after its last collective would otherwise
    # leave rank 0 in the final barrier with the result unprinted)
    if out is not None:
        if os.environ.get('GPK_BENCH_STUBS'):
            out['data'] = 'STUBBED (GPK_BENCH_STUBS): not a measurement'
        emit(out)
        if bad:
            print(f'bench.py: PARITY FAILURE -- device iterate further than {PARITY_TOL:g} (relative) from the CPU oracle in: {", ".join(bad)}',
                  file=sys.stderr, flush=True)
    if use_pg:
        import threading
        import torch.distributed as dist
        threading.Thread(target=lambda: (time.sleep(float(os.environ.get('GPK_FINAL_BARRIER_TIMEOUT', '120'))), sys.stdout.flush(), os._exit(0)),
                         daemon=True).start()
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        sys.exit(4)


if __name__ == '__main__':
    main()
