"""The multi-GPU Cholesky schedule as DATA (gpk_mg_plan_potrf, include/gpk_mg.h) -- CPU tests, no GPU, no process group.

The plan is what both executors run: the native one (csrc/gpk_mg.hip: HIP streams/events + RCCL) and the Python one
(gpk/sharded.py over torch.distributed).  Here it is checked as a schedule:
  * structure: every panel factored once by its owner, every owned block column updated once by every earlier panel,
    broadcasts in panel order on every rank (collectives must be issued in the same order everywhere), every WAIT refers
    to an event RECORDed earlier in the list (so issuing the list in order never blocks a hardware queue on the future);
  * look-ahead order: the owner of panel k+1 factors it (and hands it to the communication stream) BEFORE it applies
    panel k to its other columns;
  * dependencies are SUFFICIENT: all ranks are simulated in one process with numpy block operations, three in-order
    streams per rank, and the next operation drawn at random among all (rank, stream) heads that are allowed to run (a
    WAIT only after its event was recorded, a broadcast's receive only after the root's send) -- any legal interleaving
    the hardware may choose must give the Cholesky factor on every rank.  Transfer buffers are modelled (two slots per
    rank), so a missing buffer-reuse dependency shows up as a wrong factor.
"""
import numpy as np
import pytest
from scipy.linalg import solve_triangular

from gpk import mg

F, PK, BC, UP, UD, REC, WT = mg.FACTOR, mg.PACK, mg.BCAST, mg.UNPACK, mg.UPDATE, mg.RECORD, mg.WAIT


def _plans(n, nb, world, la):
    return [mg.plan_potrf(n, nb, world, r, la) for r in range(world)]


@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
@pytest.mark.parametrize('la', [False, True])
def test_plan_structure(world, la):
    n, nb = 1000, 64                                              # 16 panels, ragged last one
    nblk = -(-n // nb)
    for r, plan in enumerate(_plans(n, nb, world, la)):
        recorded = set()
        factored, updates, bcasts = [], set(), []
        for kind, a, b, s in plan:
            assert 0 <= s <= (2 if la else 0)
            if kind == REC:
                assert a not in recorded                          # every event is recorded once
                recorded.add(a)
            elif kind == WT:
                assert a in recorded, (r, a)                      # never a wait for something issued later
            elif kind == F:
                assert a % world == r
                factored.append(a)
            elif kind == UD:
                assert a % world == r and b < a and (a, b) not in updates
                updates.add((a, b))
            elif kind == BC:
                assert b == a % world
                bcasts.append(a)
            elif kind in (PK, UP):
                assert world > 1 and ((a % world == r) == (kind == PK))
        assert factored == [k for k in range(nblk) if k % world == r]
        assert updates == {(j, k) for j in range(nblk) if j % world == r for k in range(j)}
        assert bcasts == (list(range(nblk)) if world > 1 else [])


@pytest.mark.parametrize('world', [2, 4, 8])
def test_lookahead_order(world):
    n, nb = 2048, 64
    nblk = n // nb
    for r, plan in enumerate(_plans(n, nb, world, True)):
        pos = {}
        for i, (kind, a, b, s) in enumerate(plan):
            pos.setdefault((kind, a, b), i)
        for k in range(nblk - 1):
            if (k + 1) % world != r:
                continue
            later = [pos[(UD, j, k)] for j in range(k + 2, nblk) if j % world == r]
            # the look-ahead column first, on the panel stream; then the factorisation; only then the other columns
            assert plan[pos[(UD, k + 1, k)]][3] == 1 and plan[pos[(F, k + 1, 0)]][3] == 1
            assert pos[(UD, k + 1, k)] < pos[(F, k + 1, 0)] < pos[(PK, k + 1, 0)]
            if later:
                assert pos[(PK, k + 1, 0)] < min(later)
                assert all(plan[p][3] == 0 for p in later)        # trailing updates on the main stream
        # broadcasts travel on their own stream
        assert all(s == 2 for kind, a, b, s in plan if kind in (BC, UP))


def _simulate(A0, n, nb, world, la, rng, adversarial=False):
    """all ranks in one process; returns the list of factors.  adversarial: instead of a uniform draw, a random PRIORITY order
    of the (rank, stream) queues -- a queue runs only when nothing of higher priority can, i.e. whole streams are starved for as
    long as the dependencies allow (the interleavings a uniform draw practically never produces)."""
    plans = _plans(n, nb, world, la)
    prio = {(r, s): p for p, (r, s) in enumerate(sorted(((r, s) for r in range(world) for s in range(3)), key=lambda _: rng.rand()))}
    A = [A0.copy() for _ in range(world)]
    buf = [[None, None] for _ in range(world)]
    # per (rank, stream) in-order queues
    queues = [[[op for op in plans[r] if op[3] == s] for s in range(3)] for r in range(world)]
    head = [[0, 0, 0] for _ in range(world)]
    events = [set() for _ in range(world)]
    sent = {}                                                     # panel -> image the root put on the wire
    steps = 0
    while True:
        ready = []
        for r in range(world):
            for s in range(3):
                if head[r][s] >= len(queues[r][s]):
                    continue
                kind, a, b, _ = queues[r][s][head[r][s]]
                if kind == WT and a not in events[r]:
                    continue
                if kind == BC and b != r and a not in sent:
                    continue                                      # a receive completes only after the root has sent
                ready.append((r, s))
        if not ready:
            break
        r, s = min(ready, key=prio.get) if adversarial else ready[rng.randint(len(ready))]
        kind, a, b, _ = queues[r][s][head[r][s]]
        head[r][s] += 1
        steps += 1
        k = b if kind == UD else a
        k0 = k * nb; kb = min(nb, n - k0)
        M = A[r]
        if kind == F:
            blk = M[k0:k0 + kb, k0:k0 + kb]
            Lkk = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
            M[k0:k0 + kb, k0:k0 + kb] = np.tril(Lkk) + np.triu(blk, 1)
            if k0 + kb < n:
                M[k0 + kb:n, k0:k0 + kb] = solve_triangular(Lkk, M[k0 + kb:n, k0:k0 + kb].T, lower=True).T
        elif kind == PK:
            buf[r][k & 1] = M[k0:n, k0:k0 + kb].copy()
        elif kind == BC:
            if b == r:
                sent[a] = buf[r][k & 1].copy()
            else:
                buf[r][k & 1] = sent[a].copy()
        elif kind == UP:
            M[k0:n, k0:k0 + kb] = buf[r][k & 1]
        elif kind == UD:
            j0 = a * nb; jb = min(nb, n - j0)
            M[j0:n, j0:j0 + jb] -= M[j0:n, k0:k0 + kb] @ M[j0:j0 + jb, k0:k0 + kb].T
        elif kind == REC:
            events[r].add(a)
    assert all(head[r][s] == len(queues[r][s]) for r in range(world) for s in range(3)), 'schedule deadlocked'
    return [np.tril(M) for M in A]


@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
@pytest.mark.parametrize('la', [False, True])
def test_any_legal_interleaving_gives_the_factor(world, la):
    rng = np.random.RandomState(100 * world + la)
    n, nb = 300, 32                                               # 10 panels, ragged last one (12 columns)
    Mx = rng.normal(size=(n, n))
    A0 = Mx @ Mx.T + n * np.eye(n)
    Lref = np.linalg.cholesky(A0)
    for trial in range(24 if la else 1):
        Ls = _simulate(A0, n, nb, world, la, rng, adversarial=trial >= 4)
        for L in Ls:
            assert np.linalg.norm(L - Lref) <= 1e-12 * np.linalg.norm(A0)


def test_single_panel_and_empty():
    assert mg.plan_potrf(0, 64, 2, 0, True) == []
    assert [op[0] for op in mg.plan_potrf(50, 64, 1, 0, True)] == [F]         # one panel: nothing to look ahead to
    p1 = mg.plan_potrf(50, 64, 2, 1, True)
    assert [op[0] for op in p1] == [BC, UP]


def test_column_bounds_match_the_numpy_formula():
    """gpk_mg_column_bounds against the formula the Python schedule used before the native one existed (cumsum +
    searchsorted): identical boundaries, monotone, covering, wider first shard (early columns are cheap)."""
    def ref(ncols, lead, rows, P, align):
        c = np.arange(ncols)
        start = np.maximum(0, lead - 1 - c)
        w = np.cumsum((rows - start).astype(np.float64) ** 2)
        bounds = [0]
        for r in range(1, P):
            cut = int(np.searchsorted(w, w[-1] * r / P))
            cut = min(max(-(-cut // align) * align, bounds[-1]), ncols)
            bounds.append(cut)
        bounds.append(ncols)
        return bounds
    for ncols, lead, rows, P, align in [(16001, 16000, 34000, 8, 128), (4001, 4000, 8400, 2, 128), (71, 70, 160, 3, 4),
                                        (501, 500, 1080, 3, 64), (10001, 10000, 21000, 4, 128), (5, 4, 12, 8, 128)]:
        got = mg.column_bounds(ncols, lead, rows, P, align)
        assert got == ref(ncols, lead, rows, P, align)
        assert got[0] == 0 and got[-1] == ncols and all(a <= b for a, b in zip(got, got[1:]))
    b = mg.column_bounds(16001, 16000, 34000, 8, 128)
    assert b[1] - b[0] > b[-1] - b[-2]
