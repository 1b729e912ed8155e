"""Predicted times of the multi-GPU choices of config 5 -- a pure-host model, so that the first run on a fabric is falsifiable.

Nobody has run the sharded path on more than one GPU (the pool has 1-GPU boxes).  `gpk_mg_*` makes four choices whose best value
depends on the fabric: (1) Cholesky of Theta with or without look-ahead, (2) the exchange of the column shards of S as ONE padded
all-gather or as one exact-size broadcast per shard, (3) the Cholesky of the bordered Gauss-Newton matrix Hb replicated or
panel-sharded, (4) how many ranks.  bench.py A/B-times (1)-(3) on the fabric; this module says, BEFORE that, what each should take
from (a) the 1-GPU phase times measured in the same job (`one_gpu`: defaults = the round-5 driver run on one MI355X) and (b) the
bandwidth of the bound collectives (`fabric`: the preflight's GB/s when there is one, else one xGMI link at 153 GB/s x 0.8) -- and
bench.py prints `expected_ms` beside every `mode_probe` entry and `predicted_vs_1gpu` beside `vs_1gpu`.

Model (all times ms; K = ceil(N / nb) panels, panel k has m_k = N - k nb rows):
  panel factor      t_pf(k)  = chain_us * nb / 64 + m_k nb^2 / R_panel          (owner only: 64-column chain + tall-panel product)
  broadcast         t_bc(k)  = 8 m_k nb / B_bcast + lat                          (ring / tree broadcast runs at about one link)
  trailing update   t_up(k)  = (m_k - nb)^2 nb / R_gemm x imbalance(k, P) / P    (block columns dealt cyclically)
  sequential        sum_k t_pf + t_bc + t_up
  look-ahead        t_pf(0) + t_bc(0) + sum_k max(t_up(k), t_pf(k+1) + t_bc(k+1))
  step              solve / P + exchange(S) + product x imbalance / P + all-gather(Hb lower parts) + Cholesky(Hb) + tail
  exchange(S)       all-gather: (P-1) x widest shard / B_allgather ; broadcasts: all shards, one after the other, / B_bcast ;
                    direct (grouped send / recv, round 6): widest shard / B_link -- the P - 1 incoming shards use P - 1 links at once
                    (the broadcast form overlaps the block-row products with the arrivals: the model takes max(exchange, product) + the
                    last shard's share instead of the sum)
R_panel, R_gemm are fitted so that P = 1 reproduces the measured 1-GPU factorisation.  Pure Python + the column-shard rule of
gpk_mg_column_bounds restated (tests/test_bench_flow.py checks the restatement against gpk.mg.column_bounds).
"""
import math

LINK_GBS = 153.0                      # one xGMI link, per direction (MI355X_MICROARCH.md)
DEFAULT_FABRIC = {'bcast_gbs': 0.8 * LINK_GBS, 'allgather_gbs': 0.8 * LINK_GBS, 'p2p_link_gbs': 0.8 * LINK_GBS, 'links': 7, 'latency_ms': 0.03,
                  'source': 'assumed: one xGMI link x 0.8'}
# round-5 driver run, config 5 on ONE MI355X (BENCH_r05 / bench_detail.json: sharded_config): replaced by the same job's own 1-GPU point
DEFAULT_ONE_GPU = {'cholesky_theta_ms': 245.0, 'step_ms': 299.0, 'cholesky_hb_ms': 37.0, 'tail_ms': 3.0,
                   'solve_flops': 1.1212e13, 'product_flops': 6.009e12, 'source': 'round-5 driver run on one MI355X'}
CHAIN_US_PER_64 = 24.0                # dependent 64-column chain of the panel kernels (profiles/r05_bench_kernel_stats.csv)


def column_bounds(ncols, lead, rows, world, align=128):
    """work-balanced column shards of the leading-zero layout (csrc/gpk_mg.hip column_bounds, gpk/sharded.py column_ranges_lz)"""
    acc, w = 0.0, []
    for c in range(ncols):
        ln = float(rows - max(0, lead - 1 - c))
        acc += ln * ln
        w.append(acc)
    b = [0]
    for r in range(1, world):
        cut = 0
        if ncols > 0:
            target = w[-1] * r / world
            lo, hi = 0, ncols
            while lo < hi:                                          # lower_bound
                mid = (lo + hi) // 2
                if w[mid] < target:
                    lo = mid + 1
                else:
                    hi = mid
            cut = lo
        cut = -(-cut // align) * align
        b.append(min(max(cut, b[-1]), ncols))
    b.append(ncols)
    return b


def _potrf(n, nb, P, fabric, r_gemm, r_panel, lookahead):
    K = -(-n // nb)
    t_pf, t_bc, t_up = [], [], []
    for k in range(K):
        m = n - k * nb
        w = min(nb, m)
        t_pf.append(CHAIN_US_PER_64 * 1e-3 * w / 64.0 + 1e3 * m * w * w / r_panel)
        t_bc.append(0.0 if P == 1 else 8.0 * m * w / (fabric['bcast_gbs'] * 1e6) + fabric['latency_ms'])
        rest = max(m - w, 0)
        cols = -(-rest // nb)                                       # trailing block columns, dealt cyclically: the busiest rank has ceil(cols / P)
        imb = (-(-cols // P) * P / cols) if cols else 1.0
        t_up.append(1e3 * rest * rest * w / r_gemm * imb / P)
    if not lookahead or P == 1:
        return sum(t_pf) + sum(t_bc) + sum(t_up)
    t = t_pf[0] + t_bc[0]
    for k in range(K):
        nxt = (t_pf[k + 1] + t_bc[k + 1]) if k + 1 < K else 0.0
        t += max(t_up[k], nxt)
    return t


def _fit_rates(n, nb, target_ms, panel_share=0.25):
    """R_gemm, R_panel [flop/s] such that the P = 1 model gives target_ms with `panel_share` of it in the panel factorisations"""
    K = -(-n // nb)
    chain = sum(CHAIN_US_PER_64 * 1e-3 * min(nb, n - k * nb) / 64.0 for k in range(K))
    pf = sum((n - k * nb) * min(nb, n - k * nb) ** 2 for k in range(K))
    up = sum(max(n - k * nb - min(nb, n - k * nb), 0) ** 2 * min(nb, n - k * nb) for k in range(K))
    t_panel = max(panel_share * target_ms - chain, 1e-3)
    t_update = max(target_ms - chain - t_panel, 1e-3)
    return 1e3 * up / t_update, 1e3 * pf / t_panel


def predict(N, nz, P, nb=512, one_gpu=None, fabric=None, col_align=128):
    """-> dict of predicted milliseconds for P ranks (see the module docstring)"""
    g = dict(DEFAULT_ONE_GPU, **(one_gpu or {}))
    f = dict(DEFAULT_FABRIC, **(fabric or {}))
    nc = nz + 1
    r_gemm, r_panel = _fit_rates(N, nb, g['cholesky_theta_ms'])
    chol = {'lookahead': _potrf(N, nb, P, f, r_gemm, r_panel, True), 'sequential': _potrf(N, nb, P, f, r_gemm, r_panel, False)}
    # the step: 1-GPU phase split by executed flops at one common rate
    body = g['step_ms'] - g['cholesky_hb_ms'] - g['tail_ms']
    solve1 = body * g['solve_flops'] / (g['solve_flops'] + g['product_flops'])
    prod1 = body - solve1
    b = column_bounds(nc, nz, N, P, col_align)
    widths = [b[r + 1] - b[r] for r in range(P)]
    total_bytes = 8.0 * N * nc
    widest = 8.0 * N * max(widths)
    solve = solve1 / P                                              # shards are cut by work
    nblk = -(-nc // nb)
    share = [0.0] * P
    for i in range(nblk):
        i0, ib = i * nb, min(nb, nc - i * nb)
        share[i % P] += ib * (i0 + ib) * float(N - max(0, nz - (i0 + ib)))
    prod = prod1 * max(share) / max(sum(share), 1.0)
    lat = f['latency_ms']
    if P == 1:
        ex = {'all_gather_padded': 0.0, 'broadcasts_exact': 0.0, 'direct_p2p': 0.0}
        hb_gather = hb_direct = 0.0
    else:
        # direct exchange (grouped send / recv): a rank receives P - 1 shards over min(P - 1, links) links at once; the widest shard bounds it
        rounds = -(-(P - 1) // max(int(f.get('links', 7)), 1))
        ex = {'all_gather_padded': (P - 1) * widest / (f['allgather_gbs'] * 1e6) + lat,
              'broadcasts_exact': total_bytes / (f['bcast_gbs'] * 1e6) + P * lat,
              'direct_p2p': rounds * widest / (f['p2p_link_gbs'] * 1e6) + lat}
        hshare = max(sum(min(nb, nc - i * nb) * (i * nb + min(nb, nc - i * nb)) for i in range(r, nblk, P)) for r in range(P))
        hb_gather = (P - 1) * 8.0 * hshare / (f['allgather_gbs'] * 1e6) + lat
        hb_direct = rounds * 8.0 * hshare / (f['p2p_link_gbs'] * 1e6) + lat
    rg_h, rp_h = _fit_rates(nc, nb, g['cholesky_hb_ms'])
    hb = {'replicated': g['cholesky_hb_ms'], 'panel_sharded': _potrf(nc, nb, P, f, rg_h, rp_h, True)}
    last = 8.0 * N * widths[-1] / (f['bcast_gbs'] * 1e6) if P > 1 else 0.0
    step_by_exchange = {'all_gather': solve + ex['all_gather_padded'] + prod,
                        'broadcasts_chased_by_products': solve + max(ex['broadcasts_exact'], prod) + min(last, prod) if P > 1 else solve + prod,
                        'direct_p2p': solve + ex['direct_p2p'] + prod}
    step = {xk: {hk: xv + (hb_direct if xk == 'direct_p2p' else hb_gather) + g['tail_ms'] + hv for hk, hv in hb.items()}
            for xk, xv in step_by_exchange.items()}
    best = min(v for d in step.values() for v in d.values())
    return {'ranks': P, 'shard_widths': widths, 'shard_max_over_mean': max(widths) * P / float(nc),
            'cholesky_theta_ms': chol, 'exchange_of_S_ms': ex, 'cholesky_of_Hb_ms': hb, 'all_gather_Hb_ms': hb_gather,
            'solve_ms': solve, 'product_ms': prod, 'step_ms': step, 'step_ms_best': best,
            'predicted_vs_1gpu': g['step_ms'] / best,
            'prefer': {'lookahead': chol['lookahead'] <= chol['sequential'],
                       'overlap_s': min(step['broadcasts_chased_by_products'].values()) < min(step['all_gather'].values()),
                       'exchange': min(step, key=lambda k: min(step[k].values())),
                       'shard_hb': hb['panel_sharded'] < hb['replicated']}}


def fabric_from_preflight(pre, nbytes=139 * 2 ** 20):
    """GB/s of the bound collectives from bench.py's preflight object (gpk_mg_preflight: ms per 139 MB broadcast per root, ms per all-gather
    of the same total); None -> the assumed link rate"""
    if not isinstance(pre, dict) or 'error' in pre:
        return None
    out = {}
    if pre.get('bcast_gbs_by_root'):
        vals = [v for v in pre['bcast_gbs_by_root'] if v and v > 0]
        if vals:
            out['bcast_gbs'] = min(vals)
    if pre.get('allgather_gbs_received'):                            # bytes received per rank / time: the unit the model divides by
        out['allgather_gbs'] = pre['allgather_gbs_received']
    if pre.get('direct_gbs_received') and pre.get('ranks_seen_by_rccl', 0) > 1:
        # the preflight's direct exchange received (P - 1) equal parts at once: per-link rate = total / (P - 1) when the links are independent
        out['p2p_link_gbs'] = pre['direct_gbs_received'] / (pre['ranks_seen_by_rccl'] - 1)
    if out:
        out['source'] = 'preflight of the bound collectives'
    return out or None


def table(N=34000, nz=16000, nb=512, one_gpu=None, fabric=None, ranks=(2, 4, 8)):
    """the object bench.py attaches under `predicted` (numbers rounded to 0.1 ms by the caller's compactor)"""
    g = dict(DEFAULT_ONE_GPU, **(one_gpu or {}))
    f = dict(DEFAULT_FABRIC, **(fabric or {}))
    out = {'inputs': {'one_gpu': {k: g[k] for k in ('cholesky_theta_ms', 'step_ms', 'cholesky_hb_ms', 'source')},
                      'fabric': {k: f[k] for k in ('bcast_gbs', 'allgather_gbs', 'p2p_link_gbs', 'source')}}}
    for P in ranks:
        p = predict(N, nz, P, nb, g, f)
        out[str(P)] = {'cholesky_theta_ms': p['cholesky_theta_ms'], 'exchange_of_S_ms': p['exchange_of_S_ms'],
                       'cholesky_of_Hb_ms': p['cholesky_of_Hb_ms'], 'step_ms_best': p['step_ms_best'],
                       'predicted_vs_1gpu': p['predicted_vs_1gpu'], 'shard_max_over_mean': p['shard_max_over_mean'], 'prefer': p['prefer']}
    return out


if __name__ == '__main__':
    import json
    print(json.dumps(table(), indent=1))
