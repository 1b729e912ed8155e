// gpk_gn.hip -- device-resident Gauss-Newton step.
//
// Reference: *.GN_method / Hessian_GN / grad_loss / loss of src/PDEs.py:82-135,278-343,418-498,137-201 and
// src/InverseProblems.py:105-186.  There, per step: H = hessian(GN_loss) by forward-over-reverse autodiff through a
// general LU solve of the triangular L, g = grad(loss) through another LU solve, delta = LU-solve(H, g), and a third
// LU solve for the new loss.  Here:
//     S  = [ L^{-1} A(z) | L^{-1} F(z) ]          one blocked TRSM, nz+1 right-hand sides       (N^2 nz flops)
//     Hb = S^T S = [[H/2, g/2], [g^T/2, loss]]    one SYRK on the MFMA units                    (N nz^2 flops)
//     chol(Hb) = [[L_H, 0], [y^T, .]], y = L_H^{-1} g/2: the forward solve of POTRS comes free   (nz^3/3 flops)
//     delta = L_H^{-T} y ;  z <- z - step * delta
// A(z) is diagonal/identity/zero blocks: it is written straight into the S buffer, never built densely
// on the host.  Several row groups with their own factor are stacked (Darcy: L_a, L_u, and the data misfit as rows
// with identity factor; relaxed elliptic: the penalty rows).
#include "gpk_common.h"

namespace {

struct Group { const double* L; int ldl; int n; int off; const double* Dinv; };

struct Dims { int nz; int rows; int ngroups; Group g[3]; };

int gn_dims(const gpk_gn_problem* p, Dims& d) {
    const int Nd = p->Nd, Nb = p->Nb;
    if (Nd <= 0 || Nb < 0) return GPK_ERR_ARG;
    switch (p->system) {
        case GPK_GN_ELLIPTIC:
            d.nz = Nd; d.ngroups = 1; d.g[0] = {p->L, p->ldl, 2 * Nd + Nb, 0, p->Dinv}; d.rows = 2 * Nd + Nb; break;
        case GPK_GN_BURGERS:
        case GPK_GN_EIKONAL:
            d.nz = 3 * Nd; d.ngroups = 1; d.g[0] = {p->L, p->ldl, 4 * Nd + Nb, 0, p->Dinv}; d.rows = 4 * Nd + Nb; break;
        case GPK_GN_DARCY:
            if (p->Ndata < 0 || p->Ndata > Nd) return GPK_ERR_ARG;
            d.nz = 6 * Nd; d.ngroups = 3;
            d.g[0] = {p->L2, p->ldl2, 3 * Nd, 0, p->Dinv2};
            d.g[1] = {p->L, p->ldl, 4 * Nd + Nb, 3 * Nd, p->Dinv};
            d.g[2] = {nullptr, 0, p->Ndata, 7 * Nd + Nb, nullptr};
            d.rows = 7 * Nd + Nb + p->Ndata; break;
        case GPK_GN_ELLIPTIC_RELAXED:
            d.nz = 2 * Nd; d.ngroups = 2;
            d.g[0] = {p->L, p->ldl, 2 * Nd + Nb, 0, p->Dinv};
            d.g[1] = {nullptr, 0, Nd, 2 * Nd + Nb, nullptr};
            d.rows = 3 * Nd + Nb; break;
        default: return GPK_ERR_ARG;
    }
    return 0;
}

struct BuildArgs {
    int system, Nd, Nb, Ndata;
    double p0, p1, lam;
    const double* f; const double* gb; const double* data; const double* z;
    double* S; long lds; int fcol; int write_A;
    int rev;           // leading-zero layout of gn_step: unknown j is stored in column nz-1-pos(j), and that column of A(z) is zero above
                       // row pos(j).  1: pos(j) = j (elliptic systems).  2: Eikonal, unknown groups [v0 | v1 | v2] taken in the
                       // order v1, v2, v0: their first non-zeros sit in rows t, N_d + t, 3 N_d + t >= pos.  3: Burgers, the three
                       // unknowns of point t (columns t, N_d + t, 2 N_d + t all start in row t) interleaved: pos(j) = 3t + group,
                       // a staircase of slope 1/3 (column zero above row pos/3; gpk_ctx::lead_div = 3).  4: Darcy (round 4), unknown groups
                       // [w0 | w1 | w2 | v0 | v1 | v2] taken in the order v1, v2, w1, w2, w0, v0 -- see darcy_profiles below
    int nz;
    int family;        // elliptic system, gpk_gn_structured_prepare: 1 = only the unit entries of the first row group ([I; 0; 0], no F),
                       // 2 = only those of the second ([0; I; 0]) together with F(z); 0 = the normal [A(z) | F(z)]
};

// position of unknown j in the staircase order (see BuildArgs::rev)
__host__ __device__ __forceinline__ int stair_pos(int rev, int Nd, int j) {
    if (rev == 4) {                                                   // natural group (w0 w1 w2 v0 v1 v2) -> block of the staircase order
        const int blk = j / Nd;
        const int to = blk == 0 ? 4 : blk == 1 ? 2 : blk == 2 ? 3 : blk == 3 ? 5 : blk == 4 ? 0 : 1;
        return to * Nd + j % Nd;
    }
    return rev == 2 ? ((j / Nd + 2) % 3) * Nd + j % Nd : rev == 3 ? 3 * (j % Nd) + j / Nd : j;
}

// Darcy system, leading-zero layout (round 4).  Column c of S holds the unknown at staircase position p = n_z - 1 - c, blocks
// [v1 | v2 | w1 | w2 | w0 | v0] of N_d positions each.  First non-zero rows of A(z) (src/InverseProblems.py:127-143 of the reference):
//   u-part (rows [v1; v2; v3; v0; g], factor L_u):  v1_t: t,  v2_t: N_d + t,  w0_t, w1_t, w2_t: 2 N_d + t (the v3 row),  v0_t: 3 N_d + t
//   a-part (rows [w1; w2; w0], factor L_a):         w1_t: t,  w2_t: N_d + t,  w0_t: 2 N_d + t;  the v columns are zero there
// In column order that is, for the u-part, slope 1 over the v0 and w0 columns (c < 2 N_d: row 4 N_d - 1 - c), a flat step at 2 N_d over
// the w2, w1 columns, slope 1 again over v2, v1 (row 6 N_d - 1 - c) -- three segments of a GpkStair -- and for the a-part ONE slope-1
// staircase over the contiguous columns [N_d, 4 N_d) (row 4 N_d - 1 - c: the closed form with lead = 3 N_d on that sub-range), all
// other columns zero.  Executed flops of the solve: 27 % of the dense count; of the product: 38 %.
// Eikonal system (round 4): unknown groups in the order v1, v2, v0, i.e. columns [v0 | v2 v1] in memory.  First non-zero rows of A(z)
// (src/PDEs.py:441-449 of the reference): v1_t: t, v2_t: N_d + t, v0_t: 3 N_d + t -- slope 1 over the v2, v1 columns (row 3 N_d - 1 - c)
// and slope 1 again, N_d rows LATER, over the v0 columns (row 4 N_d - 1 - c).  Until round 3 the closed form treated the v0 columns as if
// they started at row 2 N_d + t (conservative: N_d rows of zeros were multiplied through); the two-segment profile is exact.
inline GpkStair eikonal_profile(int Nd) {
    GpkStair st;
    st.nseg = 2;
    st.c1[0] = Nd;     st.a[0] = 0; st.b[0] = 4 * Nd - 1; st.sd[0] = 1;
    st.c1[1] = 3 * Nd; st.a[1] = 0; st.b[1] = 3 * Nd - 1; st.sd[1] = 1;
    return st;
}

inline GpkStair darcy_u_profile(int Nd) {
    GpkStair st;
    st.nseg = 3;
    st.c1[0] = 2 * Nd; st.a[0] = 0;      st.b[0] = 4 * Nd - 1; st.sd[0] = 1;
    st.c1[1] = 4 * Nd; st.a[1] = 2 * Nd; st.b[1] = 4 * Nd;     st.sd[1] = 1 << 30;      // flat
    st.c1[2] = 6 * Nd; st.a[2] = 0;      st.b[2] = 6 * Nd - 1; st.sd[2] = 1;
    return st;
}

__device__ __forceinline__ void putA(const BuildArgs& a, int r, int c, double v) {
    if (a.write_A) a.S[(long)r * a.lds + (a.rev ? a.nz - 1 - stair_pos(a.rev, a.Nd, c) : c)] = v;
}
__device__ __forceinline__ void putF(const BuildArgs& a, int r, double v) { a.S[(long)r * a.lds + a.fcol] = v; }

// one thread per collocation index: writes the few non-zeros of A(z) and the entries of F(z) it owns
__global__ __launch_bounds__(256) void gn_build_kernel(BuildArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int Nd = a.Nd, Nb = a.Nb;
    const double* z = a.z;
    if (a.system == GPK_GN_ELLIPTIC) {                      // src/PDEs.py:84-85 (F), :95-96 (A)
        const double alpha = a.p0, m = a.p1;
        if (a.family == 1) {
            if (t < Nd) putA(a, t, t, 1.0);
        } else if (t < Nd) {
            const double zi = z[t];
            if (a.family == 0) putA(a, t, t, alpha * m * pow(zi, m - 1.0));
            putA(a, Nd + t, t, 1.0);
            putF(a, t, alpha * pow(zi, m) - a.f[t]);
            putF(a, Nd + t, zi);
        } else if (t < Nd + Nb) putF(a, 2 * Nd + (t - Nd), a.gb[t - Nd]);
    } else if (a.family != 0) {
        // gpk_gn_structured_prepare, systems other than the elliptic one (round 6).  Every column of A(z) has at most ONE entry that
        // depends on z -- Burgers: the PDE row t (src/PDEs.py:297-299 of the reference); Eikonal: the row 2 N_d + t (:443-445); Darcy: the
        // v3 row of the u-part (src/InverseProblems.py:131-135) -- so A(z) = A_1 diag(d(z)) + A_2 with 0/1 patterns A_1 (family 1: those
        // entries as ones) and A_2 (family 2: the entries that are constants; Darcy's 1/gamma in the data rows included).  No F column.
        if (t >= Nd) return;
        if (a.system == GPK_GN_BURGERS) {
            if (a.family == 1) { putA(a, t, t, 1.0); putA(a, t, Nd + t, 1.0); putA(a, t, 2 * Nd + t, 1.0); }
            else { putA(a, Nd + t, Nd + t, 1.0); putA(a, 2 * Nd + t, 2 * Nd + t, 1.0); putA(a, 3 * Nd + t, t, 1.0); }
        } else if (a.system == GPK_GN_EIKONAL) {
            if (a.family == 1) { putA(a, 2 * Nd + t, Nd + t, 1.0); putA(a, 2 * Nd + t, 2 * Nd + t, 1.0); }
            else { putA(a, t, Nd + t, 1.0); putA(a, Nd + t, 2 * Nd + t, 1.0); putA(a, 3 * Nd + t, t, 1.0); }
        } else if (a.system == GPK_GN_DARCY) {
            const int U = 3 * Nd, D = 7 * Nd + Nb;
            if (a.family == 1) {
                putA(a, U + 2 * Nd + t, t, 1.0); putA(a, U + 2 * Nd + t, Nd + t, 1.0); putA(a, U + 2 * Nd + t, 2 * Nd + t, 1.0);
                putA(a, U + 2 * Nd + t, 4 * Nd + t, 1.0); putA(a, U + 2 * Nd + t, 5 * Nd + t, 1.0);
            } else {
                putA(a, t, Nd + t, 1.0); putA(a, Nd + t, 2 * Nd + t, 1.0); putA(a, 2 * Nd + t, t, 1.0);
                putA(a, U + t, 4 * Nd + t, 1.0); putA(a, U + Nd + t, 5 * Nd + t, 1.0); putA(a, U + 3 * Nd + t, 3 * Nd + t, 1.0);
                if (t < a.Ndata) putA(a, D + t, 3 * Nd + t, 1.0 / a.p0);
            }
        }
    } else if (a.system == GPK_GN_BURGERS) {                // src/PDEs.py:280-287 (F), :297-305 (A)
        const double alpha = a.p0, nu = a.p1;
        if (t < Nd) {
            const double v0 = z[t], v2 = z[Nd + t], v3 = z[2 * Nd + t];
            putA(a, t, t, -alpha * v2); putA(a, t, Nd + t, -alpha * v0); putA(a, t, 2 * Nd + t, nu);
            putA(a, Nd + t, Nd + t, 1.0); putA(a, 2 * Nd + t, 2 * Nd + t, 1.0); putA(a, 3 * Nd + t, t, 1.0);
            putF(a, t, nu * v3 + a.f[t] - alpha * v0 * v2);
            putF(a, Nd + t, v2); putF(a, 2 * Nd + t, v3); putF(a, 3 * Nd + t, v0);
        } else if (t < Nd + Nb) putF(a, 4 * Nd + (t - Nd), a.gb[t - Nd]);
    } else if (a.system == GPK_GN_EIKONAL) {                // src/PDEs.py:420-428 (F), :441-449 (A)
        const double eps = a.p0;
        if (t < Nd) {
            const double v0 = z[t], v1 = z[Nd + t], v2 = z[2 * Nd + t], ft = a.f[t];
            putA(a, t, Nd + t, 1.0); putA(a, Nd + t, 2 * Nd + t, 1.0);
            putA(a, 2 * Nd + t, Nd + t, 2.0 * v1 / eps); putA(a, 2 * Nd + t, 2 * Nd + t, 2.0 * v2 / eps);
            putA(a, 3 * Nd + t, t, 1.0);
            putF(a, t, v1); putF(a, Nd + t, v2);
            putF(a, 2 * Nd + t, -(ft * ft - v1 * v1 - v2 * v2) / eps);
            putF(a, 3 * Nd + t, v0);
        } else if (t < Nd + Nb) putF(a, 4 * Nd + (t - Nd), a.gb[t - Nd]);
    } else if (a.system == GPK_GN_DARCY) {                  // src/InverseProblems.py:106-117 (F), :127-143 (A)
        const double gam = a.p0;                            // noise_level
        const int U = 3 * Nd, D = 7 * Nd + Nb;
        if (t < Nd) {
            const double w0 = z[t], w1 = z[Nd + t], w2 = z[2 * Nd + t];
            const double v0 = z[3 * Nd + t], v1 = z[4 * Nd + t], v2 = z[5 * Nd + t];
            const double fe = a.f[t] * exp(-w0);
            putA(a, t, Nd + t, 1.0);          putF(a, t, w1);                 // a-part rows [w1; w2; w0]
            putA(a, Nd + t, 2 * Nd + t, 1.0); putF(a, Nd + t, w2);
            putA(a, 2 * Nd + t, t, 1.0);      putF(a, 2 * Nd + t, w0);
            putA(a, U + t, 4 * Nd + t, 1.0);          putF(a, U + t, v1);      // u-part rows [v1; v2; v3; v0; g]
            putA(a, U + Nd + t, 5 * Nd + t, 1.0);     putF(a, U + Nd + t, v2);
            putA(a, U + 2 * Nd + t, t, fe);
            putA(a, U + 2 * Nd + t, Nd + t, -v1);
            putA(a, U + 2 * Nd + t, 2 * Nd + t, -v2);
            putA(a, U + 2 * Nd + t, 4 * Nd + t, -w1);
            putA(a, U + 2 * Nd + t, 5 * Nd + t, -w2);
            putF(a, U + 2 * Nd + t, -v1 * w1 - v2 * w2 - fe);
            putA(a, U + 3 * Nd + t, 3 * Nd + t, 1.0); putF(a, U + 3 * Nd + t, v0);
            if (t < a.Ndata) {                                                  // (1/gamma^2) sum (v0 - data)^2 as rows
                putA(a, D + t, 3 * Nd + t, 1.0 / gam);
                putF(a, D + t, (v0 - a.data[t]) / gam);
            }
        } else if (t < Nd + Nb) putF(a, U + 4 * Nd + (t - Nd), a.gb[t - Nd]);
    } else if (a.system == GPK_GN_ELLIPTIC_RELAXED) {       // src/PDEs.py:138-147 (loss), :153-165 (GN_loss)
        const double alpha = a.p0, m = a.p1, rs = 1.0 / sqrt(a.lam);
        const int P = 2 * Nd + Nb;
        if (t < Nd) {
            const double v = z[t], w = z[Nd + t];
            putA(a, t, t, 1.0); putF(a, t, v);
            putA(a, Nd + t, Nd + t, 1.0); putF(a, Nd + t, w);
            putA(a, P + t, t, -rs); putA(a, P + t, Nd + t, alpha * m * pow(w, m - 1.0) * rs);
            putF(a, P + t, (-v + alpha * pow(w, m) - a.f[t]) * rs);
        } else if (t < Nd + Nb) putF(a, 2 * Nd + (t - Nd), a.gb[t - Nd]);
    }
}

__global__ void axpy_kernel(int n, double alpha, const double* __restrict__ x, double* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

// y[j] += alpha * x[n-1-pos(j)]  (back from the staircase order of gn_step to the natural order of the unknowns)
__global__ void axpy_rev_kernel(int n, double alpha, const double* __restrict__ x, double* __restrict__ y, int rev, int Nd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += alpha * x[n - 1 - stair_pos(rev, Nd, i)];
}

__global__ void reverse_copy_kernel(int n, const double* __restrict__ x, double* __restrict__ y, int rev, int Nd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[n - 1 - stair_pos(rev, Nd, i)];
}

__global__ void scale_kernel(int n, double alpha, double* __restrict__ x) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= alpha;
}

int build(gpk_handle h, const gpk_gn_problem* p, const double* z, double* S, long lds, int fcol, int write_A, int rev = 0, int family = 0) {
    BuildArgs a;
    a.rev = rev; a.nz = fcol; a.family = family;
    a.system = p->system; a.Nd = p->Nd; a.Nb = p->Nb; a.Ndata = p->Ndata;
    a.p0 = p->p0; a.p1 = p->p1; a.lam = p->pen_lambda;
    a.f = p->rhs_f; a.gb = p->bdy_g; a.data = p->data_u; a.z = z;
    a.S = S; a.lds = lds; a.fcol = fcol; a.write_A = write_A;
    gn_build_kernel<<<gpk_ceil_div(p->Nd + p->Nb, 256), 256, 0, h->stream>>>(a);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

int check_prob(gpk_handle h, const gpk_gn_problem* p, Dims& d) {
    if (!p) return gpk_bad_arg(h, "gn: null problem");
    if (gn_dims(p, d) != 0) return gpk_bad_arg(h, "gn: system id / sizes");
    if (!p->rhs_f || (p->Nb > 0 && !p->bdy_g) || !p->L) return gpk_bad_arg(h, "gn: null rhs_f/bdy_g/L");
    if (p->system == GPK_GN_DARCY && (!p->L2 || (p->Ndata > 0 && !p->data_u))) return gpk_bad_arg(h, "gn: Darcy needs L2 and data_u");
    return 0;
}


// loss(z) = sum_k || L_k^{-1} F_k(z) ||^2 (+ the rows without a factor) by true substitution, one vector, on h->stream: `work` (rows
// doubles) receives F(z) and then the solved vector, d_out (device) the scalar.  gpk_gn_loss; the exact in-step loss of gpk_gn_step.
int substitution_loss(gpk_handle h, const gpk_gn_problem* p, const Dims& d, const double* z, double* work, double* d_out) {
    GPK_HIP(h, hipMemsetAsync(work, 0, (size_t)d.rows * sizeof(double), h->stream));
    GPK_TRY(build(h, p, z, work, 1, 0, 0));
    for (int k = 0; k < d.ngroups; ++k)
        if (d.g[k].L) GPK_TRY(gpk_i_trsv(h, false, d.g[k].L, d.g[k].n, d.g[k].ldl, work + d.g[k].off));
    return gpk_i_dot(h, work, work, d.rows, d_out);
}

// the exact in-step loss of gpk_gn_step / gpk_mg_gn_step (gpk_tune key 52): the same on the handle's own scratch vector, -> d_scalars[8]
int loss_work_reserve(gpk_handle h, const Dims& d) {
    if (h->loss_work_cap >= (size_t)d.rows) return 0;
    if (h->d_loss_work) {
        GPK_HIP(h, hipStreamSynchronize(h->stream));
        if (h->pipe_g) GPK_HIP(h, hipStreamSynchronize(h->pipe_g));
        (void)hipFree(h->d_loss_work); h->d_loss_work = nullptr; h->loss_work_cap = 0;
    }
    GPK_HIP(h, hipMalloc((void**)&h->d_loss_work, (size_t)d.rows * sizeof(double)));
    h->loss_work_cap = (size_t)d.rows;
    return 0;
}

int exact_loss(gpk_handle h, const gpk_gn_problem* p, const Dims& d, const double* z) {
    GPK_TRY(loss_work_reserve(h, d));
    return substitution_loss(h, p, d, z, h->d_loss_work, h->d_scalars + 8);
}

// The same in two halves (gpk_tune key 52 = 1): F(z) is written at the START of the step on the main stream (z is overwritten at its end);
// the chain -- one single-vector solve per factor, then the dot product -- is issued LATE, on the GEMM partition's stream: behind that
// stream's last product of the pipelined phase, or (after_main != 0: no pipelined phase) behind an event of the main stream, so that it
// runs next to the last panel chain (other partition) and the backward solve of the tail (main stream), not next to GEMM launches.
int exact_loss_build(gpk_handle h, const gpk_gn_problem* p, const Dims& d, const double* z) {
    GPK_TRY(loss_work_reserve(h, d));
    for (int i = 0; i < 2; ++i) if (!h->ev_loss[i]) GPK_HIP(h, hipEventCreateWithFlags(&h->ev_loss[i], hipEventDisableTiming));
    GPK_HIP(h, hipMemsetAsync(h->d_loss_work, 0, (size_t)d.rows * sizeof(double), h->stream));
    GPK_TRY(build(h, p, z, h->d_loss_work, 1, 0, 0));
    GPK_HIP(h, hipEventRecord(h->ev_loss[0], h->stream));            // F(z) is written
    return 0;
}

// after_main: the phase before the tail ran on the main stream (no two-partition pipeline): the chain waits for it; otherwise it only waits
// for F(z) and queues behind the GEMM partition's own last launch of the pipelined phase
int exact_loss_chain(gpk_handle h, const Dims& d, bool after_main, bool* on_side) {
    const hipStream_t main_s = h->stream, side = h->pipe_g;
    *on_side = false;
    if (!side || h->pipe_unavailable) {
        // the side stream went away between the decision at the start of the step and here (the pipelined phase re-created its streams and
        // failed: gpk_i_syrk_potrf, gpk_tune key 13): F(z) is in d_loss_work already, finish on the main stream -- never on the NULL stream
        for (int k = 0; k < d.ngroups; ++k)
            if (d.g[k].L) GPK_TRY(gpk_i_trsv(h, false, d.g[k].L, d.g[k].n, d.g[k].ldl, h->d_loss_work + d.g[k].off));
        return gpk_i_dot(h, h->d_loss_work, h->d_loss_work, d.rows, h->d_scalars + 8);
    }
    *on_side = true;
    if (after_main) GPK_HIP(h, hipEventRecord(h->ev_loss[0], main_s));
    GPK_HIP(h, hipStreamWaitEvent(side, h->ev_loss[0], 0));
    h->stream = side; h->trsv_alt = 1;
    int rc = 0;
    for (int k = 0; k < d.ngroups && rc == 0; ++k)
        if (d.g[k].L) rc = gpk_i_trsv(h, false, d.g[k].L, d.g[k].n, d.g[k].ldl, h->d_loss_work + d.g[k].off);
    if (rc == 0) rc = gpk_i_dot(h, h->d_loss_work, h->d_loss_work, d.rows, h->d_scalars + 8);
    h->stream = main_s; h->trsv_alt = 0;
    if (rc) { (void)hipStreamSynchronize(side); return rc; }
    GPK_HIP(h, hipEventRecord(h->ev_loss[1], side));
    return 0;
}

// The column layout gpk_gn_step runs a system in (BuildArgs::rev): 1 elliptic systems, 2 Eikonal, 3 Burgers, 4 Darcy on the GEMM-only
// solve path, 0 = dense schedule.
int step_layout(gpk_handle h, const gpk_gn_problem* p) {
    const bool darcy_lz = p->system == GPK_GN_DARCY && h->tune.eikonal_lz && h->tune.use_dinv && p->Dinv && p->Dinv2 && p->dinv_block > 0;
    return (p->system == GPK_GN_ELLIPTIC || p->system == GPK_GN_ELLIPTIC_RELAXED) ? 1
         : (p->system == GPK_GN_EIKONAL && h->tune.eikonal_lz) ? 2 : (p->system == GPK_GN_BURGERS && h->tune.eikonal_lz) ? 3 : darcy_lz ? 4 : 0;
}

// The handle state that goes with it for the duration of one call: the Eikonal profile (GEMM-only solve path: the exact two-segment
// profile for the solve and the products; the substitution path keeps the conservative closed form, which gpk_i_trsm_left_lz understands),
// the staircase slope of the Burgers system; everything reset on the way out, whichever way that is.
struct LayoutScope {
    gpk_handle h;
    LayoutScope(gpk_handle hh, const gpk_gn_problem* p, int rev) : h(hh) { gpk_i_gn_layout_enter(h, p, rev); }
    ~LayoutScope() { gpk_i_gn_layout_leave(h); }
    LayoutScope(const LayoutScope&) = delete;
};

// every factored row group comes with its inverted diagonal blocks (the GEMM-only solve path is available)
bool all_dinv(gpk_handle h, const gpk_gn_problem* p, const Dims& d) {
    if (!h->tune.use_dinv || p->dinv_block <= 0) return false;
    for (int k = 0; k < d.ngroups; ++k) if (d.g[k].L && !d.g[k].Dinv) return false;
    return true;
}

#define GPK_PROF_MARK(h, i) do { if ((h)->prof) { (h)->prof_phase = (i); GPK_HIP((h), hipEventRecord((h)->pev[i], (h)->stream)); } } while (0)

// S <- [L^{-1}A | L^{-1}F], Hb <- alpha * S^T S (lower triangle, bordered).
// rev != 0 (elliptic system only): unknown j is stored in column nz-1-j; column c < nz of [A | F] is then zero above row
// nz-1-c, which the solve and the SYRK exploit (about a third of the TRSM flops and a quarter of the SYRK flops).
// With the inverses of the diagonal blocks of every factor at hand (gpk_trtri_diag, gpk_gn_problem::Dinv) the solve runs
// out of place into the handle's workspace W (all-GEMM, see gpk_i_trsm_left_dinv) and the SYRK reads W; S is scratch then.
int assemble_normal_equations(gpk_handle h, const gpk_gn_problem* p, const Dims& d, const double* z, double* S, int lds,
                              double* Hb, int ldh, double alpha, int rev, double** Wout = nullptr, int family = 0) {
    const int nc = d.nz + 1;
    if (lds < nc || ldh < nc) return gpk_bad_arg(h, "gn: lds/ldh < nz+1");
    bool dinv = h->tune.use_dinv != 0;
    const int db = p->dinv_block > 0 ? p->dinv_block : 256;
    if (db != 256 && db != 512 && db != 1024 && db != 2048) return gpk_bad_arg(h, "gn: dinv_block must be 256, 512, 1024 or 2048");
    for (int k = 0; k < d.ngroups; ++k) if (d.g[k].L && !d.g[k].Dinv) dinv = false;
    double* W = S;
    if (dinv) {
        GPK_TRY(gpk_i_workspace(h, (size_t)d.rows * lds * sizeof(double), &W));
        // the solve never writes W left of the leading-zero boundary (rev) -- zero it unless the previous solve into W had
        // exactly this shape, in which case those zeros are still there
        // (the signature carries the inverted-block size -- the never-written region depends on it -- and is dropped again if the
        // solve below fails to be issued, so that a later step never trusts zeros this one did not leave behind)
        const long sig[5] = {d.rows, lds, nc, rev ? d.nz : 0, (long)(p->system + 1) * 4096 + db};
        if (memcmp(sig, h->work_sig, sizeof sig) != 0) {
            h->work_sig[0] = -1;
            GPK_HIP(h, hipMemsetAsync(W, 0, (size_t)d.rows * lds * sizeof(double), h->stream));
            memcpy(h->work_sig, sig, sizeof sig);
        }
    }
    struct SigGuard {                                                // any early return below invalidates the cached zero region
        gpk_handle h; bool armed; ~SigGuard() { if (armed) h->work_sig[0] = -1; }
    } sig_guard{h, dinv};
    GPK_PROF_MARK(h, 0);
    // (Darcy with the cached a-part: of the a-part rows of S only the F column is read -- build writes it for every row -- so those
    // 3 N_d rows, a third of the buffer, need not be cleared; the A entries build drops there are never looked at)
    const long skip = (rev == 4 && dinv && family == 0 && p->Wa && p->Ha) ? d.g[0].n : 0;
    GPK_HIP(h, hipMemsetAsync(S + skip * lds, 0, (size_t)(d.rows - skip) * lds * sizeof(double), h->stream));
    GPK_TRY(build(h, p, z, S, lds, d.nz, 1, rev, family));
    for (int k = 0; k < d.ngroups; ++k) {
        const Group& g = d.g[k];
        double* Sg = S + (long)g.off * lds;
        if (!g.L) {                                                  // identity factor (data misfit / penalty rows)
            if (dinv && g.n > 0)
                GPK_HIP(h, hipMemcpy2DAsync(W + (long)g.off * lds, (size_t)lds * 8, Sg, (size_t)lds * 8, (size_t)nc * 8, g.n,
                                            hipMemcpyDeviceToDevice, h->stream));
        } else if (dinv && rev == 4) {
            double* Wg = W + (long)g.off * lds;
            const int Nd = p->Nd;
            if (k == 0) {
                // a-part: only the columns [N_d, 4 N_d) are non-zero, a slope-1 staircase of their own (closed form on the sub-range),
                // and the dense F column as a one-column solve; the v columns of these rows stay zero in W.
                // The 3 N_d columns do not depend on z (a permutation matrix against a fixed factor): with the result of
                // gpk_gn_darcy_prepare at hand (p->Wa) that solve is not repeated -- the product below reads p->Wa / p->Ha instead
                if (!(p->Wa && p->Ha))
                    GPK_TRY(gpk_i_trsm_left_dinv(h, g.L, g.Dinv, db, g.n, g.ldl, Sg + Nd, lds, Wg + Nd, lds, 3 * Nd, 3 * Nd, 0));
                GPK_TRY(gpk_i_trsm_left_dinv(h, g.L, g.Dinv, db, g.n, g.ldl, Sg + d.nz, lds, Wg + d.nz, lds, 1, 0, 0));
            } else {
                h->stair = darcy_u_profile(Nd);                      // u-part: three segments (reset by the caller's guard)
                const int rc = gpk_i_trsm_left_dinv(h, g.L, g.Dinv, db, g.n, g.ldl, Sg, lds, Wg, lds, nc, 1, 0);
                h->stair = GpkStair(); h->stair_col0 = h->stair_row0 = 0;
                GPK_TRY(rc);
            }
        } else if (dinv) {
            GPK_TRY(gpk_i_trsm_left_dinv(h, g.L, g.Dinv, db, g.n, g.ldl, Sg, lds, W + (long)g.off * lds, lds, nc, rev ? d.nz : 0, 0));
        } else if (rev) {
            GPK_TRY(gpk_i_trsm_left_lz(h, g.L, g.n, g.ldl, Sg, nc, lds, d.nz, 0));
        } else {
            GPK_TRY(gpk_i_trsm_left_mt(h, false, g.L, g.n, g.ldl, Sg, nc, lds));
        }
    }
    GPK_PROF_MARK(h, 1);
    sig_guard.armed = false;                                         // the solve was issued completely
    if (Wout) { *Wout = W; return 0; }                               // gn_step: product and factorisation are pipelined by the caller
    GPK_TRY(gpk_i_gemm(h, true, false, nc, nc, d.rows, alpha, W, lds, W, lds, 0.0, Hb, ldh, true, rev ? d.nz : 0));
    GPK_PROF_MARK(h, 2);
    return 0;
}

// ---- structured solve of the elliptic system (optional, gpk_gn_structured_prepare) -------------------------------------------
// per-column coefficients in the internal (reversed) column order: column c holds unknown u = nz-1-c
__global__ void structured_coeff_kernel(int nz, double alpha, double m, const double* __restrict__ z, double* __restrict__ dcol,
                                        double* __restrict__ acol, double* __restrict__ zcol) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nz) return;
    const double zi = z[nz - 1 - c];
    dcol[c] = alpha * m * pow(zi, m - 1.0);
    acol[c] = alpha * pow(zi, m);
    zcol[c] = zi;
}

// one workgroup per row r: S[r][c] = d[c] W1[r][c] + W2[r][c] (c < nz), S[r][nz] = v0[r] + sum_c (a[c] W1[r][c] + z[c] W2[r][c]).
// Columns left of the leading-zero boundary (c < nz-1-r) are zero in W1, W2 and are written as zeros.
__global__ __launch_bounds__(256) void structured_form_kernel(int nz, const double* __restrict__ W1, const double* __restrict__ W2, long ldw,
                                                              const double* __restrict__ v0, const double* __restrict__ dcol,
                                                              const double* __restrict__ acol, const double* __restrict__ zcol,
                                                              double* __restrict__ S, long lds) {
    __shared__ double red[4];
    const long r = blockIdx.x;
    const double* w1 = W1 + r * ldw;
    const double* w2 = W2 + r * ldw;
    double* s = S + r * lds;
    const int cz = (int)max(0L, (long)nz - 1 - r) & ~1;             // first column that can be non-zero (even: 16-byte pairs)
    double acc = 0.0;
    for (int c = 2 * (int)threadIdx.x; c < nz; c += 512) {
        if (c + 1 < cz) { s[c] = 0.0; s[c + 1] = 0.0; continue; }
        const double a0 = w1[c], b0 = w2[c];
        s[c] = fma(dcol[c], a0, b0);
        acc = fma(acol[c], a0, fma(zcol[c], b0, acc));
        if (c + 1 < nz) {
            const double a1 = w1[c + 1], b1 = w2[c + 1];
            s[c + 1] = fma(dcol[c + 1], a1, b1);
            acc = fma(acol[c + 1], a1, fma(zcol[c + 1], b1, acc));
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) s[nz] = v0[r] + ((red[0] + red[1]) + (red[2] + red[3]));
}

// w[r] = v0[r] + sum_c (a[c] W1[r][c] + z[c] W2[r][c]) = (L^{-1}F(z))[r]: the Gram level's loss ||w||^2 is taken from w itself -- as
// the quadratic form v0^T v0 + ... of terms of size 1e16 it keeps three digits near convergence (loss 1e4)
__global__ __launch_bounds__(256) void structured_w_kernel(int nz, const double* __restrict__ W1, const double* __restrict__ W2, long ldw,
                                                           const double* __restrict__ v0, const double* __restrict__ acol,
                                                           const double* __restrict__ zcol, double* __restrict__ w) {
    __shared__ double red[4];
    const long r = blockIdx.x;
    const double* w1 = W1 + r * ldw;
    const double* w2 = W2 + r * ldw;
    const int cz = (int)max(0L, (long)nz - 1 - r);                   // W1, W2 are zero left of this column
    double acc = 0.0;
    for (int c = cz + (int)threadIdx.x; c < nz; c += 256) acc = fma(acol[c], w1[c], fma(zcol[c], w2[c], acc));
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) w[r] = v0[r] + ((red[0] + red[1]) + (red[2] + red[3]));
}

// Structured solve of the other systems (round 6): d(z) per column, in the step's internal column order (column c holds the unknown at
// staircase position nz-1-c).  The z-dependent entry of unknown j's column (see gn_build_kernel, family 1):
//   Burgers  (src/PDEs.py:297-299):            v0_t: -alpha v2_t    v2_t: -alpha v0_t    v3_t: nu
//   Eikonal  (src/PDEs.py:443-445):            v0_t: 0              v1_t: 2 v1_t / eps   v2_t: 2 v2_t / eps
//   Darcy    (src/InverseProblems.py:131-135): w0_t: f_t exp(-w0_t) w1_t: -v1_t  w2_t: -v2_t  v0_t: 0  v1_t: -w1_t  v2_t: -w2_t
__global__ void structured_coeff_general_kernel(int system, int Nd, int nz, int rev, double p0, double p1, const double* __restrict__ f,
                                                const double* __restrict__ z, double* __restrict__ dcol) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nz) return;
    const int blk = j / Nd, t = j - blk * Nd;
    double v = 0.0;
    if (system == GPK_GN_BURGERS) v = blk == 0 ? -p0 * z[Nd + t] : blk == 1 ? -p0 * z[t] : p1;
    else if (system == GPK_GN_EIKONAL) v = blk == 0 ? 0.0 : 2.0 * z[j] / p0;
    else if (system == GPK_GN_DARCY)
        v = blk == 0 ? f[t] * exp(-z[t]) : blk == 1 ? -z[4 * Nd + t] : blk == 2 ? -z[5 * Nd + t] : blk == 3 ? 0.0 : blk == 4 ? -z[Nd + t] : -z[2 * Nd + t];
    dcol[nz - 1 - stair_pos(rev, Nd, j)] = v;
}

// one workgroup per row r: W[r][c] = d[c] W1[r][c] + W2[r][c] for c < nz (16-byte pairs: ldw, lds even; nz may be odd).  The F column is
// not touched (solved separately, one column, by the caller).  Above the staircase W1 and W2 hold zeros and zeros are written.
__global__ __launch_bounds__(256) void structured_form_general_kernel(int nz, const double* __restrict__ W1, const double* __restrict__ W2, long ldw,
                                                                      const double* __restrict__ dcol, double* __restrict__ W, long lds) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const long r = blockIdx.x;
    const double* w1 = W1 + r * ldw;
    const double* w2 = W2 + r * ldw;
    double* w = W + r * lds;
    for (int c = 2 * (int)threadIdx.x; c < nz; c += 512) {
        if (c + 1 < nz) {
            const d2 a = *reinterpret_cast<const d2*>(w1 + c), b = *reinterpret_cast<const d2*>(w2 + c);
            const d2 d = *reinterpret_cast<const d2*>(dcol + c);
            *reinterpret_cast<d2*>(w + c) = (d2){fma(d.x, a.x, b.x), fma(d.y, a.y, b.y)};
        } else w[c] = fma(dcol[c], w1[c], w2[c]);
    }
}

// C[i][j] += A[i][j] for j <= i (one workgroup per row): the cached a-part contribution of the Darcy step (gpk_gn_darcy_prepare)
__global__ __launch_bounds__(256) void add_lower_kernel(int n, const double* __restrict__ A, long lda, double* __restrict__ Cm, long ldc) {
    const long i = blockIdx.x;
    const double* a = A + i * lda;
    double* c = Cm + i * ldc;
    for (int j = threadIdx.x; j <= (int)i; j += 256) c[j] += a[j];
}

__global__ void sub_kernel(long n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}

// ---- second level: the bordered matrix straight from the Gram blocks of W (optional, gpk_gn_gram_prepare) -----------------------
// one workgroup per column index c (a row of the four blocks): q1[c] = G11[c,:] a + G12[c,:] z + p1[c], q2[c] = G21[c,:] a + G22[c,:] z + p2[c]
__global__ __launch_bounds__(256) void gram_gemv_kernel(int nz, const double* __restrict__ G, long ldg, const double* __restrict__ pvec,
                                                        const double* __restrict__ acol, const double* __restrict__ zcol,
                                                        double* __restrict__ q) {
    __shared__ double red[2][4];
    const long c = blockIdx.x;
    const double* g11 = G + c * ldg;
    const double* g12 = G + ((long)nz + c) * ldg;
    const double* g21 = G + (2L * nz + c) * ldg;
    const double* g22 = G + (3L * nz + c) * ldg;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < nz; k += 256) {
        const double a = acol[k], z = zcol[k];
        s1 = fma(g11[k], a, fma(g12[k], z, s1));
        s2 = fma(g21[k], a, fma(g22[k], z, s2));
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        q[c] = pvec[c] + ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        q[nz + c] = pvec[nz + c] + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

// Hb[c][c'] (c' <= c) = d_c G11 d_c' + d_c G12 + G21 d_c' + G22; border row Hb[nz][c] = d_c q1[c] + q2[c]; one workgroup per row
__global__ __launch_bounds__(256) void gram_form_kernel(int nz, const double* __restrict__ G, long ldg, const double* __restrict__ dcol,
                                                        const double* __restrict__ q, double* __restrict__ Hb, long ldh) {
    const long c = blockIdx.x;
    if (c == nz) {
        for (int k = threadIdx.x; k < nz; k += 256) Hb[(long)nz * ldh + k] = fma(dcol[k], q[k], q[nz + k]);
        return;
    }
    const double* g11 = G + c * ldg;
    const double* g12 = G + ((long)nz + c) * ldg;
    const double* g21 = G + (2L * nz + c) * ldg;
    const double* g22 = G + (3L * nz + c) * ldg;
    const double dc = dcol[c];
    double* hrow = Hb + c * ldh;
    for (int k = threadIdx.x; k <= c; k += 256) {
        const double dk = dcol[k];
        hrow[k] = fma(dc, fma(g11[k], dk, g12[k]), fma(g21[k], dk, g22[k]));
    }
}

// loss = pvec[2nz] + a.p1 + z.p2 + a.q1 + z.q2 -> Hb[nz][nz] and d_loss
__global__ __launch_bounds__(1024) void gram_loss_kernel(int nz, const double* __restrict__ pvec, const double* __restrict__ acol,
                                                         const double* __restrict__ zcol, const double* __restrict__ q,
                                                         double* __restrict__ corner, double* __restrict__ d_loss) {
    __shared__ double red[16];
    double s = 0.0;
    for (int k = threadIdx.x; k < nz; k += 1024) s += acol[k] * (pvec[k] + q[k]) + zcol[k] * (pvec[nz + k] + q[nz + k]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        t += pvec[2 * nz];
        *corner = t; *d_loss = t;
    }
}


}  // namespace


extern "C" int gpk_gn_structured_prepare(gpk_handle h, const gpk_gn_problem* p, double* S, int lds, double* W1, double* W2, double* v0, int ldw) {
    if (!h || !S || !W1 || !W2) return GPK_ERR_ARG;
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    const bool elliptic = p->system == GPK_GN_ELLIPTIC;
    if (p->system == GPK_GN_ELLIPTIC_RELAXED) return gpk_bad_arg(h, "structured solve: not for the relaxed system");
    if (elliptic && !v0) return GPK_ERR_ARG;
    const int nc = d.nz + 1;
    if (lds < nc || ldw < nc) return gpk_bad_arg(h, "structured solve: lds/ldw < nz+1");
    // the other systems (round 6) are prepared -- and later stepped -- in the leading-zero layout of their GEMM-only solve path
    const int rev = elliptic ? 1 : step_layout(h, p);
    if (!elliptic && (rev < 2 || !all_dinv(h, p, d)))
        return gpk_bad_arg(h, "structured solve: this system needs the inverted diagonal blocks of every factor (Dinv, dinv_block) and the leading-zero layout");
    if (!elliptic && (ldw & 1)) return gpk_bad_arg(h, "structured solve: ldw must be even");
    LayoutScope scope(h, p, elliptic ? 1 : rev);
    gpk_gn_problem q = *p;                                           // (the Darcy a-part is solved here like everything else: no cache)
    q.Wa = nullptr; q.Ha = nullptr;
    double* zero = nullptr;
    GPK_HIP(h, hipMalloc((void**)&zero, (size_t)d.nz * sizeof(double)));
    hipError_t e = hipMemsetAsync(zero, 0, (size_t)d.nz * sizeof(double), h->stream);
    int rc = e == hipSuccess ? 0 : gpk_fail(h, e, "hipMemsetAsync", __FILE__, __LINE__);
    // elliptic: [I; 0; 0] -> W1;  [0; I; 0] with F(0) -> W2, v0.  Others: the 0/1 pattern of the z-dependent entries -> W1, the constant
    // entries -> W2 (gn_build_kernel, family 1 / 2), all row groups, no F column
    for (int fam = 1; fam <= 2 && rc == 0; ++fam) {
        double* W = nullptr;
        rc = assemble_normal_equations(h, &q, d, zero, S, lds, nullptr, nc, 1.0, rev, &W, fam);   // (no product: Wout is set)
        if (rc) break;
        e = hipMemcpy2DAsync(fam == 1 ? W1 : W2, (size_t)ldw * 8, W, (size_t)lds * 8, (size_t)nc * 8, d.rows, hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess && fam == 2 && elliptic)
            e = hipMemcpy2DAsync(v0, 8, W + d.nz, (size_t)lds * 8, 8, d.rows, hipMemcpyDeviceToDevice, h->stream);
        if (e != hipSuccess) rc = gpk_fail(h, e, "hipMemcpy2DAsync", __FILE__, __LINE__);
    }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(zero);
    return rc;
}

extern "C" int gpk_gn_gram_prepare(gpk_handle h, const gpk_gn_problem* p, double* G, int ldg, double* pvec) {
    if (!h || !G || !pvec) return GPK_ERR_ARG;
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    const bool elliptic = p->system == GPK_GN_ELLIPTIC;
    if (p->system == GPK_GN_ELLIPTIC_RELAXED || !p->W1 || !p->W2 || (elliptic && !p->v0))
        return gpk_bad_arg(h, "gram_prepare: needs W1/W2 (and v0 for the elliptic system) of gpk_gn_structured_prepare; not for the relaxed system");
    const int nz = d.nz;
    if (ldg < nz || p->ldw < nz + 1) return gpk_bad_arg(h, "gram_prepare: ldg/ldw");
    const double* Wm[2] = {p->W1, p->W2};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)                                  // Gij = Wi^T Wj (full blocks: the gemv reads rows of all four)
            GPK_TRY(gpk_i_gemm(h, true, false, nz, nz, d.rows, 1.0, Wm[i], p->ldw, Wm[j], p->ldw, 0.0, G + (long)(2 * i + j) * nz * ldg, ldg, false));
    if (elliptic) {
        for (int i = 0; i < 2; ++i)                                  // p_i = Wi^T v0 (a one-column product)
            GPK_TRY(gpk_i_gemm(h, true, false, nz, 1, d.rows, 1.0, Wm[i], p->ldw, p->v0, 1, 0.0, pvec + (long)i * nz, 1, false));
        GPK_TRY(gpk_i_dot(h, p->v0, p->v0, d.rows, pvec + 2L * nz));
    } else {
        // the other systems (round 6) solve the column w = L^{-1}F(z) every step and take the border from it: pvec is not used
        GPK_HIP(h, hipMemsetAsync(pvec, 0, (2 * (size_t)nz + 1) * sizeof(double), h->stream));
    }
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int gpk_gn_darcy_prepare(gpk_handle h, const gpk_gn_problem* p, double* S, int lds, double* Wa, int ldwa, double* Ha, int ldha) {
    if (!h || !S || !Wa || !Ha) return GPK_ERR_ARG;
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    if (p->system != GPK_GN_DARCY) return gpk_bad_arg(h, "darcy_prepare: Darcy system only");
    if (!p->Dinv || !p->Dinv2 || p->dinv_block <= 0) return gpk_bad_arg(h, "darcy_prepare: needs Dinv, Dinv2 and dinv_block (the GEMM-only solve path)");
    const int Nd = p->Nd, na = 3 * Nd, nc = d.nz + 1, db = p->dinv_block;
    if (lds < nc || ldwa < na || ldha < na) return gpk_bad_arg(h, "darcy_prepare: lds/ldwa/ldha");
    if (db != 256 && db != 512 && db != 1024 && db != 2048) return gpk_bad_arg(h, "gn: dinv_block must be 256, 512, 1024 or 2048");
    const Group& ga = d.g[0];
    double* zero = nullptr;
    GPK_HIP(h, hipMalloc((void**)&zero, (size_t)d.nz * sizeof(double)));
    int rc = 0;
    hipError_t e = hipMemsetAsync(zero, 0, (size_t)d.nz * sizeof(double), h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(S, 0, (size_t)d.rows * lds * sizeof(double), h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(Wa, 0, (size_t)na * ldwa * sizeof(double), h->stream);   // (never written left of the staircase)
    if (e != hipSuccess) rc = gpk_fail(h, e, "hipMemsetAsync", __FILE__, __LINE__);
    // exactly the step's launches for this block (assemble_normal_equations, rev = 4, k = 0; the product of gpk_gn_step): same
    // kernels, same shapes, same tile configurations -- only the destinations differ, hence bit-identical results
    if (rc == 0) rc = build(h, p, zero, S, lds, d.nz, 1, 4, 0);
    if (rc == 0) rc = gpk_i_trsm_left_dinv(h, ga.L, ga.Dinv, db, ga.n, ga.ldl, S + (long)ga.off * lds + Nd, lds, Wa, ldwa, na, na, 0);
    if (rc == 0) rc = gpk_i_gemm(h, true, false, na, na, ga.n, 1.0, Wa, ldwa, Wa, ldwa, 0.0, Ha, ldha, true, na);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(zero);
    return rc;
}

extern "C" int gpk_gn_dims(const gpk_gn_problem* p, int* nz, int* s_rows) {
    Dims d;
    if (!p || gn_dims(p, d) != 0) return GPK_ERR_ARG;
    if (nz) *nz = d.nz;
    if (s_rows) *s_rows = d.rows;
    return 0;
}

extern "C" int gpk_gn_worksize(const gpk_gn_problem* p, int lds, int* host_lds, size_t* S_bytes, size_t* Hb_bytes, size_t* delta_bytes,
                               size_t* handle_bytes) {
    Dims d;
    if (!p || gn_dims(p, d) != 0) return GPK_ERR_ARG;
    const int nc = d.nz + 1;
    if (lds == 0) lds = ((nc + 15) / 16) * 16;
    if (lds < nc) return GPK_ERR_ARG;
    // as assemble_normal_equations decides it: the out-of-place GEMM-only solve runs whenever EVERY factor comes with its inverted diagonal
    // blocks (dinv_block = 0 means 256 there too); the figure assumes the default gpk_tune(10, 1) -- with the substitution schedule forced
    // the handle reserves nothing
    bool dinv = false;
    for (int k = 0; k < d.ngroups; ++k) if (d.g[k].L && d.g[k].Dinv) dinv = true;
    for (int k = 0; k < d.ngroups; ++k) if (d.g[k].L && !d.g[k].Dinv) dinv = false;
    if (host_lds) *host_lds = lds;
    if (S_bytes) *S_bytes = (size_t)d.rows * lds * sizeof(double);
    if (Hb_bytes) *Hb_bytes = (size_t)nc * lds * sizeof(double);
    if (delta_bytes) *delta_bytes = (size_t)d.nz * sizeof(double);
    if (handle_bytes) *handle_bytes = (dinv ? (size_t)d.rows * lds * sizeof(double) : 0) + (size_t)d.rows * sizeof(double);
    return 0;
}

extern "C" int gpk_gn_step(gpk_handle h, const gpk_gn_problem* p, double* z, double step_size, double* S, int lds,
                           double* Hb, int ldh, double* delta, double* host_loss_in, int* host_info) {
    if (!h || !z || !S || !Hb || !delta) return GPK_ERR_ARG;
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    const int nz = d.nz;
    // reversed column order (leading-zero layout): unknown u's column of A(z) is zero above row u for the elliptic system (rows t and
    // N_d + t) and for its relaxed form (unknowns [v; w]: rows t resp. N_d + t of the Theta block, penalty rows at the bottom)
    // Eikonal: the same after regrouping the unknowns (BuildArgs::rev = 2).  Burgers: the three columns of point t all start in row
    // t -- interleaved they form a staircase of slope 1/3 (rev = 3).  Darcy (two factors with different column supports) runs the
    // dense schedule.
    // Darcy (round 4): leading-zero layout with a piecewise profile per factor (darcy_u_profile), only on the GEMM-only solve path
    if (p->system == GPK_GN_DARCY && p->Wa && p->Ha && (p->ldwa < 3 * p->Nd || p->ldha < 3 * p->Nd))
        return gpk_bad_arg(h, "gn: ldwa/ldha < 3 Nd (gpk_gn_darcy_prepare)");
    const int rev = step_layout(h, p);
    LayoutScope layout_scope(h, p, rev);
    double* W = nullptr;                                             // the solved block [L^{-1}A | L^{-1}F] (S or the workspace)
    const bool gram = h->tune.structured && p->system == GPK_GN_ELLIPTIC && p->G && p->pvec && p->ldg >= nz;
    // the same level for the Burgers / Eikonal / Darcy systems (round 6): H/2 = D G11 D + D G12 + G21 D + G22 from the Gram blocks of
    // W1 = L^{-1}A1, W2 = L^{-1}A2; the border from the SOLVED column w = L^{-1}F(z): g/2 = D W1^T w + W2^T w, loss = w^T w
    const bool gram_general = h->tune.structured && rev >= 2 && p->G && p->W1 && p->W2 && p->ldg >= nz && p->ldw >= nz + 1 && all_dinv(h, p, d);
    double* d_loss = h->d_scalars;
    bool exact = false, exact_late = false;                          // (d_scalars[8]: the loss by substitution, exact_loss*)
    GPK_HIP(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
    if ((gram || (h->tune.structured && p->W1)) && (long)d.rows * lds < 5L * nz + d.rows)
        return gpk_bad_arg(h, "gn: S too small for the scratch vectors of the structured modes");
    if (gram) {
        // optional Gram level (gpk_gn_gram_prepare): the bordered matrix assembled in O(nz^2), no solve and no product this step
        GPK_PROF_MARK(h, 0);
        double* coef = S;                                            // d, a, z (column order), then q1, q2: 5 nz doubles of scratch
        structured_coeff_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, p->p0, p->p1, z, coef, coef + nz, coef + 2 * nz);
        gram_gemv_kernel<<<nz, 256, 0, h->stream>>>(nz, p->G, p->ldg, p->pvec, coef + nz, coef + 2 * nz, coef + 3 * nz);
        GPK_PROF_MARK(h, 1);
        gram_form_kernel<<<nz + 1, 256, 0, h->stream>>>(nz, p->G, p->ldg, coef, coef + 3 * nz, Hb, ldh);
        gram_loss_kernel<<<1, 1024, 0, h->stream>>>(nz, p->pvec, coef + nz, coef + 2 * nz, coef + 3 * nz, Hb + (long)nz * ldh + nz, d_loss);
        // (the reported loss from w = L^{-1}F(z) itself, see structured_w_kernel; w lives behind the five coefficient vectors in S)
        structured_w_kernel<<<d.rows, 256, 0, h->stream>>>(nz, p->W1, p->W2, p->ldw, p->v0, coef + nz, coef + 2 * nz, coef + 5 * nz);
        GPK_LAUNCH_CHECK(h);
        GPK_TRY(gpk_i_dot(h, coef + 5 * nz, coef + 5 * nz, d.rows, d_loss));
        h->pipe_tev_used = 0; h->prof_pipelined = 0;
        GPK_TRY(gpk_i_potrf(h, Hb, nz + 1, ldh, 0));
        // the REPORTED loss by true substitution here too (round 6; the class API takes its history from this number): one vector on
        // the main stream behind the factorisation -- the level's own d_loss (above) stays the corner of Hb
        if (h->tune.exact_loss) { exact = true; GPK_TRY(exact_loss(h, p, d, z)); }
    } else if (h->tune.structured && p->system == GPK_GN_ELLIPTIC && p->W1 && p->W2 && p->v0 && p->ldw >= nz + 1) {
        // optional structured solve (gpk_gn_structured_prepare): one memory-bound pass over W1, W2 instead of the triangular solve
        GPK_PROF_MARK(h, 0);
        double* coef = S;                                            // 3 nz doubles of scratch (S is free in this mode)
        GPK_TRY(gpk_i_workspace(h, (size_t)d.rows * lds * sizeof(double), &W));
        h->work_sig[0] = -1;                                         // (the workspace no longer holds a solve of a known shape)
        structured_coeff_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, p->p0, p->p1, z, coef, coef + nz, coef + 2 * nz);
        structured_form_kernel<<<d.rows, 256, 0, h->stream>>>(nz, p->W1, p->W2, p->ldw, p->v0, coef, coef + nz, coef + 2 * nz, W, lds);
        GPK_LAUNCH_CHECK(h);
        if (h->tune.exact_loss) { exact = true; GPK_TRY(exact_loss(h, p, d, z)); }   // reported loss: true substitution (round 6), as in the plain branch
        GPK_PROF_MARK(h, 1);
    } else if (gram_general) {
        if (h->tune.exact_loss) {
            exact = true;
            exact_late = h->tune.exact_loss == 1 && h->pipe_g && !h->pipe_unavailable;
            if (exact_late) GPK_TRY(exact_loss_build(h, p, d, z));
            else GPK_TRY(exact_loss(h, p, d, z));
        }
        GPK_PROF_MARK(h, 0);
        const int db = p->dinv_block;
        double* ws = nullptr;                                        // [w (rows) | d (nz) | q1 (nz) | q2 (nz)]
        GPK_TRY(gpk_i_workspace(h, ((size_t)d.rows + 3 * (size_t)nz + 16) * sizeof(double), &ws));
        h->work_sig[0] = -1;
        double* wv = ws;
        double* dcol = ws + ((d.rows + 1) & ~1L);                    // (16-byte aligned)
        double* q = dcol + nz;
        structured_coeff_general_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(p->system, p->Nd, nz, rev, p->p0, p->p1, p->rhs_f, z, dcol);
        GPK_LAUNCH_CHECK(h);
        GPK_TRY(build(h, p, z, S, lds, nz, 0));                      // F(z) into column nz of S; w = L^{-1}F(z), one column per factor
        for (int k = 0; k < d.ngroups; ++k) {
            const Group& g = d.g[k];
            if (g.n <= 0) continue;
            if (g.L) GPK_TRY(gpk_i_trsm_left_dinv(h, g.L, g.Dinv, db, g.n, g.ldl, S + (long)g.off * lds + nz, lds, wv + g.off, 1, 1, 0, 0));
            else GPK_HIP(h, hipMemcpy2DAsync(wv + g.off, 8, S + (long)g.off * lds + nz, (size_t)lds * 8, 8, g.n, hipMemcpyDeviceToDevice, h->stream));
        }
        GPK_PROF_MARK(h, 1);
        GPK_TRY(gpk_i_gemm(h, true, false, nz, 1, d.rows, 1.0, p->W1, p->ldw, wv, 1, 0.0, q, 1, false));          // q1 = W1^T w
        GPK_TRY(gpk_i_gemm(h, true, false, nz, 1, d.rows, 1.0, p->W2, p->ldw, wv, 1, 0.0, q + nz, 1, false));     // q2 = W2^T w
        gram_form_kernel<<<nz + 1, 256, 0, h->stream>>>(nz, p->G, p->ldg, dcol, q, Hb, ldh);
        GPK_LAUNCH_CHECK(h);
        GPK_TRY(gpk_i_dot(h, wv, wv, d.rows, Hb + (long)nz * ldh + nz));
        GPK_HIP(h, hipMemcpyAsync(d_loss, Hb + (long)nz * ldh + nz, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        h->pipe_tev_used = 0; h->prof_pipelined = 0;
        GPK_TRY(gpk_i_potrf(h, Hb, nz + 1, ldh, 0));
    } else if (h->tune.structured && rev >= 2 && p->W1 && p->W2 && p->ldw >= nz + 1 && all_dinv(h, p, d)) {
        // structured solve of the Burgers / Eikonal / Darcy systems (round 6, gpk_gn_structured_prepare): W = W1 diag(d(z)) + W2 in one
        // memory-bound pass over all rows; the F column by its own one-column solve per factor (exact; rows without a factor copied)
        if (h->tune.exact_loss) {
            exact = true;
            exact_late = h->tune.exact_loss == 1 && h->pipe_g && !h->pipe_unavailable;
            if (exact_late) GPK_TRY(exact_loss_build(h, p, d, z));
            else GPK_TRY(exact_loss(h, p, d, z));
        }
        GPK_PROF_MARK(h, 0);
        const int nc = nz + 1, db = p->dinv_block;
        GPK_TRY(gpk_i_workspace(h, (size_t)d.rows * lds * sizeof(double), &W));
        h->work_sig[0] = -1;                                         // (the workspace no longer holds a solve of a known shape)
        double* coef = S;                                            // nz doubles of scratch: row 0 of S left of its F column
        structured_coeff_general_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(p->system, p->Nd, nz, rev, p->p0, p->p1, p->rhs_f, z, coef);
        structured_form_general_kernel<<<d.rows, 256, 0, h->stream>>>(nz, p->W1, p->W2, p->ldw, coef, W, lds);
        GPK_LAUNCH_CHECK(h);
        GPK_TRY(build(h, p, z, S, lds, nz, 0));                      // F(z) into column nz of S (every row has an entry)
        for (int k = 0; k < d.ngroups; ++k) {
            const Group& g = d.g[k];
            if (g.n <= 0) continue;
            if (g.L) GPK_TRY(gpk_i_trsm_left_dinv(h, g.L, g.Dinv, db, g.n, g.ldl, S + (long)g.off * lds + nz, lds, W + (long)g.off * lds + nz, lds, 1, 0, 0));
            else GPK_HIP(h, hipMemcpy2DAsync(W + (long)g.off * lds + nz, (size_t)lds * 8, S + (long)g.off * lds + nz, (size_t)lds * 8, 8, g.n,
                                             hipMemcpyDeviceToDevice, h->stream));
        }
        (void)nc;
        GPK_PROF_MARK(h, 1);
    } else {
        // the loss of the iterate this step starts from, exact (true substitution with the factors, one vector), in front of the solve
        if (h->tune.exact_loss) {
            exact = true;
            exact_late = h->tune.exact_loss == 1 && h->pipe_g && !h->pipe_unavailable;
            if (exact_late) GPK_TRY(exact_loss_build(h, p, d, z));
            else GPK_TRY(exact_loss(h, p, d, z));
        }
        GPK_TRY(assemble_normal_equations(h, p, d, z, S, lds, Hb, ldh, 1.0, rev, &W));
    }
    // Hb = W^T W and its Cholesky factor, pipelined by column blocks (gpk_factor.hip); d_loss = Hb[nz][nz] before factoring;
    // the last row of the factor is (L_H^{-1} g/2)^T
    if (rev == 4 && !gram_general) {
        // Darcy: Hb = W_u^T W_u (u-part rows + the data rows below them, piecewise profile) + W_a^T W_a (the a-part's own staircase on
        // its sub-square of columns, its F column as a border row), then the factorisation
        const int Nd = p->Nd, nc = nz + 1;
        const Group& ga = d.g[0];
        const Group& gu = d.g[1];
        const double* Wa = W + (long)ga.off * lds;
        const double* Wu = W + (long)gu.off * lds;
        const int ph = h->prof_phase;
        h->prof_phase = 2;                                           // (flop accounting: the product)
        h->pipe_tev_used = 0; h->prof_pipelined = 0;
        if (h->prof) {
            while (h->pipe_tev.size() < 2) { hipEvent_t e; GPK_HIP(h, hipEventCreate(&e)); h->pipe_tev.push_back(e); }
            GPK_HIP(h, hipEventRecord(h->pipe_tev[0], h->stream));
        }
        h->stair = darcy_u_profile(Nd); h->stair_col0 = 0; h->stair_row0 = 0;
        int rc = gpk_i_gemm(h, true, false, nc, nc, d.rows - gu.off, 1.0, Wu, lds, Wu, lds, 0.0, Hb, ldh, true, 1);
        h->stair = GpkStair();
        const bool cached_a = p->Wa && p->Ha;                       // (checked: ldwa, ldha >= 3 N_d)
        if (rc == 0 && cached_a) {
            add_lower_kernel<<<3 * Nd, 256, 0, h->stream>>>(3 * Nd, p->Ha, p->ldha, Hb + (long)Nd * ldh + Nd, ldh);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = gpk_fail(h, e, "add_lower_kernel", __FILE__, __LINE__);
        } else if (rc == 0)
            rc = gpk_i_gemm(h, true, false, 3 * Nd, 3 * Nd, ga.n, 1.0, Wa + Nd, lds, Wa + Nd, lds, 1.0, Hb + (long)Nd * ldh + Nd, ldh, true, 3 * Nd);
        if (rc == 0) rc = gpk_i_gemm(h, true, false, 1, 3 * Nd, ga.n, 1.0, Wa + nz, lds, cached_a ? p->Wa : Wa + Nd, cached_a ? p->ldwa : lds, 1.0,
                                     Hb + (long)nz * ldh + Nd, ldh, false, 3 * Nd);
        if (rc == 0) rc = gpk_i_gemm(h, true, false, 1, 1, ga.n, 1.0, Wa + nz, lds, Wa + nz, lds, 1.0, Hb + (long)nz * ldh + nz, ldh, false);
        h->prof_phase = ph;
        GPK_TRY(rc);
        if (h->prof) { GPK_HIP(h, hipEventRecord(h->pipe_tev[1], h->stream)); h->pipe_tev_used = 2; }
        GPK_HIP(h, hipMemcpyAsync(d_loss, Hb + (long)nz * ldh + nz, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        GPK_TRY(gpk_i_potrf(h, Hb, nc, ldh, 0));
    } else if (!gram && !gram_general) GPK_TRY(gpk_i_syrk_potrf(h, W, lds, d.rows, nz + 1, rev ? nz : 0, Hb, ldh, d_loss));
    GPK_PROF_MARK(h, 2);
    GPK_PROF_MARK(h, 3);
    // the chain of the exact loss: on the GEMM partition's stream from here on (pipelined phase: that stream's last product finished before the
    // last panel chain did, so it starts early; otherwise behind the factorisation just issued), next to the tail below
    // (any error return from here to the end of the call first drains the side stream: the next call's exact_loss_build memsets d_loss_work
    // on the main stream, which must not happen under a chain that is still reading it)
    struct ChainGuard {
        gpk_handle h; bool armed;
        ~ChainGuard() { if (armed && h->pipe_g) (void)hipStreamSynchronize(h->pipe_g); }
    } chain_guard{h, false};
    bool chain_on_side = false;
    if (exact_late) {
        GPK_TRY(exact_loss_chain(h, d, !h->prof_pipelined, &chain_on_side));
        chain_guard.armed = chain_on_side;
    }
    double* dl = rev ? S : delta;                                    // scratch for the (reversed-order) solution: S is free now
    GPK_HIP(h, hipMemcpyAsync(dl, Hb + (long)nz * ldh, (size_t)nz * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    GPK_TRY(gpk_i_trsv(h, true, Hb, nz, ldh, dl));
    if (rev) {
        reverse_copy_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, dl, delta, rev, p->Nd);
        axpy_rev_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, -step_size, dl, z, rev, p->Nd);
    } else {
        axpy_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, -step_size, delta, z);
    }
    GPK_LAUNCH_CHECK(h);
    GPK_PROF_MARK(h, 4);
    // the two host scalars of the step through pinned memory (h_pinned[0] = loss, the int behind h_pinned[1] = pivot status)
    GPK_HIP(h, hipMemcpyAsync(h->h_pinned + 1, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (chain_on_side) GPK_HIP(h, hipStreamWaitEvent(h->stream, h->ev_loss[1], 0));
    GPK_HIP(h, hipMemcpyAsync(h->h_pinned, exact ? h->d_scalars + 8 : d_loss, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    chain_guard.armed = false;                                       // the main stream waited for the chain's event: it is complete
    int info = *reinterpret_cast<const int*>(h->h_pinned + 1);
    const double loss = h->h_pinned[0];
    if (h->prof) {
        for (int i = 0; i < 4; ++i) {
            float ms = 0.f;
            GPK_HIP(h, hipEventElapsedTime(&ms, h->pev[i], h->pev[i + 1]));
            h->prof_ms[i] += (double)ms;
        }
        h->prof_cnt += 1;
        for (int i = 0; i + 1 < h->pipe_tev_used; i += 2) {          // SYRK launches of the pipelined product (GEMM stream)
            float ms = 0.f;
            GPK_HIP(h, hipEventElapsedTime(&ms, h->pipe_tev[i], h->pipe_tev[i + 1]));
            h->prof_syrk_ms += (double)ms;
        }
    }
    if (info == nz + 1) info = 0;          // the border pivot loss - y^T y is not part of H (may round below zero)
    if (host_info) *host_info = info;
    if (host_loss_in) *host_loss_in = loss;
    return 0;
}

extern "C" int gpk_gn_loss(gpk_handle h, const gpk_gn_problem* p, const double* z, double* work, double* host_loss) {
    if (!h || !z || !work || !host_loss) return GPK_ERR_ARG;
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    GPK_TRY(substitution_loss(h, p, d, z, work, h->d_scalars + 1));
    GPK_HIP(h, hipMemcpyAsync(host_loss, h->d_scalars + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int gpk_gn_hessian_grad(gpk_handle h, const gpk_gn_problem* p, const double* z, double* S, int lds, double* H,
                                   int ldh, double* g) {
    if (!h || !z || !S || !H) return GPK_ERR_ARG;
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    GPK_TRY(assemble_normal_equations(h, p, d, z, S, lds, H, ldh, 2.0, 0));   // H = 2 S^T S, g = 2 S^T w in the border row
    if (g) GPK_HIP(h, hipMemcpyAsync(g, H + (long)d.nz * ldh, (size_t)d.nz * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return gpk_symmetrize_lower(h, H, d.nz, ldh);
}

extern "C" int gpk_gn_build(gpk_handle h, const gpk_gn_problem* p, const double* z, double* S, int lds) {
    if (!h || !z || !S) return GPK_ERR_ARG;
    Dims d;
    gpk_gn_problem q = *p;
    if (gn_dims(&q, d) != 0) return gpk_bad_arg(h, "gn: system id / sizes");
    if (lds < d.nz + 1) return gpk_bad_arg(h, "gn: lds < nz+1");
    GPK_HIP(h, hipMemsetAsync(S, 0, (size_t)d.rows * lds * sizeof(double), h->stream));
    return build(h, &q, z, S, lds, d.nz, 1);
}

extern "C" int gpk_gn_build_rev(gpk_handle h, const gpk_gn_problem* p, const double* z, double* S, int lds) {
    if (!h || !z || !S || !p) return GPK_ERR_ARG;
    // (round 6: also the Eikonal and Burgers systems in the staircase orders of their gpk_gn_step -- the sharded step uses them)
    const int rev = step_layout(h, p);
    if (rev < 1 || rev > 4 || p->system == GPK_GN_ELLIPTIC_RELAXED) return gpk_bad_arg(h, "gn_build_rev: not for the relaxed system / needs the leading-zero layout");
    Dims d;
    gpk_gn_problem q = *p;
    if (gn_dims(&q, d) != 0) return gpk_bad_arg(h, "gn: system id / sizes");
    if (lds < d.nz + 1) return gpk_bad_arg(h, "gn: lds < nz+1");
    GPK_HIP(h, hipMemsetAsync(S, 0, (size_t)d.rows * lds * sizeof(double), h->stream));
    return build(h, &q, z, S, lds, d.nz, 1, rev);
}

extern "C" int gpk_axpy(gpk_handle h, int n, double alpha, const double* x, double* y) {
    if (!h || !x || !y || n < 0) return GPK_ERR_ARG;
    if (n == 0) return 0;
    axpy_kernel<<<gpk_ceil_div(n, 256), 256, 0, h->stream>>>(n, alpha, x, y);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

extern "C" int gpk_gn_measurement(gpk_handle h, const gpk_gn_problem* p, const double* z, double* out) {
    if (!h || !z || !out) return GPK_ERR_ARG;
    Dims d;
    gpk_gn_problem q = *p;
    if (gn_dims(&q, d) != 0) return gpk_bad_arg(h, "gn: system id / sizes");
    GPK_HIP(h, hipMemsetAsync(out, 0, (size_t)d.rows * sizeof(double), h->stream));
    return build(h, &q, z, out, 1, 0, 0);
}

// ---- building blocks of the multi-GPU step (gpk_mg.hip) ---------------------------------------------------------------------
// exact in-step loss for the sharded step (replicated on every rank: one vector), on h->stream; the device scalar comes back
int gpk_i_gn_exact_loss(gpk_handle h, const gpk_gn_problem* p, const double* z, double** d_out) {
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    GPK_TRY(exact_loss(h, p, d, z));
    *d_out = h->d_scalars + 8;
    return 0;
}

int gpk_i_gn_layout(gpk_handle h, const gpk_gn_problem* p) { return step_layout(h, p); }
void gpk_i_gn_layout_enter(gpk_handle h, const gpk_gn_problem* p, int rev) {
    if (rev == 2 && h->tune.use_dinv && p->Dinv && p->dinv_block > 0 && h->tune.eikonal_lz != 2) h->stair = eikonal_profile(p->Nd);
    h->lead_div = rev == 3 ? 3 : 1;
}
void gpk_i_gn_layout_leave(gpk_handle h) { h->stair = GpkStair(); h->stair_col0 = h->stair_row0 = 0; h->stair_base = 0; h->lead_div = 1; }
int gpk_i_gn_first_row(gpk_handle h, int nz, int c) {
    if (c >= nz) return 0;                                           // the F column is dense
    if (h->stair.nseg > 0) return gpk_stair_min(h->stair, c, c + 1);
    const int sd = h->lead_div > 0 ? h->lead_div : 1;
    return (nz - 1 - c) / sd;
}

// Darcy pieces of the sharded step (gpk_mg.hip): the u-part's three-segment profile into the handle (gpk_i_gn_layout_leave resets it) ...
void gpk_i_gn_darcy_profile(gpk_handle h, int Nd) { h->stair = darcy_u_profile(Nd); h->stair_col0 = h->stair_row0 = 0; }

// ... and the cached a-part added to the rows [r0, r1) of Hb a rank has just computed from the u-part and data rows: H_a on the rows that lie
// in [N_d, 4 N_d) and, if row n_z is among them, the border terms (L_a^{-1}F_a)^T W_a and |L_a^{-1}F_a|^2 (aF: the solved a-part F column,
// 3 N_d entries with stride ldaf) -- the launches of gpk_gn_step's rev = 4 branch, restricted to those rows
__global__ __launch_bounds__(256) void add_lower_rows_kernel(int row0, const double* __restrict__ A, long lda, double* __restrict__ Cm, long ldc) {
    const long i = row0 + blockIdx.x;                                 // row of the 3 N_d x 3 N_d lower triangle
    const double* a = A + i * lda;
    double* c = Cm + i * ldc;
    for (int j = threadIdx.x; j <= (int)i; j += 256) c[j] += a[j];
}

int gpk_i_gn_darcy_add_a(gpk_handle h, const gpk_gn_problem* p, double* Hb, int ldh, int r0, int r1, const double* aF, int ldaf) {
    const int Nd = p->Nd, na = 3 * Nd, nz = 6 * Nd;
    const int a0 = std::max(r0, Nd), a1 = std::min(r1, 4 * Nd);
    if (a1 > a0) {
        add_lower_rows_kernel<<<a1 - a0, 256, 0, h->stream>>>(a0 - Nd, p->Ha, p->ldha, Hb + (long)Nd * ldh + Nd, ldh);
        GPK_LAUNCH_CHECK(h);
    }
    if (r0 <= nz && nz < r1) {
        const GpkStair keep = h->stair;                              // (closed-form staircase of the a-part's own columns, as on one GPU)
        h->stair = GpkStair(); h->stair_col0 = h->stair_row0 = 0;
        int rc = gpk_i_gemm(h, true, false, 1, na, na, 1.0, aF, ldaf, p->Wa, p->ldwa, 1.0, Hb + (long)nz * ldh + Nd, ldh, false, na);
        if (rc == 0) rc = gpk_i_gemm(h, true, false, 1, 1, na, 1.0, aF, ldaf, aF, ldaf, 1.0, Hb + (long)nz * ldh + nz, ldh, false);
        h->stair = keep;
        GPK_TRY(rc);
    }
    return 0;
}

// n_z and the row count of the system handled by the sharded step
int gpk_i_gn_dims(gpk_handle h, const gpk_gn_problem* p, int* nz, int* rows) {
    Dims d;
    GPK_TRY(check_prob(h, p, d));
    *nz = d.nz; *rows = d.rows;
    return 0;
}

// The tail of a step once the bordered matrix Hb is factored (its last row = (L_H^{-1} g/2)^T): backward solve, back to the natural
// order of the unknowns (rev: the staircase order of gpk_gn_step), update of z.  scratch: nz doubles (rev only).
int gpk_i_gn_finish(gpk_handle h, const gpk_gn_problem* p, int nz, int rev, const double* Hb, int ldh, double* scratch, double* delta,
                    double* z, double step_size) {
    double* dl = rev ? scratch : delta;
    GPK_HIP(h, hipMemcpyAsync(dl, Hb + (long)nz * ldh, (size_t)nz * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    GPK_TRY(gpk_i_trsv(h, true, Hb, nz, ldh, dl));
    if (rev) {
        reverse_copy_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, dl, delta, rev, p->Nd);
        axpy_rev_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, -step_size, dl, z, rev, p->Nd);
    } else {
        axpy_kernel<<<gpk_ceil_div(nz, 256), 256, 0, h->stream>>>(nz, -step_size, delta, z);
    }
    GPK_LAUNCH_CHECK(h);
    return 0;
}

