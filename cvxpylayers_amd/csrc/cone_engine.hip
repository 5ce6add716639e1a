// cone_engine.hip -- MI355X (gfx950 / CDNA4) batched cone-program solve + differentiate engine.
//
// One problem instance per 256-thread workgroup (4 wave64).  The instance's data (dense A in solver
// form, the explicit inverse G of the reduced KKT matrix, all iterates) live in LDS for the whole
// solve: HBM is touched once per instance on the way in (coalesced batch-major value rows) and once
// on the way out.  fp64 throughout (the reference returns float64, diffcp_if.py:374-375).
//
// Forward  : homogeneous self-dual embedding + Douglas-Rachford splitting (SCS 3 algorithm, restated
//            in oracle/cone_oracle.c which this file must agree with), with the per-iteration KKT
//            solve done as  t = rho_x w_x - A^T w_y ;  p_x = G t ;  p_y = w_y + Dy (A p_x)  where
//            G = (rho_x I + A^T Dy A)^{-1} is formed once per (re)scaling by in-LDS Gauss-Jordan.
//            Triangular solves are latency-bound on a GPU; an explicit inverse turns them into matvecs.
// Backward : diffcp's adjoint  M^T r = dz  solved DIRECTLY: the homogeneous embedding makes M singular
//            along z, dA/db/dc are invariant to that null component, so r_tau is pinned to 0 and the
//            remaining (n+m) system is reduced, cone block by cone block, with the spectral structure
//            of DPi (eigenvalue 1 -> equality row, 0 -> eliminated, lambda in (0,1) -> Schur term) to a
//            symmetric saddle system of size n + #equality rows, solved by Gauss-Jordan with partial
//            pivoting in LDS.  See DESIGN.md section "Backward".
//
// Reference boundary mirrored: cvxpylayers/interfaces/diffcp_if.py:46-96,329-403.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <type_traits>
#include <utility>
#include <atomic>
#include <vector>

#include "cone_engine.h"
#include "ce_types.h"

namespace {
#include "ce_common.h"
#include "ce_expcone.h"
#include "ce_forward_rt.h"     // NT2, SOC_SMALL, RT_NVEC / RT_EXTRA (launch planning); its kernels are instantiated in ce_tu_fwd_other.hip
#include "ce_forward_v2.h"     // psd_project (used by k_ca_psd); k_fwd2 itself is instantiated in ce_tu_fwd2.hip
#include "ce_global_mv.h"
#include "ce_backward.h"       // k_transpose, k_parammap*  (k_backward is instantiated in ce_tu_bwd_generic.hip)
#include "ce_backward_rt.h"    // bwd_rt_union_doubles, BGC (launch planning); kernels in ce_tu_bwd_rt.hip
#include "ce_psd_mfma.h"
#include "ce_const_a.h"
#include "ce_shared_a.h"
#include "ce_shared_a_mi.h"
#include "ce_shared_a_fwd.h"
}  // namespace

// ================================================================================================
// host side
// ================================================================================================
struct ce_engine {
    int device = 0;
    DevT T{};
    std::vector<int> q, s;
    double *d_pw = nullptr;
    int *d_rowidx = nullptr, *d_colidx = nullptr, *d_rowcone = nullptr, *d_qoff = nullptr, *d_soff = nullptr, *d_sord = nullptr;
    // workspace
    double *wsA = nullptr; size_t wsA_bytes = 0;          // batch-major copy of A_vals  [B][nnz_aug]
    double *wsdA = nullptr; size_t wsdA_bytes = 0;        // batch-major dA              [B][nnz_aug]
    double *gws = nullptr; size_t gws_bytes = 0;          // global residency fallback
    const double *retained_A = nullptr; int retained_B = 0;
    // launch plan
    int fwd_mode = 0, bwd_mode = 0; size_t fwd_lds = 0, bwd_lds = 0; int nkcap = 0, ldk = 0;
    uintptr_t summary_host_checked = 0; char *summary_host_dev = nullptr;      // ce_status_summary: the last 64-byte line of host memory examined and its device alias (null: not mapped)
    int rt_variant = -1, rt_vp = 0, rt_lda = 0;   // register-tiled forward kernel variant (-1: generic kernel)
    int f2_variant = -1; int *d_idx_at = nullptr, *d_idx_ar = nullptr, *d_idx_b = nullptr; int f2_ldg = 0;   // second-generation forward kernel
    int *d_csc_ptr = nullptr, *d_csr_ptr = nullptr, *d_csr_col = nullptr, *d_csr_src = nullptr;   // sparse structure of the A part (shared-A kernels)
    // split of the A part into singleton rows and sp_r <= 64 dense rows (ce_shared_a_ops.h); sp_RP == 0: more than 64 rows with several entries
    int sp_r = 0, sp_RP = 0;
    bool sa_fwd_attr = false, sa_lsqr_attr = false, sa_lsqr_mi_attr = false;
    int lsqr_variant = 0;                          // 0: LSQR, 1: LSMR (ce_set_lsqr_variant; the calls that solve EVERY instance iteratively: ce_vjp_shared_a, ce_vjp_lsqr)
    int *d_summary = nullptr; unsigned summary_next = 0;   // ce_status_summary staging (8 slots of 3 ints)
    double *d_qT = nullptr; size_t qT_bytes = 0;         // batch-major copy of the objective values for the LSQR adjoint kernels (vjp_lsqr_launch)
    double *d_aa_ws = nullptr; size_t aa_ws_bytes = 0;   // Anderson-acceleration history of the shared-A forward kernel ([B][4][lp])
    unsigned long long *d_psd_stats = nullptr;     // CE_PSD_STATS=1: counters of the PSD projection (printed to stderr by ce_destroy)   // MaxDynamicSharedMemorySize is per device: set once per engine (an engine is bound to one device, one caller thread)
    int psd_first = 0;           // first row of the first PSD block (m when the template has none)
    int *d_sp_drow = nullptr, *d_sp_srow_col = nullptr, *d_sp_scol_ptr = nullptr, *d_sp_scol_row = nullptr, *d_sp_rowslot = nullptr, *d_sp_sing_i = nullptr; double *d_sp_sing_v = nullptr;
    double *d_sp_AdT = nullptr, *d_sp_sval = nullptr;
    int *d_bpos = nullptr;      // [m] position of the row's b entry in the boundary's value order (-1: structurally zero): the tau column of the shared-A adjoint
    bool wl = false; int wl_nq = 0; int *d_row_perm = nullptr, *d_k_rowcone = nullptr, *d_k_qoff = nullptr;   // rows packed so that cones are wave-local (k_fwd2 WL)
    // longest-first dispatch (ce_set_dispatch_history): workgroup -> instance order for the next solve of the same batch size, from this solve's iteration counts
    bool dispatch_history = false; int *d_order = nullptr; int order_B = 0, order_cap = 0;
    int *d_iters2 = nullptr; int iters2_cap = 0, order_pending_B = 0; const int *last_status = nullptr;      // last_status: the status vector of the solve whose order is pending
    int *d_iters_prev = nullptr; int iters_prev_cap = 0, iters_prev_B = 0;      // the iteration counts of the call before (k_dispatch_order compares: is the history predictive?)      // engine-owned copy of the last solve's iteration counts (the caller's buffer may be gone when the order is computed)
    int brt_variant = -1;                          // register-tiled backward kernel variant (-1: generic kernel)
    int ns_variant = -1; size_t ns_lds = 0;       // search-free null-space adjoint (ce_backward_ns.h), -1: not applicable
    // two-tile plan of the register-tiled adjoint: a smaller tile serves the instances it holds, the worst-case tile re-runs the ones it flagged.  The smaller
    // tile is chosen from the LARGEST system of the previous call of the same batch size (nk_*: device maximum, copied to pinned memory behind the launch)
    bool two_tile = false; int fast_forced = -1;
    int *d_nkmax = nullptr, *h_nkmax = nullptr; hipEvent_t nk_ev = nullptr; bool nk_pending = false, nk_have = false, nk_zeroed = false; int nk_last = 0, nk_B = 0;
    // re-solve of rank-deficient adjoint systems by LSQR (ce_set_adjoint_resolve): fix[0] = number of listed instances, fix[1 ...] = the instances the elimination
    // kernels flagged (appended on the device); diffcp's LSQR rule
    const double *call_q = nullptr; long call_sqk = 0, call_sqb = 0;      // (ce_vjp -> ce_vjp_qp: the objective values of the call in flight)
    bool resolve = true; int *d_fix = nullptr; int fix_cap = 0, fix_par = 0; double rs_atol = 1e-8, rs_btol = 1e-8, rs_conlim = 1e8; int rs_iter_lim = 0;
    // quadratic objective
    int nnz_p = 0, p_tri = 0; bool qp_native = false;
    bool aa_ok = false;                            // the forward launch carries the LDS for the Anderson-acceleration vectors
    std::vector<int> p_rows, p_cols;               // host copy of the P structure (entry -> (row, col))
    int *d_idx_p = nullptr, *d_pmap = nullptr, *d_prow = nullptr, *d_pcol = nullptr;
    // profiling
    int prof = 0;      // bit w: launches of kind w (0 forward, 1 adjoint, 2 layout passes) are bracketed by HIP events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[3];
    std::vector<hipEvent_t> ev_pool;
};

#define HIPCHK(call)                                                                 \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return CE_E_HIP;                                                         \
        }                                                                            \
    } while (0)

#ifdef CE_TIMING
static constexpr size_t LDS_LIMIT = 160 * 1024 - 512;   // debug build: room for the static time-stamp array
#else
static constexpr size_t LDS_LIMIT = 160 * 1024;
#endif

static size_t fwd_lds_bytes(const DevT &T, bool a_lds, bool g_lds, bool panel = false) {
    const int n = T.n, m = T.m, l = n + m + 1, PB = std::max(NT, std::max(n, m));
    size_t d = 0;
    if (a_lds) d += (size_t)m * T.lda;
    if (g_lds) d += (size_t)n * T.ldg;
    d += 2 * (size_t)m + 2 * (size_t)n + 5 * (size_t)l + std::max(n, m) + 2 * (size_t)PB + NW * 8 + 2 * std::max(T.nq, 1) + NW + 2 * (size_t)n;
    if (T.ns > 0) d += 2 * (size_t)T.maxs * T.maxs + 2 * (size_t)T.maxs + 8;      // PSD cones: Jacobi scratch of psd_project (ce_forward_generic.h carve)
    d += (size_t)(T.nep + T.np) + 1;                                               // roots of the exponential / power triples (+ alignment)
    if (!g_lds && panel) d += generic_gj_panel_doubles(n) + 2;      // panels of the blocked inversion of the global-memory G
    return d * 8 + 16;
}
// register-tiled forward variants: {CH1, T1, TG, CH2, T2}
static const int RT_VARIANTS[3][6] = {{8, 13, 7, 4, 13, 160}, {8, 16, 8, 4, 16, 208}, {4, 32, 32, 4, 32, 272}};
static bool rt_fits(const DevT &T, int v, int *vp, size_t *bytes, int *lda_out) {
    const int CH1 = RT_VARIANTS[v][0], T1 = RT_VARIANTS[v][1], TG = RT_VARIANTS[v][2], CH2 = RT_VARIANTS[v][3], T2 = RT_VARIANTS[v][4];
    const int VP = RT_VARIANTS[v][5];
    if (T.n * CH1 > NT2 || T.m * CH2 > NT2 || CH1 * T1 < T.m || CH1 * TG < T.n || CH2 * T2 < T.n) return false;
    const int reach = std::max(std::max(T.n + T.m + 1, T.n + CH1 * T1), std::max(std::max(CH2 * T2, CH1 * TG), std::max(NT2 / CH1, NT2 / CH2)));
    if (reach > VP) return false;
    int lda = std::max((T.n + 3) & ~3, std::max(CH2 * T2, CH1 * TG));
    while (lda % 8 != 4) lda += 4;      // conflict-free interleaved row reads (ds_read_b64, groups of CH2 lanes per row)
    *vp = VP; *lda_out = lda;
    *bytes = ((size_t)RT_NVEC * VP + RT_EXTRA + (size_t)T.m * lda) * 8;
    return *bytes <= LDS_LIMIT;
}
static size_t bwd_lds_bytes(const DevT &T, bool a_lds, bool k_lds, int nkcap, int ldk, bool panel = false) {
    const int n = T.n, m = T.m, PB = std::max(NT, std::max(n, m)), nqs = std::max(T.nq, 1);
    size_t d = 0;
    if (a_lds) d += (size_t)m * T.lda;
    if (k_lds) d += (size_t)nkcap * ldk;
    d += 5 * (size_t)m + 2 * (size_t)n + 2 * (size_t)nqs * n + 6 * nqs + PB + NW * 8;
    if (T.ns > 0 || T.nep + T.np > 0) d += (size_t)T.ns * T.maxs * T.maxs + (size_t)T.ns * T.maxs + m + 2 * (size_t)NW * T.maxs * T.maxs + 2 * T.maxs + 8 + 9 * (size_t)(T.nep + T.np);      // ce_backward.h carve
    if (!k_lds && panel) d += generic_lu_panel_doubles(nkcap);
    size_t ints = 2 * (size_t)m + 2 * nqs + 2 * (size_t)nkcap + 4;      // (perm + colrow)
    return d * 8 + ints * 4 + 16;
}

#ifndef BRT_HAS_PSD
#define BRT_HAS_PSD 1
#endif
// register-tiled backward variants {TI, TJ, TH}: K tile 16*TI x 16*TJ per workgroup, H tile 16*TH
// {TI, TJ, TH, row residues BGR}: K tile BGR*TI x 16*TJ per workgroup of BGR*16 threads
constexpr int BRT_NV = 7;
static const int BRT_VARIANTS[BRT_NV][4] = {{4, 4, 4, 16}, {5, 5, 4, 16}, {6, 6, 4, 16}, {7, 7, 4, 16}, {7, 7, 7, 16}, {5, 9, 7, 32}, {7, 13, 7, 32}};      // ({5,5,4}, {6,6,4}, {5,9,7|32}: plain cones only -- added as "fast" tiles of the two-tile plan)
static size_t bwd_rt_lds_bytes(const DevT &T, int TI, int TJ, int BGR) {
    const int n = T.n, m = T.m, nqs = std::max(T.nq, 1);
    size_t d = (size_t)m * n /* lda = n */ + 3 * (size_t)m + 2 * (size_t)n + 6 * nqs + BGR * TI + 5 /* pinfo: two 16-byte records + alignment */ + (BGR * 16 / 64) * 8 + bwd_rt_union_doubles(n, m, nqs, TI, TJ, BGR);
    if (T.ns > 0 || T.nep + T.np > 0) d += (size_t)T.ns * T.maxs * T.maxs + (size_t)T.ns * T.maxs + m + 2 * (size_t)(BGR * 16 / 64) * T.maxs * T.maxs + 2 * T.maxs + 8 + 9 * (size_t)(T.nep + T.np);
    size_t ints = 2 * (size_t)m + 2 * nqs + BGC * TJ + BGR * TI + (BGR * 16 / 64) + 1 + 8;
    return d * 8 + ints * 4 + 16;
}

// second-generation forward variants {CHT, T1, CHA, T2, CHG, TG}
// {CHT, T1, CHA, T2, CHG, TG, threads per workgroup}
constexpr int F2_NV = 5;
static const int F2_VARIANTS[F2_NV][7] = {{16, 2, 8, 2, 16, 2, 256}, {8, 8, 4, 8, 8, 4, 256}, {4, 26, 2, 26, 4, 14, 256}, {8, 20, 2, 32, 8, 8, 512}, {4, 30, 4, 26, 4, 26, 512}};
struct F2Dims { int MP, NPa, NPg, NP, VP, O_G; };
static F2Dims f2_dims(int v) {
    const int *V = F2_VARIANTS[v];
    F2Dims d; d.MP = V[0] * V[1]; d.NPa = V[2] * V[3]; d.NPg = V[4] * V[5]; d.NP = std::max(d.NPa, d.NPg); d.VP = d.MP + d.NP + 2;
    const int nw = V[6] / 64;
    d.O_G = 6 * d.VP + 2 * d.MP + 8 * d.NP + nw * 8 + nw + 16 + 20;      // (+20: the ce_math.h table, F2::O_MT)
    return d;
}
// leading dimension of G in LDS: smallest even ld >= NPg for which the 16 lanes of an LDS group (CHG segments x 16/CHG rows)
// read 16 distinct 16-byte bank groups with ds_read_b128
static int f2_pick_ldg(int v) {
    const int CHG = F2_VARIANTS[v][4], TG = F2_VARIANTS[v][5], NPg = CHG * TG;
    for (int ld = NPg; ld < NPg + 64; ld += 2) {
        bool used[16] = {false}; bool ok = true;
        for (int lane = 0; lane < 16 && ok; lane++) {
            const int jg = lane / CHG, cg = lane % CHG;
            const int g = ((jg * ld + TG * cg) / 2) % 16;
            if (used[g]) ok = false; used[g] = true;
        }
        if (ok) return ld;
    }
    return NPg;
}
static bool f2_fits(const DevT &T, int v, int *ldg, size_t *bytes, bool has_p = false) {
    const int *V = F2_VARIANTS[v];
    const F2Dims d = f2_dims(v);
    const int NTH = V[6];
    if ((T.n + 2) * V[0] > NTH || T.m * V[2] > NTH || T.n * V[4] > NTH) return false;   // two extra column groups carry phi
    if (T.m > d.MP || T.n > d.NPa || T.n > d.NPg || T.n + T.m + 1 > NTH) return false;
    if (T.maxq > SOC_SMALL && T.nq > d.NP) return false;
    *ldg = f2_pick_ldg(v);
    if ((size_t)T.n * *ldg < (size_t)d.NPa) return false;
    const size_t psd = (T.ns > 0 ? 2 * (size_t)T.maxs * T.maxs + 2 * (size_t)T.maxs + 8 : 0) + (size_t)(T.nep + T.np);      // Jacobi scratch: S, V, (c, s, p, q) per pair; one root per exponential cone
    const int ntile = (d.NPg + 15) / 16, ldp = (16 * ntile) % 32 == 16 ? 16 * ntile : 16 * ntile + 16;      // F2::NTILE, F2::LDP
    const size_t gsz = std::max(std::max((size_t)T.n * *ldg, (size_t)4 * ldp), (size_t)16 * d.NP);                                        // G region: also one 4-row panel of the S formation
    *bytes = ((size_t)d.O_G + d.MP /* SOC row info (2 int arrays) */ + gsz + psd + (has_p ? d.NP : 0) /* P-hat g_x */) * 8;
    return *bytes <= LDS_LIMIT;
}

// Row order for k_fwd2's wave-local cone exchange (ce_forward_v2.h, WL): the rows of one wave in the (i2, c2) row layout form a
// window of W = 64 / CHA rows, and no cone may straddle two windows.  Zero-cone rows stay first (the kernel tells them by i < z);
// then the SOC blocks in template order, each pushed to the next window when it would straddle, the gap filled with nonnegative
// rows (which are interchangeable: they count as cones of dimension 1); the remaining nonnegative rows go last.  Exact fit only
// (no padding rows: they would change the size of the embedding and with it the iterates); returns false when that fails.
static bool pack_rows(const ce_template *tpl, int W, std::vector<int> &korig, std::vector<int> &k_rowcone, std::vector<int> &k_qoff) {
    const int z = tpl->z, l = tpl->l, m = tpl->m;
    if (tpl->ns > 0 || tpl->nep + tpl->np > 0) return false;
    for (int c = 0; c < tpl->nq; c++) if (tpl->q[c] > W) return false;
    korig.clear(); k_rowcone.assign(m, -1); k_qoff.clear();
    for (int i = 0; i < z; i++) korig.push_back(i);
    int next_single = z, singles_left = l, orig = z + l;
    auto place_single = [&]() { k_rowcone[korig.size()] = (int)k_qoff.size(); k_qoff.push_back((int)korig.size()); korig.push_back(next_single++); singles_left--; };
    for (int c = 0; c < tpl->nq; c++) {
        const int d = tpl->q[c];
        const int used = (int)korig.size() % W;
        if (used + d > W) {
            const int need = W - used;
            if (singles_left < need) return false;
            for (int k = 0; k < need; k++) place_single();
        }
        k_qoff.push_back((int)korig.size());
        for (int k = 0; k < d; k++) { k_rowcone[korig.size()] = (int)k_qoff.size() - 1; korig.push_back(orig++); }
    }
    while (singles_left > 0) place_single();
    k_qoff.push_back((int)korig.size());
    return (int)korig.size() == m;
}

extern "C" {

const char *ce_last_error(void) { return g_err.c_str(); }
int ce_abi_version(void) { return CE_ABI_VERSION; }
// k_fwd2 when its history fits LDS; the first-generation register-tiled k_forward_rt and the size-generic k_forward keep the history in global memory.
int ce_set_lsqr_variant(ce_handle h, int variant) {
    if (!h || variant < 0 || variant > 1) { g_err = "ce_set_lsqr_variant: variant must be 0 (LSQR) or 1 (LSMR)"; return CE_E_BADARG; }
    h->lsqr_variant = variant;
    return CE_OK;
}
int ce_acceleration_available(ce_handle h) { return (h && ((h->fwd_mode == 4 && h->aa_ok) || h->fwd_mode <= 3)) ? 1 : 0; }
int ce_struct_size(int which) { return which == 0 ? (int)sizeof(ce_template) : which == 1 ? (int)sizeof(ce_settings) : -1; }

void ce_default_settings(ce_settings *s) {
    s->eps_abs = 1e-4; s->eps_rel = 1e-4; s->eps_infeas = 1e-7; s->alpha = 1.5; s->rho_x = 1e-6; s->scale = 0.1;
    s->max_iters = 100000; s->normalize = 1; s->adaptive_scale = 1; s->warm_start = 0; s->acceleration_lookback = 10; s->acceleration_interval = 10;   // SCS 3 defaults, which diffcp forwards (diffcp_if.py:356-367)
}

int ce_create(const ce_template *tpl, int device, ce_handle *out) {
    if (!tpl || !out || tpl->n <= 0 || tpl->m <= 0 || !tpl->indices || !tpl->indptr) { g_err = "bad template"; return CE_E_BADARG; }
    if (tpl->nep < 0 || tpl->np < 0 || (tpl->np > 0 && !tpl->p)) { g_err = "bad exponential / power cone description"; return CE_E_BADARG; }
    for (int i = 0; i < tpl->np; i++) if (!(fabs(tpl->p[i]) > 0.0 && fabs(tpl->p[i]) < 1.0)) { g_err = "power cone exponent must lie in (-1, 0) or (0, 1)"; return CE_E_BADARG; }
    int rows = tpl->z + tpl->l;
    for (int i = 0; i < tpl->nq; i++) { if (tpl->q[i] < 1) { g_err = "bad SOC dim"; return CE_E_BADARG; } rows += tpl->q[i]; }
    for (int i = 0; i < tpl->ns; i++) { if (tpl->s[i] < 1) { g_err = "bad PSD order"; return CE_E_BADARG; } rows += tpl->s[i] * (tpl->s[i] + 1) / 2; }
    rows += 3 * tpl->nep + 3 * tpl->np;
    if (rows != tpl->m) { g_err = "cone dims do not add up to m"; return CE_E_BADARG; }
    if (tpl->indptr[tpl->n + 1] != tpl->nnz_aug) { g_err = "indptr[n+1] != nnz_aug"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(device));
    ce_engine *h = new ce_engine();
    h->device = device;
    DevT &T = h->T;
    T.n = tpl->n; T.m = tpl->m; T.nnz_aug = tpl->nnz_aug; T.nnzA = tpl->indptr[tpl->n]; T.z = tpl->z; T.l = tpl->l; T.nq = tpl->nq;
    T.lda = tpl->n | 1; T.ldg = tpl->n | 1; T.maxq = 0;
    for (int i = 0; i < tpl->nq; i++) T.maxq = std::max(T.maxq, tpl->q[i]);     // odd leading dimension: conflict-free ds_read_b64 down a column of rows
    std::vector<int> colidx(tpl->nnz_aug), rowcone(tpl->m, -1), qoff(tpl->nq + 1, 0);
    for (int j = 0; j <= tpl->n; j++)
        for (int k = tpl->indptr[j]; k < tpl->indptr[j + 1]; k++) {
            if (tpl->indices[k] < 0 || tpl->indices[k] >= tpl->m) { delete h; g_err = "row index out of range"; return CE_E_BADARG; }
            colidx[k] = j;
        }
    int r = tpl->z + tpl->l;
    for (int c = 0; c < tpl->nq; c++) { qoff[c] = r; for (int i = 0; i < tpl->q[c]; i++) rowcone[r + i] = c; r += tpl->q[c]; }
    qoff[tpl->nq] = r;
    std::vector<int> soff(tpl->ns + 1, r), sord(std::max(tpl->ns, 1), 0);
    T.ns = tpl->ns; T.maxs = 0;
    for (int c = 0; c < tpl->ns; c++) { soff[c] = r; sord[c] = tpl->s[c]; T.maxs = std::max(T.maxs, tpl->s[c]); r += tpl->s[c] * (tpl->s[c] + 1) / 2; }
    soff[tpl->ns] = r;
    h->psd_first = soff[0];      // (= first row after the second-order cones: PSD blocks, then exponential / power triples, follow)
    T.nep = tpl->nep; T.eoff = r; T.np = tpl->np; T.pw = nullptr;
    if (tpl->np > 0) {
        HIPCHK(hipMalloc(&h->d_pw, sizeof(double) * tpl->np));
        HIPCHK(hipMemcpy(h->d_pw, tpl->p, sizeof(double) * tpl->np, hipMemcpyHostToDevice));
        T.pw = h->d_pw;
    }
    h->q.assign(tpl->q, tpl->q + tpl->nq);
    if (tpl->nnz_p > 0) {
        if (!tpl->p_indices || !tpl->p_indptr || tpl->p_indptr[tpl->n] != tpl->nnz_p) { delete h; g_err = "bad P structure"; return CE_E_BADARG; }
        h->nnz_p = tpl->nnz_p; h->p_rows.resize(tpl->nnz_p); h->p_cols.resize(tpl->nnz_p);
        bool upper = true, lower = true;
        for (int j = 0; j < tpl->n; j++)
            for (int k = tpl->p_indptr[j]; k < tpl->p_indptr[j + 1]; k++) {
                const int i = tpl->p_indices[k];
                if (i < 0 || i >= tpl->n) { delete h; g_err = "P row index out of range"; return CE_E_BADARG; }
                h->p_rows[k] = i; h->p_cols[k] = j; upper = upper && i <= j; lower = lower && i >= j;
            }
        h->p_tri = (upper || lower) ? 1 : 0;
    }
    HIPCHK(hipMalloc(&h->d_rowidx, sizeof(int) * tpl->nnz_aug));
    HIPCHK(hipMalloc(&h->d_colidx, sizeof(int) * tpl->nnz_aug));
    HIPCHK(hipMalloc(&h->d_rowcone, sizeof(int) * tpl->m));
    HIPCHK(hipMalloc(&h->d_qoff, sizeof(int) * (tpl->nq + 1)));
    HIPCHK(hipMemcpy(h->d_rowidx, tpl->indices, sizeof(int) * tpl->nnz_aug, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_colidx, colidx.data(), sizeof(int) * tpl->nnz_aug, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_rowcone, rowcone.data(), sizeof(int) * tpl->m, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_qoff, qoff.data(), sizeof(int) * (tpl->nq + 1), hipMemcpyHostToDevice));
    {   // CSR of the A part (entry positions refer to the boundary's value order) + its CSC column starts, for the shared-A kernels
        const int nnzA = tpl->indptr[tpl->n];
        std::vector<int> rptr(tpl->m + 1, 0), rcol(std::max(nnzA, 1)), rsrc(std::max(nnzA, 1));
        for (int k = 0; k < nnzA; k++) rptr[tpl->indices[k] + 1]++;
        for (int i = 0; i < tpl->m; i++) rptr[i + 1] += rptr[i];
        std::vector<int> fill(rptr.begin(), rptr.end() - 1);
        for (int j = 0; j < tpl->n; j++)
            for (int k = tpl->indptr[j]; k < tpl->indptr[j + 1]; k++) { const int pos = fill[tpl->indices[k]]++; rcol[pos] = j; rsrc[pos] = k; }
        HIPCHK(hipMalloc(&h->d_csc_ptr, sizeof(int) * (tpl->n + 1))); HIPCHK(hipMalloc(&h->d_csr_ptr, sizeof(int) * (tpl->m + 1)));
        HIPCHK(hipMalloc(&h->d_csr_col, sizeof(int) * rcol.size())); HIPCHK(hipMalloc(&h->d_csr_src, sizeof(int) * rsrc.size()));
        HIPCHK(hipMemcpy(h->d_csc_ptr, tpl->indptr, sizeof(int) * (tpl->n + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_csr_ptr, rptr.data(), sizeof(int) * (tpl->m + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_csr_col, rcol.data(), sizeof(int) * rcol.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_csr_src, rsrc.data(), sizeof(int) * rsrc.size(), hipMemcpyHostToDevice));
        std::vector<int> bpos(tpl->m, -1);
        for (int k = tpl->indptr[tpl->n]; k < tpl->indptr[tpl->n + 1]; k++) bpos[tpl->indices[k]] = k;
        HIPCHK(hipMalloc(&h->d_bpos, sizeof(int) * tpl->m));
        HIPCHK(hipMemcpy(h->d_bpos, bpos.data(), sizeof(int) * tpl->m, hipMemcpyHostToDevice));
        // split: rows with one entry / rows with several
        std::vector<int> drow, rowslot(tpl->m, -1), srow_col(tpl->m, -1);
        for (int i = 0; i < tpl->m; i++) {
            const int cnt = rptr[i + 1] - rptr[i];
            if (cnt >= 2) { rowslot[i] = (int)drow.size(); drow.push_back(i); }
            else if (cnt == 1) srow_col[i] = rcol[rptr[i]];
        }
        if ((int)drow.size() <= 64) {
            h->sp_r = (int)drow.size();
            h->sp_RP = h->sp_r <= 16 ? 16 : (h->sp_r <= 32 ? 32 : 64);
            std::vector<int> scol_ptr(tpl->n + 1, 0), scol_row;
            for (int i = 0; i < tpl->m; i++) if (srow_col[i] >= 0) scol_ptr[srow_col[i] + 1]++;
            for (int j = 0; j < tpl->n; j++) scol_ptr[j + 1] += scol_ptr[j];
            scol_row.resize(std::max(scol_ptr[tpl->n], 1));
            std::vector<int> sfill(scol_ptr.begin(), scol_ptr.end() - 1);
            for (int i = 0; i < tpl->m; i++) if (srow_col[i] >= 0) scol_row[sfill[srow_col[i]]++] = i;
            if (drow.empty()) drow.push_back(0);
            auto up = [&](int **dst, const std::vector<int> &v) -> int { HIPCHK(hipMalloc(dst, sizeof(int) * v.size())); HIPCHK(hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice)); return 0; };
            std::vector<int> sing_i(std::max(tpl->n, 1), -1);      // the one singleton row of a column (-1: none, -2: several)
            for (int j = 0; j < tpl->n; j++) { const int cnt = scol_ptr[j + 1] - scol_ptr[j]; sing_i[j] = cnt == 1 ? scol_row[scol_ptr[j]] : (cnt == 0 ? -1 : -2); }
            if (up(&h->d_sp_drow, drow) || up(&h->d_sp_srow_col, srow_col) || up(&h->d_sp_scol_ptr, scol_ptr) || up(&h->d_sp_scol_row, scol_row) || up(&h->d_sp_rowslot, rowslot) || up(&h->d_sp_sing_i, sing_i)) return CE_E_HIP;
            HIPCHK(hipMalloc(&h->d_sp_sing_v, sizeof(double) * std::max(tpl->n, 1)));
            HIPCHK(hipMalloc(&h->d_sp_AdT, sizeof(double) * (size_t)std::max(tpl->n, 1) * h->sp_RP));
            HIPCHK(hipMalloc(&h->d_sp_sval, sizeof(double) * std::max(tpl->m, 1)));
        }
    }
    HIPCHK(hipMalloc(&h->d_soff, sizeof(int) * (tpl->ns + 1))); HIPCHK(hipMalloc(&h->d_sord, sizeof(int) * std::max(tpl->ns, 1)));
    HIPCHK(hipMemcpy(h->d_soff, soff.data(), sizeof(int) * (tpl->ns + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_sord, sord.data(), sizeof(int) * std::max(tpl->ns, 1), hipMemcpyHostToDevice));
    T.soff = h->d_soff; T.sord = h->d_sord;
    T.rowidx = h->d_rowidx; T.colidx = h->d_colidx; T.rowcone = h->d_rowcone; T.qoff = h->d_qoff;
    // residency plan: mode 0 = everything in LDS, 1 = A in LDS / big matrix in global, 2 = both in global
    if (fwd_lds_bytes(T, true, true) <= LDS_LIMIT) h->fwd_mode = 0;
    else if (fwd_lds_bytes(T, true, false) <= LDS_LIMIT) h->fwd_mode = 1;
    else if (fwd_lds_bytes(T, false, false) <= LDS_LIMIT) h->fwd_mode = 2;
    else { ce_destroy(h); g_err = "instance vectors do not fit LDS"; return CE_E_TOO_LARGE; }
    // (modes 1, 2: the blocked inversion needs its column panel in LDS; templates where that does not fit keep the unblocked loop)
    T.gen_blocked_f = (h->fwd_mode >= 1 && fwd_lds_bytes(T, h->fwd_mode <= 1, false, true) <= LDS_LIMIT) ? 1 : 0;
    h->fwd_lds = fwd_lds_bytes(T, h->fwd_mode <= 1, h->fwd_mode == 0, T.gen_blocked_f != 0);
    if (!getenv("CE_FORCE_GENERIC") && T.ns == 0 && T.nep + T.np == 0) {      // (k_forward_rt: zero / nonnegative / second-order cones only)
        for (int v = 0; v < 3; v++) { int vp, ld; size_t by; if (rt_fits(T, v, &vp, &by, &ld)) { h->rt_variant = v; h->rt_vp = vp; h->fwd_lds = by; h->fwd_mode = 3; h->rt_lda = ld; break; } }
    }
    const char *fwd_env = getenv("CE_FWD");      // "v2" (default when it fits), "rt", "generic": A/B switch for benchmarking
    if (!getenv("CE_FORCE_GENERIC") && !(fwd_env && (!strcmp(fwd_env, "rt") || !strcmp(fwd_env, "generic")))) {
        const bool has_p = h->nnz_p > 0 && T.ns == 0 && T.nep + T.np == 0;     // P inside the kernels: plain cones only (else: epigraph form upstream)
        for (int v = 0; v < F2_NV; v++) {
            int ldg; size_t by;
            if (has_p && v < 2) continue;                   // the quadratic-objective kernels are instantiated for variants 2..4
            if (!f2_fits(T, v, &ldg, &by, has_p)) continue;
            const int *V = F2_VARIANTS[v];
            const int CHT = V[0], T1 = V[1], CHA = V[2], T2 = V[3], NTH = V[6];
            const int S1 = (T1 + 3) & ~3, S2 = (T2 + 3) & ~3;       // thread-major gather maps, rows padded to 16 bytes (ce_forward_v2.h idx_stride)
            std::vector<int> pos((size_t)T.m * T.n, -1), ib0(T.m, -1), ib(T.m, -1), iat((size_t)S1 * NTH, -1), iar((size_t)S2 * NTH, -1);
            for (int j = 0; j <= T.n; j++)
                for (int k = tpl->indptr[j]; k < tpl->indptr[j + 1]; k++) { if (j < T.n) pos[(size_t)tpl->indices[k] * T.n + j] = k; else ib0[tpl->indices[k]] = k; }
            // kernel row order: packed for the wave-local cone exchange when the template allows it (plain cones, linear objective)
            std::vector<int> korig(T.m), k_rowcone, k_qoff;
            for (int i = 0; i < T.m; i++) korig[i] = i;
            {
                std::vector<int> ko, krc, kq;
                const char *wl_env = getenv("CE_WL");                  // "0": keep the template's row order (A/B switch for benchmarking)
                if (!has_p && !(wl_env && !strcmp(wl_env, "0")) && pack_rows(tpl, 64 / CHA, ko, krc, kq)) { korig = ko; k_rowcone = krc; k_qoff = kq; h->wl = true; h->wl_nq = (int)kq.size() - 1; }
            }
            for (int r = 0; r < T.m; r++) ib[r] = ib0[korig[r]];
            for (int t = 0; t < NTH; t++) {
                const int j1 = t / CHT, c1 = t % CHT, i2 = t / CHA, c2 = t % CHA;
                for (int k = 0; k < T1; k++) { const int r = T1 * c1 + k; if (j1 < T.n && r < T.m) iat[(size_t)t * S1 + k] = pos[(size_t)korig[r] * T.n + j1]; }
                for (int k = 0; k < T2; k++) { const int c = T2 * c2 + k; if (i2 < T.m && c < T.n) iar[(size_t)t * S2 + k] = pos[(size_t)korig[i2] * T.n + c]; }
            }
            if (h->wl) {
                HIPCHK(hipMalloc(&h->d_row_perm, sizeof(int) * T.m)); HIPCHK(hipMalloc(&h->d_k_rowcone, sizeof(int) * T.m)); HIPCHK(hipMalloc(&h->d_k_qoff, sizeof(int) * k_qoff.size()));
                HIPCHK(hipMemcpy(h->d_row_perm, korig.data(), sizeof(int) * T.m, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_k_rowcone, k_rowcone.data(), sizeof(int) * T.m, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_k_qoff, k_qoff.data(), sizeof(int) * k_qoff.size(), hipMemcpyHostToDevice));
            }
            HIPCHK(hipMalloc(&h->d_idx_at, sizeof(int) * iat.size())); HIPCHK(hipMalloc(&h->d_idx_ar, sizeof(int) * iar.size())); HIPCHK(hipMalloc(&h->d_idx_b, sizeof(int) * T.m));
            HIPCHK(hipMemcpy(h->d_idx_at, iat.data(), sizeof(int) * iat.size(), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->d_idx_ar, iar.data(), sizeof(int) * iar.size(), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->d_idx_b, ib.data(), sizeof(int) * T.m, hipMemcpyHostToDevice));
            h->f2_variant = v; h->f2_ldg = ldg; h->fwd_lds = by; h->fwd_mode = 4;
            {   // five more vectors (w_prev, x_prev, f_prev, f_save, x_save) when they fit: Anderson acceleration available
                const F2Dims dd = f2_dims(v);
                size_t tail = 5 * (size_t)dd.VP;
                const size_t by_aa = by + tail * 8;
                if (by_aa <= LDS_LIMIT) { h->fwd_lds = by_aa; h->aa_ok = true; }
            }
            if (has_p) {      // gather map of the (jg, cg) tile layout and the dense n x n entry map
                const int CHG = V[4], TG = V[5];
                const int SG = (TG + 3) & ~3;
                std::vector<int> pmap((size_t)T.n * T.n, -1), ip((size_t)SG * NTH, -1);
                for (int k = 0; k < h->nnz_p; k++) { pmap[(size_t)h->p_rows[k] * T.n + h->p_cols[k]] = k; if (h->p_tri) pmap[(size_t)h->p_cols[k] * T.n + h->p_rows[k]] = k; }
                for (int t = 0; t < NTH; t++) {
                    const int jg = t / CHG, cg = t % CHG;
                    for (int k = 0; k < TG; k++) { const int c = TG * cg + k; if (jg < T.n && c < T.n) ip[(size_t)t * SG + k] = pmap[(size_t)jg * T.n + c]; }
                }
                HIPCHK(hipMalloc(&h->d_idx_p, sizeof(int) * ip.size())); HIPCHK(hipMalloc(&h->d_pmap, sizeof(int) * pmap.size()));
                HIPCHK(hipMalloc(&h->d_prow, sizeof(int) * h->nnz_p)); HIPCHK(hipMalloc(&h->d_pcol, sizeof(int) * h->nnz_p));
                HIPCHK(hipMemcpy(h->d_idx_p, ip.data(), sizeof(int) * ip.size(), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_pmap, pmap.data(), sizeof(int) * pmap.size(), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_prow, h->p_rows.data(), sizeof(int) * h->nnz_p, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_pcol, h->p_cols.data(), sizeof(int) * h->nnz_p, hipMemcpyHostToDevice));
                h->qp_native = true;
            }
            break;
        }
    }
    if (fwd_env && !strcmp(fwd_env, "generic") && h->fwd_mode == 3) {   // forced generic kernel
        h->rt_variant = -1;
        if (fwd_lds_bytes(T, true, true) <= LDS_LIMIT) h->fwd_mode = 0; else if (fwd_lds_bytes(T, true, false) <= LDS_LIMIT) h->fwd_mode = 1; else h->fwd_mode = 2;
        T.gen_blocked_f = (h->fwd_mode >= 1 && fwd_lds_bytes(T, h->fwd_mode <= 1, false, true) <= LDS_LIMIT) ? 1 : 0;
        h->fwd_lds = fwd_lds_bytes(T, h->fwd_mode <= 1, h->fwd_mode == 0, T.gen_blocked_f != 0);
    }
    h->nkcap = T.n + std::min(T.m, T.n);
    h->ldk = (h->nkcap + 1) | 1;
    if (bwd_lds_bytes(T, true, true, h->nkcap, h->ldk) <= LDS_LIMIT) h->bwd_mode = 0;
    else if (bwd_lds_bytes(T, true, false, h->nkcap, h->ldk) <= LDS_LIMIT) h->bwd_mode = 1;
    else if (bwd_lds_bytes(T, false, false, h->nkcap, h->ldk) <= LDS_LIMIT) h->bwd_mode = 2;
    else { ce_destroy(h); g_err = "instance vectors do not fit LDS"; return CE_E_TOO_LARGE; }
    { const char *e = getenv("CE_F2_NEUMANN"); T.f2_neumann = (e && atoi(e) == 0) ? 0 : 1; }
    T.gen_blocked_b = (h->bwd_mode >= 1 && bwd_lds_bytes(T, h->bwd_mode <= 1, false, h->nkcap, h->ldk, true) <= LDS_LIMIT) ? 1 : 0;
    { const char *gb = getenv("CE_GEN_BLOCKED"); if (gb && !strcmp(gb, "0")) T.gen_blocked_b = 0; }      // A/B switch (tests): the unblocked elimination of the size-generic backward kernel
    h->bwd_lds = bwd_lds_bytes(T, h->bwd_mode <= 1, h->bwd_mode == 0, h->nkcap, h->ldk, T.gen_blocked_b != 0);
    if (!getenv("CE_FORCE_GENERIC")) {
        const bool plain = T.ns == 0 && T.nep + T.np == 0;
        for (int v = 0; v < BRT_NV; v++) {
            const int TI = BRT_VARIANTS[v][0], TJ = BRT_VARIANTS[v][1], TH = BRT_VARIANTS[v][2], BGR = BRT_VARIANTS[v][3];
            if ((v == 1 || v == 2 || v == 5) && !plain) continue;    // (instantiated for plain cones only)
            if (h->nkcap <= BGC * TJ - 1 && h->nkcap <= BGR * TI && T.n <= BGC * TH && bwd_rt_lds_bytes(T, TI, TJ, BGR) <= LDS_LIMIT) {
                h->brt_variant = v; h->bwd_mode = 3; h->bwd_lds = bwd_rt_lds_bytes(T, TI, TJ, BGR); break;
            }
        }
        // Two-tile plan.  The tile above holds the template's WORST case (NK <= n + min(m, n): every row active); the systems of a batch are usually much
        // smaller (metric configuration: NK = 61 .. 81 of 111) and on the worst-case tile most of every pivot's broadcast and rank-1 update runs over
        // empty column slots.  ce_vjp therefore serves the batch on the smallest tile that held the LARGEST system of the previous call (+ margin) and
        // re-runs the instances that tile flags (adj 2) on the worst-case tile, which exits at once for everybody else.  A retry is expensive however few
        // there are (its launch lasts as long as one instance takes on an idle device, ~0.09 ms at the metric configuration: profiles/r04/e_ab_bwd_two_tile.log),
        // hence the history instead of an a-priori guess (config 3: half of the instances have a fully active cone, NK up to 170 of 200 -- no smaller tile).
        // CE_BWD_TWO_TILE=0 disables; CE_BWD_FAST_VARIANT=v forces the first tile (tests).
        const char *tt = getenv("CE_BWD_TWO_TILE"), *fv = getenv("CE_BWD_FAST_VARIANT");
        h->two_tile = h->bwd_mode == 3 && plain && h->nnz_p == 0 && h->brt_variant > 0 && !(tt && !strcmp(tt, "0"));
        h->fast_forced = fv ? atoi(fv) : -1;
    }
    // Search-free null-space adjoint (ce_backward_ns.h): plain cones, linear objective, 4 ceil(n / 4) + 1 columns in the variant's tiles.  It serves ce_vjp calls
    // whose LSQR re-solve is armed (rank-deficient instances are detected, flagged and handed to LSQR, not resolved by the elimination).  CE_BWD_NS=0 disables.
    {
        static const int NSV[3][2] = {{2, 256}, {4, 256}, {7, 512}};
        const char *e = getenv("CE_BWD_NS");
        const bool plain = T.ns == 0 && T.nep + T.np == 0;
        if (plain && h->nnz_p == 0 && !(e && atoi(e) == 0) && !getenv("CE_FORCE_GENERIC")) {
            for (int v = 0; v < 3; v++) {
                if (4 * ((T.n + 3) / 4) + 1 <= 16 * NSV[v][0] && ce_bwd_ns_lds_bytes(T.n, T.m, T.nq, v) <= LDS_LIMIT) { h->ns_variant = v; h->ns_lds = ce_bwd_ns_lds_bytes(T.n, T.m, T.nq, v); break; }
            }
        }
        HIPCHK(ce_setattr_bwd_ns((int)LDS_LIMIT));
    }
    if (h->qp_native && h->bwd_mode != 3) h->qp_native = false;      // the adjoint with P lives in the register-tiled backward kernel
    HIPCHK(ce_setattr_fwd_generic((int)LDS_LIMIT)); HIPCHK(ce_setattr_fwd_rt((int)LDS_LIMIT));
    HIPCHK(ce_setattr_fwd2_plain((int)LDS_LIMIT)); HIPCHK(ce_setattr_fwd2_psd((int)LDS_LIMIT)); HIPCHK(ce_setattr_fwd2_qp((int)LDS_LIMIT));
    HIPCHK(ce_setattr_bwd_rt_plain((int)LDS_LIMIT)); HIPCHK(ce_setattr_bwd_rt_psd((int)LDS_LIMIT)); HIPCHK(ce_setattr_bwd_generic((int)LDS_LIMIT));
    *out = h;
    return CE_OK;
}

int ce_adjoint_ns_variant(ce_handle h) { return h ? h->ns_variant : -1; }
int ce_set_adjoint_resolve(ce_handle h, int enable, double atol, double btol, double conlim, int iter_lim) {
    if (!h) { g_err = "null argument"; return CE_E_BADARG; }
    h->resolve = enable != 0;
    h->rs_atol = atol > 0 ? atol : 1e-8; h->rs_btol = btol > 0 ? btol : 1e-8; h->rs_conlim = conlim; h->rs_iter_lim = iter_lim > 0 ? iter_lim : 0;
    return CE_OK;
}
int ce_destroy(ce_handle h) {
    if (!h) return CE_OK;
    hipSetDevice(h->device);
    hipFree(h->d_rowidx); hipFree(h->d_colidx); hipFree(h->d_rowcone); hipFree(h->d_qoff); hipFree(h->d_soff); hipFree(h->d_sord); hipFree(h->d_pw); hipFree(h->d_idx_p); hipFree(h->d_pmap); hipFree(h->d_prow); hipFree(h->d_pcol);
    hipFree(h->wsA); hipFree(h->wsdA); hipFree(h->gws); hipFree(h->d_idx_at); hipFree(h->d_idx_ar); hipFree(h->d_idx_b); hipFree(h->d_order); hipFree(h->d_iters2); hipFree(h->d_iters_prev); hipFree(h->d_nkmax); if (h->h_nkmax) hipHostFree(h->h_nkmax); if (h->nk_ev) hipEventDestroy(h->nk_ev); hipFree(h->d_row_perm); hipFree(h->d_k_rowcone); hipFree(h->d_k_qoff); hipFree(h->d_csc_ptr); hipFree(h->d_csr_ptr); hipFree(h->d_csr_col); hipFree(h->d_csr_src);
    hipFree(h->d_sp_drow); hipFree(h->d_sp_srow_col); hipFree(h->d_sp_scol_ptr); hipFree(h->d_sp_scol_row); hipFree(h->d_sp_rowslot); hipFree(h->d_sp_sing_i); hipFree(h->d_sp_sing_v); hipFree(h->d_sp_AdT); hipFree(h->d_sp_sval); hipFree(h->d_bpos); hipFree(h->d_aa_ws); hipFree(h->d_qT); hipFree(h->d_summary);
    for (auto &v : h->ev) for (auto &p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto &e : h->ev_pool) hipEventDestroy(e);
    if (h->d_psd_stats) {
        unsigned long long c[16] = {0};
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(c, h->d_psd_stats, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[cone_engine] PSD projections %llu: refinement steps %llu (%.2f per projection), warm Jacobi fall-backs %llu, cold starts %llu; "
                            "clock64 ticks per projection %.0f (of which Jacobi sweeps %.0f), per iteration up to the end of the projection %.0f\n",
                    c[0], c[1], c[0] ? (double)c[1] / (double)c[0] : 0.0, c[2], c[3], c[0] ? (double)c[4] / c[0] : 0.0, c[0] ? (double)c[6] / c[0] : 0.0, c[0] ? (double)c[5] / c[0] : 0.0),
            fprintf(stderr, "[cone_engine]   ticks per projection by phase: T=SV,R %.0f | D=V'T %.0f | E %.0f | reduce %.0f | V+=VE %.0f | X, store %.0f\n",
                    (double)c[8] / (c[0] ? c[0] : 1), (double)c[9] / (c[0] ? c[0] : 1), (double)c[10] / (c[0] ? c[0] : 1), (double)c[11] / (c[0] ? c[0] : 1), (double)c[12] / (c[0] ? c[0] : 1), (double)c[13] / (c[0] ? c[0] : 1));
        hipFree(h->d_psd_stats);
    }
    delete h;
    return CE_OK;
}

// (rows x cols) row-major -> (cols x rows) row-major
static void launch_transpose(hipStream_t st, const double *in, double *out, int rows, int cols) {
    static const int ts = [] { const char *e = getenv("CE_TR_TILE"); return (e && atoi(e) == 32) ? 32 : 64; }();
    if (ts == 32) hipLaunchKernelGGL(k_transpose<32>, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, in, out, rows, cols);
    else hipLaunchKernelGGL(k_transpose<64>, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, in, out, rows, cols);
}
static int ensure(double **ptr, size_t *have, size_t need) {
    if (*have >= need) return CE_OK;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr; *have = 0;
    HIPCHK(hipMalloc(ptr, need));
    *have = need;
    return CE_OK;
}

struct ProfScope {
    ce_engine *h; int which; hipStream_t st; hipEvent_t a{}, b{}; bool on;
    ProfScope(ce_engine *h_, int w, hipStream_t s) : h(h_), which(w), st(s), on(((h_->prof >> w) & 1) != 0) {
        // events come from a pool filled by earlier scopes (ce_reset_profile returns them): creating a pair per launch cost host time in front of every
        // kernel of a profiled run -- bench.py's timed region is one
        if (on) {
            if (h->ev_pool.size() >= 2) { a = h->ev_pool.back(); h->ev_pool.pop_back(); b = h->ev_pool.back(); h->ev_pool.pop_back(); }
            else { hipEventCreate(&a); hipEventCreate(&b); }
            hipEventRecord(a, st);
        }
    }
    ~ProfScope() { if (on) { hipEventRecord(b, st); h->ev[which].push_back({a, b}); } }
};

// batch-minor (K x B) -> batch-major (B x K) when needed; returns the batch-major pointer
static int to_batch_major(ce_engine *h, int B, const double *vals, long sk, long sb, hipStream_t st, const double **out) {
    const int K = h->T.nnz_aug;
    if (sk == 1 && sb == K) { *out = vals; return CE_OK; }
    if (!(sb == 1 && sk == B)) { g_err = "A_vals must be contiguous batch-minor (sk=B,sb=1) or batch-major (sk=1,sb=nnz_aug)"; return CE_E_BADARG; }
    int rc = ensure(&h->wsA, &h->wsA_bytes, sizeof(double) * (size_t)B * K);
    if (rc) return rc;
    {
        ProfScope ps(h, 2, st);
        launch_transpose(st, vals, h->wsA, K, B);
    }
    *out = h->wsA;
    return CE_OK;
}

struct ce_engine;
static int flush_dispatch_order(ce_engine *h, hipStream_t st, const int *sum_status, int *sum_out);      // (defined next to ce_set_dispatch_history)

int ce_qp_native(ce_handle h) { return (h && h->qp_native) ? 1 : 0; }

int ce_solve(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *q_vals, long sq_k, long sq_b,
             const ce_settings *settings, double *x, double *y, double *s, int *iters, int *status, double *resid, void *stream) {
    return ce_solve_qp(h, B, A_vals, sA_k, sA_b, q_vals, sq_k, sq_b, nullptr, settings, x, y, s, iters, status, resid, stream);
}

int ce_solve_qp(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *q_vals, long sq_k, long sq_b,
                const double *P_vals, const ce_settings *settings, double *x, double *y, double *s, int *iters, int *status, double *resid, void *stream) {
    if (!h || B <= 0 || !A_vals || !q_vals || !x || !y || !s || !iters || !status) { g_err = "null argument"; return CE_E_BADARG; }
    if (P_vals && !h->qp_native) { g_err = "quadratic objective: this template does not run P inside the kernels (ce_qp_native == 0); use the epigraph form"; return CE_E_UNSUPPORTED; }
    if (!P_vals && h->qp_native) { g_err = "template created with a P structure: P_vals is required"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    // (PSD / exponential / power cones beyond k_fwd2's sizes run on the size-generic kernel: fwd_mode 0..2)
    if ((h->T.ns > 0 || h->T.nep + h->T.np > 0) && h->fwd_mode == 3) { g_err = "PSD / exponential / power cones: internal error, k_forward_rt selected"; return CE_E_UNSUPPORTED; }
    ce_settings S; if (settings) S = *settings; else ce_default_settings(&S);
    if (!ce_acceleration_available(h)) S.acceleration_lookback = 0;          // k_fwd2 (when its vectors fit LDS) and the size-generic kernel implement it
    if (S.acceleration_interval <= 0) S.acceleration_interval = 10;
    const double *Abm = nullptr;
    int rc = to_batch_major(h, B, A_vals, sA_k, sA_b, st, &Abm);
    if (rc) return rc;
    h->retained_A = Abm; h->retained_B = B;
    const DevT &T = h->T;
    double *gA = nullptr, *gG = nullptr;
    if (h->fwd_mode == 1 || h->fwd_mode == 2) {
        size_t perA = (h->fwd_mode == 2) ? (size_t)T.m * T.lda : 0, perG = (size_t)T.n * T.ldg;
        rc = ensure(&h->gws, &h->gws_bytes, sizeof(double) * (size_t)B * (perA + perG));
        if (rc) return rc;
        gG = h->gws; gA = h->gws + (size_t)B * perG;
    }
    double *aa_ws = nullptr;
    if (h->fwd_mode <= 3 && S.acceleration_lookback > 0) {      // k_forward_rt and the size-generic kernel keep the acceleration history in global memory ([B][4][lp], shared with the shared-A kernel's)
        const size_t l = (size_t)T.n + T.m + 1, lp = l + (l & 1);
        rc = ensure(&h->d_aa_ws, &h->aa_ws_bytes, sizeof(double) * (size_t)B * 4 * lp);
        if (rc) return rc;
        aa_ws = h->d_aa_ws;
    }
    bool fa_iters2 = false;
    {
        ProfScope ps(h, 0, st);
        CeFwdArgs fa{};
        fa.aa_ws = aa_ws;
        fa.T = T; fa.S = S; fa.Abm = Abm; fa.q = q_vals; fa.sqk = sq_k; fa.sqb = sq_b; fa.idx_at = h->d_idx_at; fa.idx_ar = h->d_idx_ar; fa.idx_b = h->d_idx_b;
        fa.x = x; fa.y = y; fa.s = s; fa.iters = iters; fa.status = status; fa.resid = resid; fa.P = P_vals; fa.nnz_p = h->nnz_p; fa.idx_p = h->d_idx_p; fa.gA = gA; fa.gG = gG;
        int lrc;
        if (h->dispatch_history && h->fwd_mode == 4) {
            rc = flush_dispatch_order(h, st, nullptr, nullptr); if (rc) return rc;          // (a solve whose status was never summarised: the order is still owed)
            if (h->iters2_cap < B) { hipFree(h->d_iters2); h->d_iters2 = nullptr; h->iters2_cap = 0; HIPCHK(hipMalloc(&h->d_iters2, sizeof(int) * (size_t)B)); h->iters2_cap = B; }
            fa.iters2 = h->d_iters2; fa_iters2 = true;
        }
        fa.order = (h->dispatch_history && h->order_B == B && h->fwd_mode == 4) ? h->d_order : nullptr;
        if (h->fwd_mode == 4) {
            fa.T.ldg = h->f2_ldg;
            if (h->wl) {      // rows packed for the wave-local cone exchange: the kernel sees the cone layout in ITS row order
                fa.row_perm = h->d_row_perm; fa.T.rowcone = h->d_k_rowcone; fa.T.qoff = h->d_k_qoff; fa.T.nq = h->wl_nq; fa.T.l = 0;
            }
            if (P_vals) lrc = ce_launch_fwd2_qp(h->f2_variant, B, h->fwd_lds, st, fa);
            else if (T.ns > 0 || T.nep + T.np > 0) lrc = ce_launch_fwd2_psd(h->f2_variant, B, h->fwd_lds, st, fa);
            else lrc = ce_launch_fwd2_plain(h->f2_variant, B, h->fwd_lds, st, fa);
        } else if (h->fwd_mode == 3) {
            fa.T.lda = h->rt_lda;
            lrc = ce_launch_fwd_rt(h->rt_variant, B, h->fwd_lds, st, fa);
        } else lrc = ce_launch_fwd_generic(h->fwd_mode, B, h->fwd_lds, st, fa);
        if (lrc) { g_err = "internal: no forward kernel for the planned variant"; return CE_E_BADARG; }
    }
    HIPCHK(hipGetLastError());
    if (fa_iters2) { h->order_pending_B = B; h->last_status = status; }          // (computed by flush_dispatch_order, off the critical path)
    return CE_OK;
}

int ce_vjp(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *q_vals, long sq_k, long sq_b,
           const double *x, const double *y, const double *s, const double *dx, const double *dy,
           double *dA_vals, long sdA_k, long sdA_b, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, void *stream) {
    // b, c do not enter the elimination (r_tau pinned to 0); they do enter diffcp's full system, which the instances the elimination flags as rank deficient
    // are re-solved on (ce_set_adjoint_resolve): q_vals == NULL switches the re-solve off for this call
    if (!h) { g_err = "null argument"; return CE_E_BADARG; }
    h->call_q = q_vals; h->call_sqk = sq_k; h->call_sqb = sq_b;
    const int rc = ce_vjp_qp(h, B, A_vals, sA_k, sA_b, nullptr, x, y, s, dx, dy, dA_vals, sdA_k, sdA_b, dq_vals, sdq_k, sdq_b, nullptr, adj_status, stream);
    h->call_q = nullptr;
    return rc;
}

static int vjp_lsqr_launch(ce_handle h, int B, const double *A_vals0, long sA_b, int per_inst, const double *q_vals, long sq_k, long sq_b,
                           const double *x, const double *y, const double *s, const double *dx, const double *dy,
                           double *dA_bm, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, int *lsqr_iters, double atol, double btol, double conlim, int iter_lim, void *stream,
                           const int *sel = nullptr, int status_or = 0, int *sel_reset = nullptr);
int ce_vjp_qp(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *P_vals,
              const double *x, const double *y, const double *s, const double *dx, const double *dy,
              double *dA_vals, long sdA_k, long sdA_b, double *dq_vals, long sdq_k, long sdq_b, double *dP_vals, int *adj_status, void *stream) {
    if (!h || B <= 0 || !x || !y || !s || !dx || !dy || !dA_vals || !dq_vals) { g_err = "null argument"; return CE_E_BADARG; }
    if ((P_vals != nullptr) != h->qp_native || (P_vals && !dP_vals)) { g_err = "quadratic objective: P_vals / dP_vals must be given exactly when ce_qp_native(h) == 1"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const DevT &T = h->T;
    if ((T.ns > 0 || T.nep + T.np > 0) && h->bwd_mode == 3 && !BRT_HAS_PSD) { g_err = "PSD / exponential cones: the register-tiled adjoint was built without them"; return CE_E_UNSUPPORTED; }
    const double *Abm = nullptr;
    int rc;
    if (A_vals) { rc = to_batch_major(h, B, A_vals, sA_k, sA_b, st, &Abm); if (rc) return rc; }
    else { if (!h->retained_A || h->retained_B != B) { g_err = "no retained forward inputs"; return CE_E_STATE; } Abm = h->retained_A; }
    const int K = T.nnz_aug;
    double *dAbm = nullptr; bool need_tr = false;
    if (sdA_k == 1 && sdA_b == K) dAbm = dA_vals;
    else if (sdA_b == 1 && sdA_k == B) { rc = ensure(&h->wsdA, &h->wsdA_bytes, sizeof(double) * (size_t)B * K); if (rc) return rc; dAbm = h->wsdA; need_tr = true; }
    else { g_err = "dA_vals must be contiguous batch-minor or batch-major"; return CE_E_BADARG; }
    double *gA = nullptr, *gK = nullptr;
    if (h->bwd_mode > 0 && h->bwd_mode < 3) {
        size_t perA = (h->bwd_mode == 2) ? (size_t)T.m * T.lda : 0, perK = (size_t)h->nkcap * h->ldk;
        rc = ensure(&h->gws, &h->gws_bytes, sizeof(double) * (size_t)B * (perA + perK));
        if (rc) return rc;
        gK = h->gws; gA = h->gws + (size_t)B * perK;
    }
    // Rank-deficient adjoint systems (redundant equality rows, degenerate active sets): the elimination kernels set the free variables to zero -- a BASIC
    // solution -- where the reference's LSQR (diffcp_if.py:86 -> adj_batch) returns the minimum-norm one.  The kernels append such instances (and the ones whose
    // system exceeds the register tile) to a device-side list; a fixed grid of LSQR workgroups behind them walks the list and overwrites those instances'
    // gradients with diffcp's answer (k_sa_lsqr with the instance's own A, full (n + m + 1) system, diffcp's stopping rule).  No host round trip; an empty list
    // costs one launch of workgroups that return at once.
    bool do_fix = h->resolve && h->call_q && !P_vals &&
                  sa_lsqr_lds_doubles(T.n, T.m, T.nq, T.ns, T.maxs, 0, h->psd_first, T.nep + T.np) * 8 <= LDS_LIMIT;
    if (do_fix && h->fix_cap < B) {
        if (h->d_fix) { hipFree(h->d_fix); h->d_fix = nullptr; h->fix_cap = 0; }
        HIPCHK(hipMalloc(&h->d_fix, sizeof(int) * 2 * ((size_t)B + 1))); h->fix_cap = B; h->fix_par = 0;
        HIPCHK(hipMemsetAsync(h->d_fix, 0, sizeof(int) * 2 * ((size_t)B + 1), st));      // TWO lists (count | entries), used alternately: the LSQR launch of a call empties the list of the call before
    }
    {
        ProfScope ps(h, 1, st);
        CeBwdArgs ba{};
        int *const fix_cur = do_fix ? h->d_fix + (size_t)h->fix_par * (h->fix_cap + 1) : nullptr, *const fix_oth = do_fix ? h->d_fix + (size_t)(1 - h->fix_par) * (h->fix_cap + 1) : nullptr;
        ba.fix = fix_cur;
        ba.T = T; ba.nkcap = h->nkcap; ba.ldk = h->ldk; ba.Abm = Abm; ba.x = x; ba.y = y; ba.s = s; ba.dx = dx; ba.dy = dy; ba.dA = dAbm; ba.dq = dq_vals;
        ba.sdqk = sdq_k; ba.sdqb = sdq_b; ba.adj = adj_status; ba.P = P_vals; ba.nnz_p = h->nnz_p; ba.pmap = h->d_pmap; ba.prow = h->d_prow; ba.pcol = h->d_pcol;
        ba.p_tri = h->p_tri; ba.dP = dP_vals; ba.gA = gA; ba.gK = gK;
        int lrc;
        if (do_fix && h->ns_variant >= 0) {
            ba.T.lda = T.n;
            lrc = ce_launch_bwd_ns(h->ns_variant, B, h->ns_lds, st, ba);
        } else if (h->bwd_mode == 3) {
            ba.T.lda = T.n;
            int fast = -1; size_t fast_lds = 0;
            if (h->two_tile && adj_status && !P_vals) {
                if (!h->d_nkmax) { HIPCHK(hipMalloc(&h->d_nkmax, sizeof(int))); HIPCHK(hipHostMalloc(&h->h_nkmax, sizeof(int))); HIPCHK(hipEventCreateWithFlags(&h->nk_ev, hipEventDisableTiming)); }
                if (h->nk_pending && hipEventQuery(h->nk_ev) == hipSuccess) {
                    h->nk_last = *h->h_nkmax; h->nk_pending = false; h->nk_have = true;
                    static const bool nk_debug = getenv("CE_NK_DEBUG") != nullptr;
                    if (nk_debug) fprintf(stderr, "cone_engine: largest adjoint system of the previous call: NK = %d (B = %d)\n", h->nk_last, h->nk_B);
                }
                (void)hipGetLastError();          // (hipErrorNotReady of the query is not an error)
                const int need = (h->nk_have && !h->nk_pending && h->nk_B == B) ? h->nk_last + 8 : (1 << 30);
                for (int v = (h->fast_forced >= 0 ? h->fast_forced : 0); v < h->brt_variant; v++) {
                    const int TI = BRT_VARIANTS[v][0], TJ = BRT_VARIANTS[v][1], TH = BRT_VARIANTS[v][2], BGR = BRT_VARIANTS[v][3];
                    if ((h->fast_forced >= 0 || (need <= BGC * TJ - 1 && need <= BGR * TI)) && T.n <= BGC * TH && bwd_rt_lds_bytes(T, TI, TJ, BGR) <= LDS_LIMIT) { fast = v; fast_lds = bwd_rt_lds_bytes(T, TI, TJ, BGR); break; }
                }
                if (!h->nk_zeroed) { HIPCHK(hipMemsetAsync(h->d_nkmax, 0, sizeof(int), st)); h->nk_zeroed = true; }      // (first call; afterwards the counter is reset behind the read-back)
                ba.nk_max = h->d_nkmax;
            }
            if (fast >= 0) {
                ba.nonfinal = 1;      // (an instance this tile does not hold is the retry launch's business, not yet the list's)
                lrc = ce_launch_bwd_rt_plain(fast, B, fast_lds, st, ba);
                ba.retry = 1; ba.nonfinal = 0;
                if (!lrc) lrc = ce_launch_bwd_rt_plain(h->brt_variant, B, h->bwd_lds, st, ba);
            } else
            lrc = (T.ns > 0 || T.nep + T.np > 0) ? ce_launch_bwd_rt_psd(h->brt_variant, B, h->bwd_lds, st, ba) : ce_launch_bwd_rt_plain(h->brt_variant, B, h->bwd_lds, st, ba);
        } else lrc = ce_launch_bwd_generic(h->bwd_mode, B, h->bwd_lds, st, ba);
        if (lrc) { g_err = "internal: no backward kernel for the planned variant"; return CE_E_BADARG; }
        if (do_fix) {
            const int grid = B < 768 ? B : 768;          // three workgroups per CU: what the LSQR kernel's LDS allows; an empty list costs one pass of workgroups that return at once
            const int prof_keep = h->prof; h->prof = 0;          // (inside this scope's bracket already)
            rc = vjp_lsqr_launch(h, grid, Abm, K, 1, h->call_q, h->call_sqk, h->call_sqb, x, y, s, dx, dy, dAbm, dq_vals, sdq_k, sdq_b, adj_status, nullptr,
                                 h->rs_atol, h->rs_btol, h->rs_conlim, h->rs_iter_lim, stream, fix_cur, 4 | 8, fix_oth);
            h->fix_par ^= 1;
            h->prof = prof_keep;
            if (rc) return rc;
        }
        if (ba.nk_max) {      // the largest system of this call, for the tile choice of the next one (read once the copy has landed: no synchronisation here)
            HIPCHK(hipMemcpyAsync(h->h_nkmax, h->d_nkmax, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipEventRecord(h->nk_ev, st));
            HIPCHK(hipMemsetAsync(h->d_nkmax, 0, sizeof(int), st));          // for the next call (kept off the path in front of its kernel)
            h->nk_pending = true; h->nk_B = B;
        }
    }
    if (need_tr) {
        ProfScope ps(h, 2, st);
        launch_transpose(st, dAbm, dA_vals, B, K);
    }
    HIPCHK(hipGetLastError());
    return flush_dispatch_order(h, st, nullptr, nullptr);
}

// summary of an int32 vector v[B] (status of a forward call, or adj_status of a backward call): out[0] = min v, out[1] = #{v == 2} ("solved,
// inaccurate"), out[2] = #{(v & 3) != 0} (adjoint flags: bits 0-1 = failed / too many active rows): what a caller needs to decide whether
// the slow path (per-instance inspection, messages) is necessary at all
__global__ void __launch_bounds__(256) k_status_summary(int B, const int *__restrict__ status, int *__restrict__ out) {
    int mn = 0x7fffffff, n2 = 0, nf = 0;
    for (int i = threadIdx.x; i < B; i += 256) { const int s = status[i]; mn = min(mn, s); n2 += (s == 2); nf += ((s & 3) != 0); }
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o)); n2 += __shfl_xor(n2, o); nf += __shfl_xor(nf, o); }
    __shared__ int sm[12];
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = mn; sm[4 + (threadIdx.x >> 6)] = n2; sm[8 + (threadIdx.x >> 6)] = nf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = min(min(sm[0], sm[1]), min(sm[2], sm[3])); out[1] = sm[4] + sm[5] + sm[6] + sm[7]; out[2] = sm[8] + sm[9] + sm[10] + sm[11];
        __threadfence_system();          // (out may be mapped host memory: the three values are visible to the host before the flag)
        __hip_atomic_store(out + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // out[3] = 1: "ready" -- a host that cleared it before the call may poll it instead of synchronising the stream
    }
}
// order[] = the instances sorted by iteration count, largest first (counting sort over check intervals; ties in arbitrary order): one workgroup.
// order[B] = 1 when the history is PREDICTIVE: at least 70 % of the instances stopped in the same check interval as the instance at the same position of the call
// before (iters_prev, updated here; have_prev = 0: no such call).  Re-solved or slowly changing batches score ~1, unrelated batches of the metric configuration
// ~0.43 (the chance that two draws of the count distribution agree): there the permutation predicts nothing and is not applied (k_fwd2 reads the flag).
// sum_status / sum_out given: the status summary of k_status_summary comes FIRST (its ready flag is what the host polls), the sort behind it in the same launch
__global__ void __launch_bounds__(1024) k_dispatch_order(int B, const int *__restrict__ iters, int *__restrict__ order, int *__restrict__ iters_prev, int have_prev,
                                                         const int *__restrict__ sum_status = nullptr, int *__restrict__ sum_out = nullptr) {
    constexpr int NB = 512;                       // buckets of CONVERGED_INTERVAL iterations; anything longer shares the last one
    __shared__ int cnt[NB], tmp[NB];
    __shared__ int same;
    if (sum_out) {
        int mn = 0x7fffffff, n2 = 0, nf = 0;
        for (int i = threadIdx.x; i < B; i += 1024) { const int s = sum_status[i]; mn = min(mn, s); n2 += (s == 2); nf += ((s & 3) != 0); }
        for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o)); n2 += __shfl_xor(n2, o); nf += __shfl_xor(nf, o); }
        if ((threadIdx.x & 63) == 0) { tmp[threadIdx.x >> 6] = mn; tmp[16 + (threadIdx.x >> 6)] = n2; tmp[32 + (threadIdx.x >> 6)] = nf; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int a = tmp[0], b2 = 0, c = 0;
            for (int w = 0; w < 16; w++) { a = min(a, tmp[w]); b2 += tmp[16 + w]; c += tmp[32 + w]; }
            sum_out[0] = a; sum_out[1] = b2; sum_out[2] = c;
            __threadfence_system();
            __hip_atomic_store(sum_out + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) same = 0;
    for (int b = threadIdx.x; b < NB; b += 1024) cnt[b] = 0;
    __syncthreads();
    {
        int mine = 0;
        for (int i = threadIdx.x; i < B; i += 1024) {
            const int it = iters[i];
            if (have_prev) mine += (max(it, 0) / CONVERGED_INTERVAL == max(iters_prev[i], 0) / CONVERGED_INTERVAL);
            iters_prev[i] = it;
        }
        if (have_prev && mine) atomicAdd(&same, mine);
    }
    __syncthreads();
    // not predictive (unrelated batches, the first call): the order would not be applied -- the sort is skipped, the kernel is ~5 us shorter on the path to the caller's next launch
    if (!(have_prev && 10 * same >= 7 * B)) { if (threadIdx.x == 0) order[B] = 0; return; }
    for (int i = threadIdx.x; i < B; i += 1024) { const int b = min(max(iters[i], 0) / CONVERGED_INTERVAL, NB - 1); atomicAdd(&cnt[NB - 1 - b], 1); }      // (bucket 0 = longest)
    __syncthreads();
    // exclusive prefix sum over the buckets (two per thread, log-step scan: a serial loop over 512 LDS entries cost 13 us on the path to the status read-back)
    int *src = cnt, *dst = tmp;
    for (int off = 1; off < NB; off <<= 1) {
        for (int b = threadIdx.x; b < NB; b += 1024) dst[b] = src[b] + (b >= off ? src[b - off] : 0);
        __syncthreads();
        int *t = src; src = dst; dst = t;
    }
    for (int b = threadIdx.x; b < NB; b += 1024) dst[b] = b > 0 ? src[b - 1] : 0;          // inclusive -> exclusive
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 1024) { const int b = min(max(iters[i], 0) / CONVERGED_INTERVAL, NB - 1); order[atomicAdd(&dst[NB - 1 - b], 1)] = i; }
    if (threadIdx.x == 0) order[B] = 1;
}
// the order of the NEXT solve is computed off the critical path: behind the status summary (the host is busy with autograd then, the device idle), or at the
// latest in front of the next solve / behind the next adjoint
static int flush_dispatch_order(ce_engine *h, hipStream_t st, const int *sum_status = nullptr, int *sum_out = nullptr) {
    if (!h->order_pending_B) return CE_OK;
    const int B = h->order_pending_B;
    h->order_pending_B = 0;
    if (h->order_cap < B) { hipFree(h->d_order); h->d_order = nullptr; h->order_cap = 0; HIPCHK(hipMalloc(&h->d_order, sizeof(int) * ((size_t)B + 1))); h->order_cap = B; }
    if (h->iters_prev_cap < B) { hipFree(h->d_iters_prev); h->d_iters_prev = nullptr; h->iters_prev_cap = 0; h->iters_prev_B = 0; HIPCHK(hipMalloc(&h->d_iters_prev, sizeof(int) * (size_t)B)); h->iters_prev_cap = B; }
    // (on a stream of the engine's own, ordered by two events, this kernel was measured SLOWER: a cross-queue dependency costs more than the 10 us it would hide -- 2.13 against 2.05 ms
    //  per replayed step, no change on rotating batches)
    hipLaunchKernelGGL(k_dispatch_order, dim3(1), dim3(1024), 0, st, B, h->d_iters2, h->d_order, h->d_iters_prev, h->iters_prev_B == B ? 1 : 0, sum_status, sum_out);
    h->iters_prev_B = B;
    HIPCHK(hipGetLastError());
    h->order_B = B;
    return CE_OK;
}
int ce_set_dispatch_history(ce_handle h, int on) {
    if (!h) { g_err = "null argument"; return CE_E_BADARG; }
    h->dispatch_history = on != 0;
    if (!on) { h->order_B = 0; h->order_pending_B = 0; h->iters_prev_B = 0; }
    return CE_OK;
}

int ce_status_summary(ce_handle h, int B, const int *status, int *summary_host, void *stream) {
    if (!h || B <= 0 || !status || !summary_host) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    if (!h->d_summary) HIPCHK(hipMalloc(&h->d_summary, 8 * 4 * sizeof(int)));
    // Pinned host memory is mapped into the device's address space: the kernel stores the three ints there itself (no copy kernel behind it: two launches
    // less per step of the plugin).  Anything else (the pointer is checked once and remembered) goes through a device slot and an asynchronous copy.
    const uintptr_t page = (uintptr_t)summary_host & ~(uintptr_t)63;      // (the plugin alternates two slots of one 32-byte pinned buffer: remember the 64-byte line)
    if (page != h->summary_host_checked) {
        hipPointerAttribute_t at; char *dp = nullptr;
        const bool mapped = hipPointerGetAttributes(&at, (void *)page) == hipSuccess && at.type == hipMemoryTypeHost &&
                            hipHostGetDevicePointer((void **)&dp, (void *)page, 0) == hipSuccess && dp != nullptr;
        (void)hipGetLastError();
        h->summary_host_checked = page; h->summary_host_dev = mapped ? dp : nullptr;
    }
    if (h->summary_host_dev) {
        int *out = reinterpret_cast<int *>(h->summary_host_dev + ((uintptr_t)summary_host - page));
        // a solve whose dispatch order is still owed and whose status this is: ONE launch does both, the summary (and its ready flag) first
        if (h->order_pending_B == B && status == h->last_status) return flush_dispatch_order(h, (hipStream_t)stream, status, out);
        hipLaunchKernelGGL(k_status_summary, dim3(1), dim3(256), 0, (hipStream_t)stream, B, status, out);
        HIPCHK(hipGetLastError());
        return flush_dispatch_order(h, (hipStream_t)stream, nullptr, nullptr);          // (behind the summary: the host reads the flag while this runs)
    }
    int *slot = h->d_summary + 4 * (h->summary_next++ & 7);      // a few calls may be in flight on the stream before the caller synchronises
    hipLaunchKernelGGL(k_status_summary, dim3(1), dim3(256), 0, (hipStream_t)stream, B, status, slot);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(summary_host, slot, 4 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));      // (three values + the ready flag)
    return flush_dispatch_order(h, (hipStream_t)stream, nullptr, nullptr);
}

int ce_transpose(ce_handle h, int rows, int cols, const double *in, double *out, void *stream) {
    if (!h || !in || !out || rows <= 0 || cols <= 0) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    {
        ProfScope ps(h, 2, st);
        launch_transpose(st, in, out, rows, cols);
    }
    HIPCHK(hipGetLastError());
    return CE_OK;
}

int ce_ca_step(ce_handle h, int B, int lp, double *W, double *UT, double *U, const double *PX, long ld_px, const double *QY, long ld_qy,
               const double *G, const double *PHI, const double *scale, const double *inv_den, const int *active,
               int update_w, int norm_after, double alpha, void *stream) {
    if (!h || B <= 0 || !W || !UT || !U || !PX || !QY || !G || !PHI || !scale || !inv_den || !active) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    const DevT &T = h->T;
    const size_t lds = ((size_t)(T.n + T.m + 1) + 2 * std::max(T.nq, 1) + NW * 8) * 8;
    if (lds > 64 * 1024) { g_err = "constant-A path: instance vectors do not fit LDS"; return CE_E_TOO_LARGE; }
    hipLaunchKernelGGL(k_ca_step, dim3(B), dim3(NT), lds, (hipStream_t)stream, T, lp, W, UT, U, PX, ld_px, QY, ld_qy, G, PHI, scale, inv_den, active, update_w, norm_after, alpha);
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_ca_check(ce_handle h, int B, int lp, int iter, const ce_settings *settings, double *W, const double *UT, const double *U,
                const double *AX, long ld_ax, const double *ATY, long ld_aty, const double *D, const double *E,
                const double *b_hat, const double *c_hat, const double *sigma, const double *nrm_b0, const double *nrm_c0,
                double *scale, double *sum_log, int *n_log, int *last_scale_iter, int *active, int *status, int *iters,
                double *resid, int *rescaled, void *stream) {
    if (!h || B <= 0 || !settings || !W || !UT || !U || !AX || !ATY || !D || !E || !b_hat || !c_hat || !sigma || !scale || !active || !status || !iters || !resid || !rescaled) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_ca_check, dim3(B), dim3(NT), NW * 8 * 8, (hipStream_t)stream, h->T, *settings, lp, iter, W, UT, U, AX, ld_ax, ATY, ld_aty, D, E, b_hat, c_hat,
                       sigma, nrm_b0, nrm_c0, scale, sum_log, n_log, last_scale_iter, active, status, iters, resid, rescaled);
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_ca_psd(ce_handle h, int B, int lp, double *U, const int *active, void *stream) {
    if (!h || B <= 0 || !U || !active) { g_err = "null argument"; return CE_E_BADARG; }
    if (h->T.ns == 0) return CE_OK;
    HIPCHK(hipSetDevice(h->device));
    const size_t lds = (2 * (size_t)h->T.maxs * h->T.maxs + 2 * h->T.maxs + 8 + NW * 8) * 8;
    if (lds > 64 * 1024) { g_err = "PSD order too large for the LDS-resident Jacobi projection"; return CE_E_TOO_LARGE; }
    hipLaunchKernelGGL(k_ca_psd, dim3(B, h->T.ns), dim3(NT), lds, (hipStream_t)stream, h->T, lp, U, active);
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_ca_psd_mfma(ce_handle h, int B, int lp, double *U, double *Vstate, int warm, const int *active, void *stream) {
    if (!h || B <= 0 || !U || !Vstate || !active) { g_err = "null argument"; return CE_E_BADARG; }
    if (h->T.ns == 0) return CE_OK;
    HIPCHK(hipSetDevice(h->device));
    const size_t lds = ((size_t)h->T.maxs * psd_refine_pitch(h->T.maxs) + psd_refine_scratch_doubles(h->T.maxs) + NW * 8) * 8;
    if (lds > 64 * 1024) { g_err = "PSD order too large for the LDS-resident MFMA projection (order <= 39)"; return CE_E_TOO_LARGE; }
    hipLaunchKernelGGL(k_ca_psd_mfma, dim3(B, h->T.ns), dim3(NT), lds, (hipStream_t)stream, h->T, lp, U, Vstate, warm, active);
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_solve_shared_a(ce_handle h, int B, int r, int RP, const double *AdT, const int *drow, const int *srow_col, const double *srow_val,
                      const int *scol_ptr, const int *scol_row, const double *gs, const double *Dv, const double *Ev, const double *b_hat,
                      const double *c_hat, const double *sigma, const double *nrm_b0, const double *nrm_c0, const ce_settings *settings,
                      const double *warm_x, const double *warm_y, const double *warm_s,
                      double *x, double *y, double *s, int *iters, int *status, double *resid, void *stream) {
    if (!h || B <= 0 || !AdT || !drow || !srow_col || !srow_val || !scol_ptr || !scol_row || !gs || !Dv || !Ev || !b_hat || !c_hat || !sigma || !nrm_b0 || !nrm_c0 ||
        !settings || !x || !y || !s || !iters || !status) { g_err = "null argument"; return CE_E_BADARG; }
    const DevT &T = h->T;
    if (r < 0 || r > RP || (RP != 16 && RP != 32 && RP != 64)) { g_err = "shared-A forward kernel: at most 64 dense rows (RP in 16, 32, 64)"; return CE_E_UNSUPPORTED; }
    // 512 threads per instance when the iterates of one instance leave room for a single workgroup per CU anyway (CE_SA_NT=256 / 512 forces);
    // that instantiation also keeps the template's index arrays in LDS when they fit (CE_SA_CIDX=0 disables)
    int nth = 256;
    if (T.ns == 0 && sa_fwd_lds_doubles(T.n, T.m, T.nq, T.ns, T.maxs, RP, 256, T.nep + T.np) * 8 > LDS_LIMIT / 2) nth = 512;
    if (const char *e = getenv("CE_SA_NT")) { const int v = atoi(e); if (v == 256 || (v == 512 && T.ns == 0)) nth = v; }
    size_t lds = sa_fwd_lds_doubles(T.n, T.m, T.nq, T.ns, T.maxs, RP, nth, T.nep + T.np) * 8;
    if (lds > LDS_LIMIT) { g_err = "shared-A forward kernel: the iterates of one instance do not fit LDS"; return CE_E_TOO_LARGE; }
    bool cidx = false;
    if (nth == 512) {
        const size_t with = lds + sa_fwd_cidx_doubles(T.n, T.m, T.nq, r, T.m) * 8;      // (at most m single-entry rows)
        const char *e = getenv("CE_SA_CIDX");
        if (with <= LDS_LIMIT && !(e && atoi(e) == 0)) { cidx = true; lds = with; }
    }
    HIPCHK(hipSetDevice(h->device));
    if (!h->sa_fwd_attr) {
#define SA_ATTR(...) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sa_fwd<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT))
        SA_ATTR(16, 256); SA_ATTR(32, 256); SA_ATTR(64, 256); SA_ATTR(16, 512); SA_ATTR(32, 512); SA_ATTR(64, 512); SA_ATTR(16, 512, true); SA_ATTR(32, 512, true); SA_ATTR(64, 512, true);
        SA_ATTR(16, 512, true, false); SA_ATTR(32, 512, true, false); SA_ATTR(64, 512, true, false);
#undef SA_ATTR
        h->sa_fwd_attr = true;
    }
    if (!h->d_psd_stats && T.ns > 0) { const char *e = getenv("CE_PSD_STATS"); if (e && atoi(e) != 0) { HIPCHK(hipMalloc(&h->d_psd_stats, 16 * sizeof(unsigned long long))); HIPCHK(hipMemset(h->d_psd_stats, 0, 16 * sizeof(unsigned long long))); } }
    int psd_refine = 1; if (const char *e = getenv("CE_PSD_REFINE")) psd_refine = atoi(e) != 0;
    double *aa_ws = nullptr; int aa_w_lds = 0;
    if (settings->acceleration_lookback > 0) {
        const size_t l = (size_t)T.n + T.m + 1, lp = l + (l & 1);
        int rc = ensure(&h->d_aa_ws, &h->aa_ws_bytes, sizeof(double) * (size_t)B * 4 * lp);
        if (rc) return rc;
        aa_ws = h->d_aa_ws;
        // the input of the last iteration in LDS when that does not cost a workgroup per CU (config 4: 68.7 + 3.5 KB, still two per CU)
        const size_t per_cu = nth == 256 ? LDS_LIMIT / 2 : LDS_LIMIT;
        if (lds + lp * 8 <= per_cu || (lds > LDS_LIMIT / 2 && lds + lp * 8 <= LDS_LIMIT)) { aa_w_lds = 1; lds += lp * 8; }
    }
    SaFwd F{r, RP, AdT, drow, srow_col, srow_val, scol_ptr, scol_row, gs, Dv, Ev, h->d_psd_stats, aa_ws, aa_w_lds, psd_refine};
    {
        ProfScope ps(h, 0, (hipStream_t)stream);
#define LAUNCH_SA(NTV, ...) hipLaunchKernelGGL((k_sa_fwd<__VA_ARGS__>), dim3(B), dim3(NTV), lds, (hipStream_t)stream, T, F, *settings, b_hat, c_hat, sigma, nrm_b0, nrm_c0, warm_x, warm_y, warm_s, x, y, s, iters, status, resid)
        if (nth == 256) { if (RP == 16) LAUNCH_SA(256, 16, 256); else if (RP == 32) LAUNCH_SA(256, 32, 256); else LAUNCH_SA(256, 64, 256); }
        else if (!cidx) { if (RP == 16) LAUNCH_SA(512, 16, 512); else if (RP == 32) LAUNCH_SA(512, 32, 512); else LAUNCH_SA(512, 64, 512); }
        else if (T.nep + T.np == 0) { if (RP == 16) LAUNCH_SA(512, 16, 512, true, false); else if (RP == 32) LAUNCH_SA(512, 32, 512, true, false); else LAUNCH_SA(512, 64, 512, true, false); }
        else { if (RP == 16) LAUNCH_SA(512, 16, 512, true); else if (RP == 32) LAUNCH_SA(512, 32, 512, true); else LAUNCH_SA(512, 64, 512, true); }
#undef LAUNCH_SA
    }
    HIPCHK(hipGetLastError());
    return CE_OK;
}
static int vjp_lsqr_launch(ce_handle h, int B, const double *A_vals0, long sA_b, int per_inst, const double *q_vals, long sq_k, long sq_b,
                           const double *x, const double *y, const double *s, const double *dx, const double *dy,
                           double *dA_bm, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, int *lsqr_iters, double atol, double btol, double conlim, int iter_lim, void *stream,
                           const int *sel, int status_or, int *sel_reset) {
    if (!h || B <= 0 || !A_vals0 || !x || !y || !s || !dx || !dy || !dA_bm || !dq_vals) { g_err = "null argument"; return CE_E_BADARG; }
    const DevT &T = h->T;
    // products through the singleton / dense-row split when the template has one (CE_SA_SPLIT=0: CSR / CSC products)
    int RP = h->sp_RP;
    if (const char *e = getenv("CE_SA_SPLIT")) { if (atoi(e) == 0) RP = 0; }
    if (per_inst) RP = 0;      // the split's dense rows are ONE matrix (instance 0's values); per-instance values go through the CSR / CSC products
    const int lsmr = (h->lsqr_variant == 1 && !sel) ? 1 : 0;      // (the re-solve list of ce_vjp stays diffcp's default, LSQR)
    if (RP > 0 && sa_lsqr_lds_doubles(T.n, T.m, T.nq, T.ns, T.maxs, RP, h->psd_first, T.nep + T.np, lsmr) * 8 > LDS_LIMIT) RP = 0;
    size_t lds = sa_lsqr_lds_doubles(T.n, T.m, T.nq, T.ns, T.maxs, RP, h->psd_first, T.nep + T.np, lsmr) * 8;
    if (lds > LDS_LIMIT) { g_err = "shared-A adjoint kernel: the LSQR vectors of one instance do not fit LDS"; return CE_E_TOO_LARGE; }
    // per-instance A: staged dense in LDS when it fits behind the vectors with three workgroups per CU to spare (CE_LSQR_A_LDS=0 disables)
    int a_lds = 0;
    if (per_inst && RP == 0) {
        const char *e = getenv("CE_LSQR_A_LDS");
        const size_t with = lds + 8 + sizeof(double) * (size_t)T.m * T.n;
        if (!(e && atoi(e) == 0) && with <= LDS_LIMIT / 3) { a_lds = 1; lds = with; }
    }
    HIPCHK(hipSetDevice(h->device));
    if (!h->sa_lsqr_attr) {
#define SA_ATTR(...) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sa_lsqr<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT))
        SA_ATTR(0); SA_ATTR(16); SA_ATTR(32); SA_ATTR(64); SA_ATTR(0, false, false); SA_ATTR(16, false, false); SA_ATTR(32, false, false); SA_ATTR(64, false, false);
        SA_ATTR(16, true, false); SA_ATTR(32, true, false); SA_ATTR(64, true, false);
        SA_ATTR(0, true, true, true); SA_ATTR(16, true, true, true); SA_ATTR(32, true, true, true); SA_ATTR(64, true, true, true);
#undef SA_ATTR
        h->sa_lsqr_attr = true;
    }
    // The streaming passes read c_j of their instance with every row of the matrix.  In the boundary's layout (q_eval (n + 1, B): consecutive j are B doubles apart) each of
    // those loads is a line of its own, and the lines of all resident instances (config 5: 768 x 501 x 128 B) live in the memory-side cache, not in L2: the pass waited for
    // THEM, not for the matrix.  One transpose per call gives every instance a contiguous c (4 KB, L2-resident for the whole solve).  Not for the re-solve list
    // (a handful of instances; the launch sits on the metric configuration's hot path).
    if (q_vals && !sel && sq_b == 1 && sq_k == (long)B && B > 1) {
        int rc = ensure(&h->d_qT, &h->qT_bytes, sizeof(double) * (size_t)B * (T.n + 1));
        if (rc) return rc;
        launch_transpose((hipStream_t)stream, q_vals, h->d_qT, T.n + 1, B);
        q_vals = h->d_qT; sq_k = 1; sq_b = T.n + 1;
    }
    SaStruct S{h->d_csc_ptr, h->d_rowidx, h->d_csr_ptr, h->d_csr_col, h->d_csr_src, T.nnzA, h->d_bpos};
    SaSplit F{h->sp_r, RP, h->d_sp_AdT, h->d_sp_drow, h->d_sp_srow_col, h->d_sp_sval, h->d_sp_scol_ptr, h->d_sp_scol_row, h->d_sp_rowslot, h->d_sp_sing_i, h->d_sp_sing_v};
    if (RP > 0) {      // the values may differ between calls: refill A_d^T / singleton values from this call's A (n RP + m doubles)
        HIPCHK(hipMemsetAsync(h->d_sp_AdT, 0, sizeof(double) * (size_t)T.n * RP, (hipStream_t)stream));
        HIPCHK(hipMemsetAsync(h->d_sp_sval, 0, sizeof(double) * T.m, (hipStream_t)stream));
        HIPCHK(hipMemsetAsync(h->d_sp_sing_v, 0, sizeof(double) * T.n, (hipStream_t)stream));
        if (T.nnzA > 0) hipLaunchKernelGGL(k_sa_fill_split, dim3((T.nnzA + 255) / 256), dim3(256), 0, (hipStream_t)stream, T.nnzA, RP, h->d_rowidx, h->d_colidx, h->d_sp_rowslot, A_vals0, h->d_sp_AdT, h->d_sp_sval, h->d_sp_sing_i, h->d_sp_sing_v);
    }
    // Several instances per workgroup share the stream over A_d^T (ce_shared_a_mi.h) where the template allows it: plain cones, the split's products, the solution
    // in the owners' registers, no re-solve list, and enough instances to fill the device either way.  CE_SA_LSQR_NI=1 keeps one instance per workgroup (A/B), 2 / 3 force.
    int ni = 0;
    if (RP > 0 && !per_inst && !sel && !lsmr && T.ns == 0 && T.nep + T.np == 0 && T.n <= SAMI_EL * 256 && T.m <= SAMI_EL * 256) {
        ni = 0;      // (measured slower than one instance per workgroup at config 5: profiles/r06/n_*; opt-in)
        if (const char *e = getenv("CE_SA_LSQR_NI")) { const int v = atoi(e); ni = (v == 2 || v == 3) ? v : 0; }
        while (ni >= 2 && sa_lsqr_mi_lds_doubles(T.n, T.m, T.nq, RP, ni) * 8 > LDS_LIMIT) ni--;
        if (ni < 2) ni = 0;
    }
    if (ni) {
        if (!h->sa_lsqr_mi_attr) {
#define SA_ATTR(RPV, NIV) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sa_lsqr_mi<RPV, NIV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT))
            SA_ATTR(16, 2); SA_ATTR(32, 2); SA_ATTR(64, 2); SA_ATTR(16, 3); SA_ATTR(32, 3); SA_ATTR(64, 3);
#undef SA_ATTR
            h->sa_lsqr_mi_attr = true;
        }
        const size_t lds_mi = sa_lsqr_mi_lds_doubles(T.n, T.m, T.nq, RP, ni) * 8;
        ProfScope ps(h, 1, (hipStream_t)stream);
#define LAUNCH_MI(RPV, NIV) hipLaunchKernelGGL((k_sa_lsqr_mi<RPV, NIV>), dim3((B + NIV - 1) / NIV), dim3(NIV * 256), lds_mi, (hipStream_t)stream, T, S, F, A_vals0, sA_b, q_vals, sq_k, sq_b, x, y, s, dx, dy, dA_bm, dq_vals, sdq_k, sdq_b, adj_status, lsqr_iters, atol, btol, conlim, iter_lim > 0 ? iter_lim : 2 * (T.n + T.m + 1), B)
        if (ni == 2) { if (RP == 16) LAUNCH_MI(16, 2); else if (RP == 32) LAUNCH_MI(32, 2); else LAUNCH_MI(64, 2); }
        else { if (RP == 16) LAUNCH_MI(16, 3); else if (RP == 32) LAUNCH_MI(32, 3); else LAUNCH_MI(64, 3); }
#undef LAUNCH_MI
        HIPCHK(hipGetLastError());
        return CE_OK;
    }
    if (const char *e = getenv("CE_SA_LSQR_PADLDS")) { const size_t want = (size_t)atoi(e) * 1024; if (want > lds && want <= LDS_LIMIT) lds = want; }      // (residency experiment: workgroups per CU)
    {
        ProfScope ps(h, 1, (hipStream_t)stream);
#define LAUNCH_SAL(...) hipLaunchKernelGGL((k_sa_lsqr<__VA_ARGS__>), dim3(B), dim3(NT), lds, (hipStream_t)stream, T, S, F, A_vals0, sA_b, per_inst, q_vals, sq_k, sq_b, x, y, s, dx, dy, dA_bm, dq_vals, sdq_k, sdq_b, adj_status, lsqr_iters, atol, btol, conlim, iter_lim > 0 ? iter_lim : 2 * (T.n + T.m + 1), sel, status_or, a_lds, sel_reset)
        // plain cones / PSD without triples: instantiations without the other cones' code (CE_SA_LSQR_SPEC=0: the general kernel)
        const bool tri = T.nep + T.np > 0, psd = T.ns > 0;
        int spec = 1; if (const char *e = getenv("CE_SA_LSQR_SPEC")) spec = atoi(e);
        if (lsmr) { if (RP == 0) LAUNCH_SAL(0, true, true, true); else if (RP == 16) LAUNCH_SAL(16, true, true, true); else if (RP == 32) LAUNCH_SAL(32, true, true, true); else LAUNCH_SAL(64, true, true, true); }
        else if (spec && !tri && !psd) { if (RP == 0) LAUNCH_SAL(0, false, false); else if (RP == 16) LAUNCH_SAL(16, false, false); else if (RP == 32) LAUNCH_SAL(32, false, false); else LAUNCH_SAL(64, false, false); }
        else if (spec && !tri && RP > 0) { if (RP == 16) LAUNCH_SAL(16, true, false); else if (RP == 32) LAUNCH_SAL(32, true, false); else LAUNCH_SAL(64, true, false); }
        else if (RP == 0) LAUNCH_SAL(0); else if (RP == 16) LAUNCH_SAL(16); else if (RP == 32) LAUNCH_SAL(32); else LAUNCH_SAL(64);
#undef LAUNCH_SAL
    }
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_vjp_shared_a(ce_handle h, int B, const double *A_vals0, long sA_b, const double *q_vals, long sq_k, long sq_b,
                    const double *x, const double *y, const double *s, const double *dx, const double *dy,
                    double *dA_bm, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, int *lsqr_iters, double atol, double btol, double conlim, int iter_lim, void *stream) {
    return vjp_lsqr_launch(h, B, A_vals0, sA_b, 0, q_vals, sq_k, sq_b, x, y, s, dx, dy, dA_bm, dq_vals, sdq_k, sdq_b, adj_status, lsqr_iters, atol, btol, conlim, iter_lim, stream);
}
int ce_vjp_lsqr(ce_handle h, int B, const double *A_vals_bm, long sA_b, const double *q_vals, long sq_k, long sq_b,
                const double *x, const double *y, const double *s, const double *dx, const double *dy,
                double *dA_bm, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, int *lsqr_iters, double atol, double btol, double conlim, int iter_lim, void *stream) {
    if (sA_b == 0 && B > 1) { g_err = "ce_vjp_lsqr: per-instance values need a batch stride"; return CE_E_BADARG; }
    return vjp_lsqr_launch(h, B, A_vals_bm, sA_b, 1, q_vals, sq_k, sq_b, x, y, s, dx, dy, dA_bm, dq_vals, sdq_k, sdq_b, adj_status, lsqr_iters, atol, btol, conlim, iter_lim, stream);
}
int ce_ca_triples(ce_handle h, int B, int lp, double *U, double *roots, const int *active, void *stream) {
    if (!h || B <= 0 || !U || !roots || !active) { g_err = "null argument"; return CE_E_BADARG; }
    const int ntri = h->T.nep + h->T.np;
    if (ntri == 0) return CE_OK;
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_ca_triples, dim3(((size_t)B * ntri + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, h->T, lp, B, U, roots, active);
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_ca_triple_jac(ce_handle h, int B, const double *v, long ld_v, double *J, void *stream) {
    if (!h || B <= 0 || !v || !J) { g_err = "null argument"; return CE_E_BADARG; }
    const int ntri = h->T.nep + h->T.np;
    if (ntri == 0) return CE_OK;
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_ca_triple_jac, dim3(((size_t)B * ntri + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, h->T, B, v, ld_v, J);
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_ca_update(ce_handle h, int B, int lp, double *W, const double *UT, const double *U, const int *active, int norm_after, double alpha, void *stream) {
    if (!h || B <= 0 || !W || !UT || !U || !active) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_ca_update, dim3(B), dim3(NT), 0, (hipStream_t)stream, h->T.n + h->T.m + 1, lp, W, UT, U, active, norm_after, alpha);
    HIPCHK(hipGetLastError());
    return CE_OK;
}
int ce_ca_finish(ce_handle h, int B, int lp, int max_iters, const double *W, const double *UT, const double *U, const double *D,
                 const double *E, const double *b_hat, const double *c_hat, const double *sigma, const double *scale,
                 const int *active, int *status, int *iters, double *x, double *y, double *s, void *stream) {
    if (!h || B <= 0 || !W || !UT || !U || !x || !y || !s) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_ca_finish, dim3(B), dim3(NT), NW * 8 * 8, (hipStream_t)stream, h->T, lp, max_iters, W, UT, U, D, E, b_hat, c_hat, sigma, scale, active, status, iters, x, y, s);
    HIPCHK(hipGetLastError());
    return CE_OK;
}

extern "C++" {
template <bool ACC>
static int parammap_launch(int device, int B, int rows, int cols, const int *indptr, const int *indices, const double *vals,
                           const double *P, long ld_p, double *out, long ld_out, void *stream) {
    if (B <= 0 || rows <= 0 || !indptr || !P || !out) { g_err = "null argument"; return CE_E_BADARG; }   // indices / vals may be null for an all-zero map
    HIPCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)cols * sizeof(double);
    if (cols > 0 && lds <= 64 * 1024 && B >= 256) {          // source row fits LDS (2+ workgroups per CU) and the batch fills the chip
        static std::atomic<bool> attr_done[2][64];                 // per device (the attribute is per device); re-setting it is harmless, so a benign race at most repeats the call
        const int dslot = device & 63;
        if (!attr_done[ACC][dslot].load(std::memory_order_acquire)) {
            HIPCHK(hipFuncSetAttribute((const void *)k_parammap_lds<ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            attr_done[ACC][dslot].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(k_parammap_lds<ACC>, dim3(B), dim3(512), lds, st, rows, cols, indptr, indices, vals, P, ld_p, out, ld_out);
    } else if (B >= 1024) {
        dim3 grid((rows + 255) / 256, (B + 3) / 4);
        hipLaunchKernelGGL((k_parammap<4, ACC>), grid, dim3(256), 0, st, rows, B, indptr, indices, vals, P, ld_p, out, ld_out);
    } else {
        dim3 grid((rows + 255) / 256, B);
        hipLaunchKernelGGL((k_parammap<1, ACC>), grid, dim3(256), 0, st, rows, B, indptr, indices, vals, P, ld_p, out, ld_out);
    }
    HIPCHK(hipGetLastError());
    return CE_OK;
}
}  // extern "C++"
int ce_parammap_apply(int device, int B, int rows, const int *indptr, const int *indices, const double *vals,
                      const double *P, long ld_p, double *out, long ld_out, void *stream) {
    return parammap_launch<false>(device, B, rows, 0, indptr, indices, vals, P, ld_p, out, ld_out, stream);
}
int ce_parammap_apply2(int device, int B, int rows, int cols, int accumulate, const int *indptr, const int *indices, const double *vals,
                       const double *P, long ld_p, double *out, long ld_out, void *stream) {
    return accumulate ? parammap_launch<true>(device, B, rows, cols, indptr, indices, vals, P, ld_p, out, ld_out, stream)
                      : parammap_launch<false>(device, B, rows, cols, indptr, indices, vals, P, ld_p, out, ld_out, stream);
}

int ce_set_profiling(ce_handle h, int enable) {
    if (!h) return CE_E_BADARG;
    h->prof = enable == 1 ? 7 : (enable > 1 ? (enable >> 1) & 7 : 0);      // 1: every kind; 2 / 4 / 8 (or sums): forward / adjoint / layout launches only
    if (h->prof) { while (h->ev_pool.size() < 2048) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; h->ev_pool.push_back(e); } }      // (created HERE, not in front of the timed launches)
    return CE_OK;
}
int ce_reset_profile(ce_handle h) {
    if (!h) return CE_E_BADARG;
    for (auto &v : h->ev) { for (auto &p : v) { h->ev_pool.push_back(p.first); h->ev_pool.push_back(p.second); } v.clear(); }      // (kept for the next scopes)
    return CE_OK;
}
int ce_get_profile(ce_handle h, int which, double *mean_ms, int *launches) {
    if (!h || which < 0 || which > 2) return CE_E_BADARG;
    double tot = 0; int nl = 0;
    for (auto &p : h->ev[which]) { HIPCHK(hipEventSynchronize(p.second)); float ms = 0; HIPCHK(hipEventElapsedTime(&ms, p.first, p.second)); tot += ms; nl++; }
    if (mean_ms) *mean_ms = nl ? tot / nl : 0.0;
    if (launches) *launches = nl;
    return CE_OK;
}
int ce_get_launch_info(ce_handle h, int *fl, int *bl, int *fm, int *bm) {
    if (!h) return CE_E_BADARG;
    if (fl) *fl = (int)h->fwd_lds; if (bl) *bl = (int)h->bwd_lds; if (fm) *fm = h->fwd_mode; if (bm) *bm = h->bwd_mode;
    return CE_OK;
}

}  // extern "C"
