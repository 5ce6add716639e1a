/*
 * cone_engine.h -- C ABI of the MI355X-native batched cone-program solve + differentiate engine.
 *
 * This is the drop-in boundary underneath the Python solver plugin
 * (cvxpylayers_amd/interfaces/mi355_if.py), which mirrors the reference plugin
 * cvxpylayers/interfaces/diffcp_if.py.  The reference has NO FFI for this path (its arithmetic is
 * in the third-party diffcp/SCS packages, reached through Python calls), so every entry point
 * below cites the reference *call site* whose work it replaces:
 *
 *   ce_create   <- DIFFCP_ctx.__init__            (diffcp_if.py:105-120; interfaces/__init__.py:26-33)
 *                  : keeps the CSC structure of the augmented matrix [A_cvx | b_cvx] and the cone dims.
 *   ce_solve    <- _build_diffcp_matrices + diffcp.solve_and_derivative_batch / solve_only_batch
 *                  (diffcp_if.py:46-70, 365-372): cuts (A,b,c) out of A_eval / q_eval, A = -A_cvx,
 *                  solves every instance, returns primal (B,n), dual (B,m) (+ slack, status).
 *   ce_vjp      <- _compute_gradients -> adj_batch(dxs, dys, dss=0)   (diffcp_if.py:73-96, 385-403):
 *                  returns d/dA_eval = [-dA.data, db[b_idx]] and d/dq_eval = [dc, 0].
 *   ce_destroy  <- (garbage collection of DIFFCP_ctx)
 *
 * All data pointers are CALLER-OWNED DEVICE memory (fp64 / int32), valid on the device the handle
 * was created for; `stream` is a hipStream_t (NULL = default stream).  Calls only enqueue work:
 * there is no hidden host synchronisation.  The handle owns workspace only and is not thread-safe
 * (reference contract: one layer, one caller thread -- moreau_if.py:14-15).
 * Return value: 0 on success, negative CE_E_* otherwise; per-instance solver status is written to
 * `status[]` with SCS's codes (1 solved, 2 solved/inaccurate, -1 unbounded, -2 infeasible,
 * -6/-7 inaccurate certificates, -4 failed).
 */
#ifndef CONE_ENGINE_H
#define CONE_ENGINE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ce_engine *ce_handle;

enum {
    CE_OK = 0,
    CE_E_BADARG = -1,        /* inconsistent template / null pointer */
    CE_E_UNSUPPORTED = -2,   /* template / argument combination not implemented on the selected device path (ce_last_error says which) */
    CE_E_TOO_LARGE = -3,     /* instance does not fit the implemented residency modes */
    CE_E_HIP = -4,           /* a HIP runtime call failed (ce_last_error has the string) */
    CE_E_STATE = -5          /* ce_vjp called without the retained forward state it was asked to reuse */
};

/* One-time, host-side description of the canonical template (what CVXPY's ParamConeProg gives a plugin). */
typedef struct {
    int n;                 /* canonical variables */
    int m;                 /* cone rows */
    int nnz_aug;           /* structural non-zeros of the augmented m x (n+1) matrix [A_cvx | b_cvx] */
    const int *indices;    /* [nnz_aug] CSC row indices   (HOST memory) */
    const int *indptr;     /* [n+2]     CSC column starts (HOST memory); column n holds b */
    int z;                 /* zero-cone rows    (dims_to_solver_dict key "z") */
    int l;                 /* nonnegative rows  ("l") */
    int nq;                /* number of second-order cones */
    const int *q;          /* [nq] SOC dimensions ("q"), layout (t, x) */
    int ns;                /* number of PSD cones ("s") */
    const int *s;          /* [ns] PSD orders; svec lower-tri column-major, sqrt(2) off-diagonals */
    int nep;               /* exponential cones ("ep"): triples (x, y, z), y exp(x/y) <= z, after the PSD blocks (SCS row order z,l,q,s,ep,p) */
    int np;                /* 3-d power cones ("p"): triples (x, y, z), x^a y^(1-a) >= |z|, after the exponential cones */
    const double *p;       /* [np] exponents a in (0, 1); a negative entry -a is the DUAL power cone of exponent a (SCS convention)  (HOST memory) */
    /* optional quadratic objective 1/2 x^T P x (param_prob.reduced_P.problem_data_index, interfaces/__init__.py:27): CSC structure
     * of the n x n matrix P, either one triangle or structurally symmetric; nnz_p = 0: linear objective */
    int nnz_p;
    const int *p_indices;  /* [nnz_p] row indices (HOST memory) */
    const int *p_indptr;   /* [n+1] */
} ce_template;

/* Solver settings; names follow SCS / diffcp keyword arguments (diffcp maps eps -> eps_abs, eps_rel). */
typedef struct {
    double eps_abs, eps_rel, eps_infeas;
    double alpha;          /* over-relaxation, 1.5 */
    double rho_x;          /* 1e-6 */
    double scale;          /* initial dual scale, 0.1 */
    int max_iters;         /* 100000 */
    int normalize;         /* Ruiz + l2 equilibration, 1 */
    int adaptive_scale;    /* 1 */
    int warm_start;       /* != 0: x, y, s hold an initial primal / dual / slack point on entry (SCS warm start: u = (x, y, 1), v = (0, s, 0));
                             instances whose point is not finite start cold.  The reference's DIFFCP plugin exposes this as
                             diffcp's `warm_starts` solve argument; MOREAU as `warm_start` (torch/cvxpylayer.py:464-487) */
    int acceleration_lookback;   /* SCS name; default 10 like SCS (diffcp forwards SCS's defaults, diffcp_if.py:356-367).  0: plain iteration.
                                    > 0: type-I Anderson acceleration of the iteration map with a ONE-pair secant history whatever the
                                    value (profiles/r02/aa_memory.json: iteration counts within 2.5 % of lookback 10 on the BASELINE
                                    configurations) and SCS's residual safeguard.  Honoured where ce_acceleration_available() /
                                    the shared-A kernels implement it; other paths iterate plainly (the Python plugin warns once). */
    int acceleration_interval;   /* applied every this many iterations (SCS default 10) */
} ce_settings;

void ce_default_settings(ce_settings *s);

/* ABI guard.  The structs above carry no size field, so a binding compiled / written against an older header would hand the
 * library short structs.  Bindings must check  ce_abi_version() == CE_ABI_VERSION  and  ce_struct_size(which) == sizeof(their
 * struct)  (which: 0 ce_template, 1 ce_settings) once at load time and refuse to continue otherwise (cvxpylayers_amd/_lib.py
 * does; tests/test_cabi.py checks the stub printed in INTEGRATION.md the same way).  CE_ABI_VERSION is bumped whenever a struct
 * layout, an entry point's signature or the meaning of an argument changes (11: ce_vjp re-solves rank-deficient instances by LSQR when q_vals is given, adj_status is a bit field, ce_set_adjoint_resolve added; 10: ce_vjp_lsqr added; 9: ce_vjp_shared_a takes sA_b and q_vals -- the adjoint system gains diffcp's tau row and column --, its iter_lim default is diffcp's 2 (n + m + 1); 8: ce_set_dispatch_history added, ce_status_summary writes a fourth "ready" int; 7: ce_status_summary added; 6: ce_default_settings = SCS defaults incl. acceleration_lookback 10, ce_acceleration_available). */
#define CE_ABI_VERSION 11
int ce_abi_version(void);
int ce_struct_size(int which);
/* The iterative adjoint solver of ce_vjp_shared_a / ce_vjp_lsqr (the calls that solve EVERY instance iteratively): 0 = LSQR (Paige & Saunders; diffcp's default
 * mode, diffcp_if.py:86), 1 = LSMR (Fong & Saunders; diffcp's mode="lsmr": same operator, same atol / btol / conlim / iter_lim arguments, scipy.sparse.linalg.lsmr's
 * recurrences and stopping tests).  Engine state, default 0; the re-solve of rank-deficient instances inside ce_vjp always runs LSQR.  CE_E_BADARG for other values. */
int ce_set_lsqr_variant(ce_handle h, int variant);
/* 1 when ce_solve / ce_solve_qp on this engine honour ce_settings.acceleration_lookback > 0 (second-generation forward kernel with
 * room for its five extra vectors in LDS; the first-generation and size-generic kernels, which keep them in global memory), else 0: the
 * request is ignored on that engine (plain iteration). */
int ce_acceleration_available(ce_handle h);

int ce_create(const ce_template *tpl, int device, ce_handle *out);
int ce_destroy(ce_handle h);
const char *ce_last_error(void);

/*
 * Forward.  Element (k, i) of the reference's A_eval (nnz_aug x B) is read at A_vals[k*sA_k + i*sA_b];
 * element (k, i) of q_eval ((n+1) x B) at q_vals[k*sq_k + i*sq_b] (strides in elements), so both the
 * reference layout (batch-minor: sA_k = B, sA_b = 1) and the engine-native batch-major layout
 * (sA_k = 1, sA_b = nnz_aug) are accepted; the former costs one transpose pass.
 * Outputs (row-major, contiguous): x [B][n], y [B][m], s [B][m]; iters [B], status [B] int32;
 * resid [B][3] = (primal residual, dual residual, gap) or NULL.
 * The batch-major copy of A_vals and (x,y,s) pointers are retained for ce_vjp(reuse_forward=1).
 */
int ce_solve(ce_handle h, int B,
             const double *A_vals, long sA_k, long sA_b,
             const double *q_vals, long sq_k, long sq_b,
             const ce_settings *settings,
             double *x, double *y, double *s, int *iters, int *status, double *resid,
             void *stream);

/*
 * Backward (vector-Jacobian product), diffcp adjoint with ds = 0 (diffcp_if.py:84).
 * dx [B][n], dy [B][m] contiguous.  Gradients are written in the *boundary* convention
 * dA_vals (k, i) at [k*sdA_k + i*sdA_b] = [-dA.data, db[b_idx]],  dq_vals (k,i) at [k*sdq_k + i*sdq_b] = [dc, 0].
 * If A_vals == NULL the batch-major copy retained by the last ce_solve on this handle is used.
 * The system M^T r = dz is reduced cone block by cone block and eliminated directly with r_tau pinned to 0 (the same gradients as diffcp's LSQR wherever it is
 * regular).  Where it is RANK DEFICIENT (redundant equality rows, degenerate active sets) the elimination only has a basic solution to offer while diffcp's LSQR
 * (diffcp_if.py:86 -> adj_batch) returns the minimum-norm one: with q_vals given (the call's q_eval, as ce_solve; c and b enter diffcp's full (n + m + 1)
 * system through it) such instances -- and instances whose active set exceeds the kernel's tile -- are listed on the device by the elimination kernel and
 * re-solved behind it by the LSQR kernel of ce_vjp_lsqr (one more launch whose workgroups walk the list; no host round trip), so the DEFAULT answer is
 * diffcp's on every instance.  q_vals == NULL, ce_set_adjoint_resolve(h, 0, ...) or a quadratic objective: no re-solve.
 * adj_status [B] (or NULL), bit field: 1 = LSQR re-solve stopped at its iteration limit; 2 = active set larger than the direct solve holds and no re-solve
 * (zero gradient); 4 = rank-deficient system (free variables set to zero by the elimination); 8 = gradients replaced by the LSQR re-solve.
 */
int ce_vjp(ce_handle h, int B,
           const double *A_vals, long sA_k, long sA_b,
           const double *q_vals, long sq_k, long sq_b,
           const double *x, const double *y, const double *s,
           const double *dx, const double *dy,
           double *dA_vals, long sdA_k, long sdA_b,
           double *dq_vals, long sdq_k, long sdq_b,
           int *adj_status,
           void *stream);

/* Layout helper: out (cols x rows, row-major) = transpose of in (rows x cols, row-major), fp64, caller-owned
 * device buffers.  Used by the plugin to turn the reference's batch-minor A_eval (nnz_aug x B) into the
 * engine-native batch-major (B x nnz_aug) once, into a tensor it keeps for backward. */
int ce_transpose(ce_handle h, int rows, int cols, const double *in, double *out, void *stream);

/* Enqueues, behind whatever produced v[] (B int32 on the device: the status of a forward call or the adj_status of a backward call), the reduction
 * summary_host[0] = min_i v[i], [1] = #{i: v[i] == 2}, [2] = #{i: (v[i] & 3) != 0} and its copy to summary_host (FOUR ints of PINNED host memory: the
 * fourth is set to 1 after the three values are visible).  The caller synchronises the stream (or an event) before reading it -- or clears
 * summary_host[3] before the call and polls it.  This is the only host <- device traffic a forward call of the Python plugin needs in order to
 * honour the reference's contract that a failed instance raises SolverError from forward() (diffcp_if.py:365-372 raises inside the call): 12 bytes instead
 * of the status vector.  When summary_host is pinned memory mapped into the device's address space (hipHostMalloc / torch's pin_memory) the kernel stores
 * there directly; any other host pointer goes through one of EIGHT rotating device slots and an asynchronous copy: at most eight such calls may be in flight
 * on the stream between two synchronisations of the caller.
 * LIFETIME: the engine remembers, per 64-byte line of host memory, whether the line is mapped and its device alias (one hipPointerGetAttributes per line,
 * not per call).  A buffer handed to this function must therefore stay allocated AND pinned for as long as the engine lives (or until another line has
 * been passed in between): freeing it and re-using the address for a different allocation leaves the engine with a stale alias. */
int ce_status_summary(ce_handle h, int B, const int *status, int *summary_host, void *stream);

/*
 * Quadratic objective (templates created with nnz_p > 0): ce_solve / ce_vjp with the values of P  <- P_eval of the QP-capable
 * plugins (moreau_if.py:399-404; _quad_form_dpp.py:32).  P_vals is BATCH-MAJOR (B, nnz_p) contiguous device memory (the P entries
 * in the template's structure; for a structurally symmetric P the values must be symmetric).  SCS 3's QP embedding: P joins the
 * reduced KKT matrix, tau-tilde is the positive root of a quadratic, dual residual / gap / objective include P.  ce_vjp_qp also
 * returns dP_vals (B, nnz_p) batch-major (the gradient of a one-triangle entry is that of both matrix entries it stands for).
 * ce_qp_native(h): 1 if this template's quadratic objective runs inside the kernels (fits the register-tiled variants, no
 * PSD / exponential / power cones); otherwise callers reduce it to an epigraph SOC themselves (mi355_if.py QuadEpigraph).
 */
int ce_qp_native(ce_handle h);
int ce_solve_qp(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *q_vals, long sq_k, long sq_b,
                const double *P_vals, const ce_settings *settings, double *x, double *y, double *s, int *iters, int *status, double *resid, void *stream);
int ce_vjp_qp(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *P_vals,
              const double *x, const double *y, const double *s, const double *dx, const double *dy,
              double *dA_vals, long sdA_k, long sdA_b, double *dq_vals, long sdq_k, long sdq_b, double *dP_vals, int *adj_status, void *stream);

/*
 * Parameter-map evaluation on the device, batch-major  <- CvxpyLayer.forward's  A_eval = A_map @ p_stack,
 * q_eval = q_map @ p_stack  (torch/cvxpylayer.py:433-451; _ScipySparseMatmul :12-37) and, called with the transposed map,
 * their backward (grad_p_stack = map^T @ grad, :32-37).  The map is CSR (rows x cols; indptr/indices/vals are DEVICE arrays):
 *     out[b*ld_out + r] = sum_{t = indptr[r] .. indptr[r+1]-1} vals[t] * P[b*ld_p + indices[t]]      b < B, r < rows
 * i.e. parameters and results are (B, .) row-major, so the result is already in the engine-native batch-major layout and the
 * reference's p_stack transpose and the layout pass of ce_solve both disappear.  No handle: any device pointer set works.
 */
int ce_parammap_apply(int device, int B, int rows, const int *indptr, const int *indices, const double *vals,
                      const double *P, long ld_p, double *out, long ld_out, void *stream);
/* Same map, with the number of source columns known (cols > 0: when a source row fits LDS it is staged there once per instance,
 * so maps that transpose a matrix parameter stay one pass over HBM) and optional accumulation: accumulate != 0 computes
 * out[b, r] += ... and leaves rows without entries untouched -- the sum of the A-map and q-map gradients into one p_stack
 * gradient (autograd's add in the reference, torch/cvxpylayer.py:32-37 called once per map). */
int ce_parammap_apply2(int device, int B, int rows, int cols, int accumulate, const int *indptr, const int *indices, const double *vals,
                       const double *P, long ld_p, double *out, long ld_out, void *stream);

/*
 * Constant-A path (A batch-invariant; only b, c vary): the matrix products of the iteration are batch GEMMs done by the
 * caller with rocBLAS (cvxpylayers_amd/interfaces/const_a.py); these entry points are the per-instance elementwise part.
 * All vectors are (B, lp) row-major with layout (x[n] | y[m] | tau); per-instance scalars are arrays of length B.
 *   ce_ca_step   : tau-tilde, u-tilde, cone projection, (optionally) relaxed update + renormalisation of w
 *   ce_ca_check  : termination test / certificates / adaptive scale of a check iteration, then that iteration's update
 *   ce_ca_psd    : in-place projection of the PSD blocks of U (the step kernel leaves the cone input there)
 *   ce_ca_finish : classification of unfinished instances and un-normalised write-back of x (B,n), y (B,m), s (B,m)
 * They replace the same steps of diffcp.solve_and_derivative_batch -> SCS (diffcp_if.py:365-372) as ce_solve does.
 */
int ce_ca_step(ce_handle h, int B, int lp, double *W, double *UT, double *U, const double *PX, long ld_px, const double *QY, long ld_qy,
               const double *G, const double *PHI, const double *scale, const double *inv_den, const int *active,
               int update_w, int norm_after, double alpha, void *stream);
int ce_ca_check(ce_handle h, int B, int lp, int iter, const ce_settings *settings, double *W, const double *UT, const double *U,
                const double *AX, long ld_ax, const double *ATY, long ld_aty, const double *D, const double *E,
                const double *b_hat, const double *c_hat, const double *sigma, const double *nrm_b0, const double *nrm_c0,
                double *scale, double *sum_log, int *n_log, int *last_scale_iter, int *active, int *status, int *iters,
                double *resid, int *rescaled, void *stream);
int ce_ca_psd(ce_handle h, int B, int lp, double *U, const int *active, void *stream);
/* The same projection on the matrix cores (v_mfma_f64_16x16x4_f64), warm-started from the eigenvectors of the previous call: Vstate
 * (B, ns, maxs * maxs; row-major k x k per block) is caller-owned state, warm = 0: no previous call.  Warm calls REFINE the previous
 * decomposition (R = I - V^T V, D = V^T S V, first-order correction V <- V + V E, quadratically convergent, orthogonality self-correcting:
 * no periodic restart needed) and fall back to Jacobi sweeps when a correction would leave the basin of the first-order step.  warm: 0 cold, 1 warm;
 * bit 1 (warm = 3) is a debugging switch: warm-started Jacobi sweeps only, no refinement (scripts/psd_refine_debug.py).  PSD orders <= 39. */
int ce_ca_psd_mfma(ce_handle h, int B, int lp, double *U, double *Vstate, int warm, const int *active, void *stream);
/* Exponential / power cone triples of the cone input U (B, lp) projected in place (after ce_ca_step, like ce_ca_psd); roots
 * (B, nep + np) is caller-owned state: each cone's root of the previous iteration (zero-initialised). */
int ce_ca_triples(ce_handle h, int B, int lp, double *U, double *roots, const int *active, void *stream);
/* J (B, nep + np, 9): the 3x3 Jacobians of that projection at v = y - s (rows of v have pitch ld_v), used by the batched-LSQR
 * adjoint of the constant-A path in place of diffcp's per-instance dpi operator (diffcp_if.py:86 -> adj_batch). */
int ce_ca_triple_jac(ce_handle h, int B, const double *v, long ld_v, double *J, void *stream);
/* w += alpha (u - ut) (+ renormalisation) as its own launch, for templates whose PSD blocks are projected after ce_ca_step */
int ce_ca_update(ce_handle h, int B, int lp, double *W, const double *UT, const double *U, const int *active, int norm_after, double alpha, void *stream);
int ce_ca_finish(ce_handle h, int B, int lp, int max_iters, const double *W, const double *UT, const double *U, const double *D,
                 const double *E, const double *b_hat, const double *c_hat, const double *sigma, const double *scale,
                 const int *active, int *status, int *iters, double *x, double *y, double *s, void *stream);

/*
 * Shared-A forward  <- _build_diffcp_matrices + diffcp.solve_and_derivative_batch (diffcp_if.py:46-70, 365-372) for templates whose A does
 * not depend on the parameters and consists of r <= 64 rows with several entries plus rows with a single entry: the whole solve of an
 * instance in one persistent kernel (iterates in LDS, reduced KKT matrix = diagonal + rank r applied by the Woodbury identity, its
 * r x r core formed on the matrix cores, termination / adaptive scale in the kernel, PSD projection with MFMA contractions).
 * The caller equilibrates the ONE shared matrix (cvxpylayers_amd/interfaces/const_a.py) and passes, all device memory:
 *   AdT (n, RP) row-major: the equilibrated dense rows transposed (solver sign A = -A_cvx), zero padded to RP in {16, 32, 64};
 *   drow (r): their row indices; srow_col / srow_val (m): column (-2 - a: the dense row in slot a, -1: empty row) and value of every single-entry row;
 *   scol_ptr (n + 1) / scol_row: the single-entry rows of every column; gs (n): sum over them of d0_i a_i^2 (d0 = 1000 on zero-cone rows);
 *   Dv (m), Ev (n): the equilibration; b_hat (B, m), c_hat (B, n), sigma / nrm_b0 / nrm_c0 (B): normalised data as in ce_ca_check;
 *   warm_x / warm_y / warm_s: (B, .) initial point used when settings->warm_start != 0, else NULL.
 * Outputs as ce_solve.  CE_E_UNSUPPORTED / CE_E_TOO_LARGE: callers use the batch-GEMM path of const_a.py instead.
 */
int ce_solve_shared_a(ce_handle h, int B, int r, int RP, const double *AdT, const int *drow, const int *srow_col, const double *srow_val,
                      const int *scol_ptr, const int *scol_row, const double *gs, const double *Dv, const double *Ev, const double *b_hat,
                      const double *c_hat, const double *sigma, const double *nrm_b0, const double *nrm_c0, const ce_settings *settings,
                      const double *warm_x, const double *warm_y, const double *warm_s,
                      double *x, double *y, double *s, int *iters, int *status, double *resid, void *stream);

/*
 * Shared-A adjoint  <- _compute_gradients -> adj_batch (diffcp_if.py:73-96, 385-403) for templates whose A does not depend on the
 * parameters: diffcp's adjoint system (r_tau = 0) solved by LSQR -- diffcp's own default mode -- entirely inside one kernel, one
 * workgroup per instance, A applied from its sparse structure, the PSD cone's derivative on the matrix cores.  A_vals0: the nnz_aug
 * boundary values of instance 0 (the A part is shared); instance i's b entries are read at A_vals0 + i * sA_b (sA_b = nnz_aug for a batch-major
 * value matrix, 0 when b is shared too); q_vals: c of instance i at [j * sq_k + i * sq_b] (as ce_solve).  With q_vals the system is diffcp's
 * FULL (n + m + 1) adjoint system, tau row and column included -- on rank-deficient systems LSQR's minimum-norm solution is then diffcp's;
 * q_vals == NULL pins r_tau = 0 (the n + m system of ABI <= 8: the same gradients wherever the system is regular).
 * x, y, s, dx, dy as ce_vjp; dA_bm (B, nnz_aug) batch-major; dq at
 * [k * sdq_k + i * sdq_b]; adj_status[i] = 1 when LSQR hit iter_lim (0: diffcp's 2 (n + m + 1)); lsqr_iters (B) or NULL; atol / btol: LSQR stopping
 * tolerances (diffcp runs 1e-8 / 1e-8: the plugin's default, solver_args lsqr_atol / lsqr_btol / lsqr_iter_lim override); conlim: LSQR stops when its estimate
 * of cond(M^T) exceeds it (diffcp / scipy default 1e8; <= 0 disables the test).  The engine refills its own scratch (split of this call's A values) on `stream`: one engine, one stream at a time.  All cone types (zero / nonnegative / second-order / PSD / exponential / power: the triples' derivative is a symmetrised
 * 3 x 3 block computed once per call); CE_E_TOO_LARGE when the LSQR vectors of one instance exceed LDS (callers fall back to the batched
 * path of const_a.py).
 */
int ce_vjp_shared_a(ce_handle h, int B, const double *A_vals0, long sA_b, const double *q_vals, long sq_k, long sq_b,
                    const double *x, const double *y, const double *s, const double *dx, const double *dy,
                    double *dA_bm, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, int *lsqr_iters, double atol, double btol, double conlim, int iter_lim, void *stream);

/*
 * diffcp's LSQR adjoint for templates whose A DOES depend on the parameters  <- adj_batch(..., mode="lsqr"), the mode diffcp_if.py:86 runs by default.
 * ce_vjp solves the adjoint system by a rank-revealing direct elimination (the same gradients wherever the system is regular: every BASELINE configuration);
 * on a rank-deficient system it returns a basic solution where diffcp's LSQR returns the minimum-norm one.  This entry runs ce_vjp_shared_a's LSQR kernel with
 * the A part read per instance: A_vals_bm (B, nnz_aug) batch-major (sA_b = nnz_aug), every other argument as ce_vjp_shared_a.  The plugin selects it with
 * solver_args mode="lsqr".  A is streamed from L2 / HBM twice per LSQR iteration: the direct elimination stays the default for speed.
 */
int ce_vjp_lsqr(ce_handle h, int B, const double *A_vals_bm, long sA_b, const double *q_vals, long sq_k, long sq_b,
                const double *x, const double *y, const double *s, const double *dx, const double *dy,
                double *dA_bm, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, int *lsqr_iters, double atol, double btol, double conlim, int iter_lim, void *stream);

/* Longest-first dispatch.  Workgroups are dispatched in index order and one workgroup owns one instance, so the tail of a forward launch is set by the
 * instances that happen to start last: when they are long ones the last slots drain slowly (13 % of the metric configuration's kernel time).  With the switch
 * on, every ce_solve also records the order "instances by iteration count, largest first" (one tiny kernel behind the solve) and the NEXT ce_solve of the
 * same batch size dispatches its workgroups in that order -- PROVIDED the history has been predictive: the order is applied only when at least 70 % of the
 * instances of the last call stopped in the same check interval as the instance at the same position of the call before (decided on the device, no host
 * round trip).  Re-solved or slowly changing batches (full-batch training loops, parameter sweeps over a fixed data set) qualify from the third call on; on
 * unrelated batches (fresh mini-batches) the workgroups keep the index order, which is what is fastest there (a permutation that predicts nothing only scatters
 * the instances' rows over HBM: 1.5 % slower on the metric configuration).  A scheduling hint only: results are bit-identical in any order.
 * Register-tiled forward kernels only (fwd_mode 4).  Off by default at the C ABI. */
int ce_set_dispatch_history(ce_handle h, int on);

/* The LSQR re-solve of ce_vjp (see there): enable (default 1) and its stopping rule -- Paige & Saunders' atol / btol / conlim and the iteration limit
 * (0: diffcp's 2 (n + m + 1)); defaults = diffcp's adj_batch(mode="lsqr"): 1e-8, 1e-8, 1e8, 0.  <- diffcp_if.py:86. */
int ce_set_adjoint_resolve(ce_handle h, int enable, double atol, double btol, double conlim, int iter_lim);
/* >= 0: ce_vjp calls with the re-solve armed (q_vals given, enabled) run the SEARCH-FREE elimination kernel k_backward_ns (csrc/ce_backward_ns.h: the equality rows
 * are eliminated by one wave with column pivoting, the reduced Hessian on the null space is formed and swept on the matrix cores without a pivot search; a
 * vanishing pivot flags the instance for the LSQR re-solve) -- plain-cone templates with a linear objective and n <= 108; the value is the tile variant.
 * -1: the pivoting elimination kernels (k_backward_rt / k_backward) serve every call.  Introspection for tests / bench. */
int ce_adjoint_ns_variant(ce_handle h);

/* Introspection used by bench.py / tests: per-kernel HIP-event timing on the launch stream.  enable: 0 off, 1 every launch, or a sum of 2 (forward launches),
 * 4 (adjoint launches), 8 (layout passes) to bracket only those kinds (two event records per bracketed launch are host work in front of the launch). */
int ce_set_profiling(ce_handle h, int enable);
/* which: 0 forward kernel, 1 backward kernel, 2 layout (transpose) kernels.  Returns the mean ms per launch
 * since the last ce_reset_profile and the launch count; synchronises the recorded events. */
int ce_get_profile(ce_handle h, int which, double *mean_ms, int *launches);
int ce_reset_profile(ce_handle h);
/* LDS bytes per workgroup and residency mode chosen for (forward, backward); for DESIGN/bench reporting.
 * fwd_mode: 0..2 size-generic kernel (everything in LDS / G in global memory / A and G in global memory), 3 k_forward_rt, 4 k_fwd2 (register tiles,
 * LDS operand streams).
 * bwd_mode: 0..2 size-generic kernel, 3 k_backward_rt. */
int ce_get_launch_info(ce_handle h, int *fwd_lds_bytes, int *bwd_lds_bytes, int *fwd_mode, int *bwd_mode);

#ifdef __cplusplus
}
#endif
#endif /* CONE_ENGINE_H */
