/*
 * semiclassical_hip.h -- C-ABI of the MI355X (gfx950) trajectory engine.
 *
 * The reference (humeniuka/semiclassical) has no FFI boundary: its hot path is a
 * sequence of eager PyTorch ops inside two Python classes.  This header is the
 * boundary the build introduces underneath those classes: every entry point
 * replaces one group of reference ops (cited per function, paths relative to the
 * reference repository).  The Python classes in semiclassical_amd/propagators.py
 * bind these symbols with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - all pointers inside the structs are DEVICE pointers (torch owns every
 *     buffer; nothing is allocated, freed or retained by the library);
 *   - fp64 everywhere; complex numbers are interleaved (re, im) doubles;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - every function returns 0 on success or a negative SC_ERR_* code and never
 *     synchronises the device; sc_last_error() gives the message of the last
 *     failure on the calling thread.
 *
 * Engine-native state layout ("trajectory-major"):
 *     qp   [n][2*D]        q(0..D-1), p(0..D-1) of one trajectory are contiguous
 *     act  [n]             classical action S
 *     mono [n][4][D][D]    monodromy blocks Mqq, Mqp, Mpq, Mpp (row-major a,b)
 *     c2   [n] complex     HK prefactor squared  (the sign tracker's "previous")
 *     sgn  [n]             accumulated branch sign (+1/-1) of sqrt(c2)
 * The reference's (rows, n) tensor `y` is produced from / converted to this
 * layout by sc_state_from_reference / sc_state_to_reference.
 */
#ifndef SEMICLASSICAL_HIP_H
#define SEMICLASSICAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on every change of a struct layout or a function signature below.  semiclassical_amd/_lib.py refuses a
 * library whose sc_abi_version() or struct sizes differ from its own declarations. */
#define SC_ABI_VERSION        17

#define SC_OK                 0
#define SC_ERR_BAD_ARGUMENT  -1
#define SC_ERR_UNSUPPORTED   -2   /* shape outside what the kernels hold on chip */
#define SC_ERR_LAUNCH        -3   /* hip launch / runtime error */

/* potential kinds: reference semiclassical/potentials.py */
#define SC_POT_MORSE          1   /* MorsePotential, chi != 0           potentials.py:274-327 */
#define SC_POT_HARMONIC_SEP   2   /* MorsePotential, all chi == 0       potentials.py:267-312 */
#define SC_POT_EPS_MORSE      3   /* NonHarmonicPotential               potentials.py:63-134  */
#define SC_POT_HARMONIC_DENSE 4   /* MolecularHarmonicPotential         potentials.py:553-593 */

typedef struct sc_potential {
    int32_t kind;           /* SC_POT_* */
    int32_t dim;            /* D */
    const double *par0;     /* MORSE: a[D]  HARMONIC_SEP: omega^2[D]  EPS_MORSE: eps[D]  DENSE: pos0[D]      */
    const double *par1;     /* MORSE: De[D]                            EPS_MORSE: b[D]    DENSE: grad0[D]     */
    const double *par2;     /*                                                            DENSE: hess0[D][D]  */
    double        scalar0;  /*                                                            DENSE: energy0 - origin */
    const double *inv_mass; /* 1/m [D]                                 potentials.py:261-263, 549 */
    const double *lin_prop; /* DENSE, optional (may be NULL): the RK4 step matrix Phi[2D][2D] = sum_{k<=4} (lin_dt G)^k / k!
                               of the monodromy equations, G = [[0, diag 1/m], [-hess0, 0]] (row-major).  One RK4
                               step of a linear constant-coefficient system is the product with Phi; sc_hk_step
                               uses it for its small-D register kernel when lin_dt equals the step it is called with. */
    double        lin_dt;
} sc_potential;

/* Storage order of the 4 D^2 doubles mono[i] of one trajectory.
 *   SC_MONO_ROWMAJOR  [4][D][D]: Mqq, Mqp, Mpq, Mpp one after the other, each row-major (a, b).  Every entry point
 *                     takes this order.
 *   SC_MONO_TILED16   16 x 16 tiles in the order (ra, rb) of tile rows / columns; tile (ra, rb) stores its nra x ncb
 *                     part (nra = min(16, D - 16 ra), ncb likewise) of the plane PAIR (Mqq, Mqp) element by element
 *                     (row-major inside the tile, the two planes of an element side by side), then the pair (Mpq, Mpp)
 *                     the same way (ABI 16; up to ABI 15 the four planes followed one another):
 *                         offset(p, a, b) = 4 (16 ra D + 16 nra rb) + (p / 2) 2 nra ncb + 2 ((a - 16 ra) ncb + (b - 16 rb)) + p % 2
 *                     A thread of the fast kernel then moves both planes of a pair with one 16-byte access.
 *                     Identical to SC_MONO_ROWMAJOR for D <= 16.  Only sc_hk_step on its separable / diagonal-width
 *                     fast path takes it (the 16 x 16 thread grid of that kernel then walks a trajectory linearly
 *                     through HBM); sc_mono_convert switches a state between the two in place. */
#define SC_MONO_ROWMAJOR 0
#define SC_MONO_TILED16  1

typedef struct sc_state {
    int64_t n;              /* trajectories in this batch (this rank's shard) */
    int32_t dim;            /* D */
    int32_t mono_layout;    /* SC_MONO_ROWMAJOR or SC_MONO_TILED16: storage order inside mono[i], see below */
    double *qp;
    double *act;
    double *mono;
    double *c2;
    double *sgn;
    double *work;           /* [n][4][D] scratch of the separable fast path: RK4 propagators of the monodromy rows */
    int32_t *flags;         /* [n + 2], zero-initialised scratch: flags[i] != 0 marks a trajectory whose determinant
                               the fast path hands to the fully pivoted elimination, flags[n] counts them for the
                               current step, flags[n + 1] is the cursor through which the fast kernel hands out
                               trajectories to its workgroups.  May be NULL: the fixed-pivot-order register kernels
                               (which rely on the fix-up) are then not used at all -- every trajectory takes the fully
                               pivoted LDS kernel, correct but several times slower */
} sc_state;

/* constants of the HK prefactor, reference propagators.py:951-1004.
 * diag != 0: Gamma_i, Gamma_t diagonal and of full rank; st = sqrt(diag Gamma_t),
 *            si = sqrt(diag Gamma_i).
 * diag == 0: L1 = U^T Gt^{1/2}, L2 = U^T Gt^{-1/2} (d' x D complex),
 *            R1 = Gi^{-1/2} U,  R2 = Gi^{1/2} U   (D x d' complex).
 *            real_lr != 0: the caller asserts that the imaginary parts of all four are zero (they are products of real
 *            matrices for positive semi-definite widths; the reference's complex128 copies carry rounding dust at most):
 *            the register kernels then run the sandwiches in real arithmetic.  0 is always correct. */
typedef struct sc_hk_consts {
    int32_t dim, dprime, diag, real_lr;
    const double *st, *si;
    const double *L1, *L2, *R1, *R2;
} sc_hk_consts;

/* constants of <q,p,G_bra | qk,pk,G_ket>, reference propagators.py:124-240.
 * diag != 0: A, B, C hold D diagonal entries, else D x D row-major.
 *   A = Gbra.iG.Gket   B = iG = (Gbra+Gket)^+   C = Gket.iG   fac = sqrt(2^rank sqrt(detGbra detGket)/detG) */
typedef struct sc_overlap_consts {
    int32_t dim, diag;
    const double *A, *B, *C;
    const double *qk, *pk;  /* centre of the ket, [D] each */
    double fac;
} sc_overlap_consts;

/* constants of the non-adiabatic coupling factors, reference propagators.py:886-903
 *   n1 = -hbar^2 nac/m ;  rn = R.n1 with R = G0.iGi0.Gi ;  gn = (G0.iGi0)^T.n1 ;
 *   p0n1 = p0.n1 ;  n2 = -hbar^2/2 sum_k tau2_k/m_k (0 for every reference potential) */
typedef struct sc_nac_consts {
    int32_t dim, _pad;
    const double *rn, *gn;
    const double *q0, *p0;
    double p0n1, n2;
} sc_nac_consts;

/* constants and per-trajectory tracker state of the Walton-Manolopoulos prefactor, reference
 * propagators.py:1102-1130 (_prepare), :1195-1389 (_prefactor).  e = 2 d'.
 *   U     D x d' (real)      eigenvectors of Gamma_0 + Gamma_i with non-zero eigenvalue      :495-498
 *   Gt, G0, iGi0, S = iGi0.G0, Cqq = G0 - G0.iGi0.G0                     D x D real          :1280-1296
 *   Cst   e x e complex = U2^T (2 filinov + Eqz^T Gi Eqz - 2i/hbar Epz^T Eqz) U2            :1227-1238
 *   Bq    D x e complex = [Gi U, -i/hbar U]                                                  :1264
 *   n1 = -hbar^2 tau1/m, s_n1 = S n1, w_n1 = G0 n1 (NULL until the potential is known)       :1693
 *   inv_scale_a = 1/(2 sqrt(alpha beta))   :1328       inv_two_pi = 1/(2 pi)   :1358
 *   pre = detG0^1/2 detGt^1/4 detGi^1/4 / sqrt(detGi0) with the pi-absorbing determinants    :1116-1125, 1598-1599
 *   detA, detM [n] complex + sgnA, sgnM [n]: "previous" values and branch signs of the two trackers :1336, 1389 */
typedef struct sc_wm_consts {
    int32_t dim, dprime;
    const double *U, *Gt, *G0, *iGi0, *S, *Cqq;
    const double *Cst, *Bq;
    const double *q0, *p0;
    const double *n1, *s_n1, *w_n1;
    double inv_scale_a, inv_two_pi, pre, p0n1, n2;
    double *detA, *detM, *sgnA, *sgnM;
    /* optional per-trajectory export behind WaltonManolopoulosPropagator.coefficients() / wavefunction() / norm()
     * (propagators.py:1391-1575); each may be NULL:
     *   coef_out [n] complex   v_n of eqn (75) without the x-dependent part, :1408-1432, with
     *                          pre_coef = detG0^1/4 detGt^1/4 detGi^1/4 / sqrt(detGi0)
     *   cqq_out  [n][D][D] complex   C_QQ of eqn (70)
     *   dvec_out [n][D] complex      C_qQ^T (q0 - q) + i/hbar PI_Q, :1513 */
    double pre_coef;
    double *coef_out, *cqq_out, *dvec_out;
    /* matrix storage for shapes whose per-trajectory matrices exceed the LDS of a compute unit (D >~ 24 at full
     * rank): sc_wm_scratch_bytes(n, D, d') bytes of device memory, 0 / NULL when not needed.  Contents are scratch. */
    double *scratch;
    int64_t scratch_bytes;
    /* [n + 1] int32 scratch, or NULL.  The register-resident kernel eliminates in a fixed pivot order; a trajectory
     * whose pivot falls more than a factor 16 below what partial pivoting would have picked is marked flags[i] = 1
     * (flags[n] counts them) and recomputed with full partial pivoting by the LDS kernel in the same call.  NULL: the
     * register kernel is not used, every trajectory takes the fully pivoted LDS kernel. */
    int32_t *flags;
    /* [n][3 D + 3] or NULL.  Position-dependent derivative couplings (potentials whose derivative_coupling_1st / _2nd
     * depend on r; the reference evaluates them at the initial and the current points of every trajectory,
     * propagators.py:1685-1693): per trajectory n1(q_i)[D], S n1(q_i)[D], G0 n1(Q)[D], p0.n1(Q), n2(q_i), n2(Q), filled by
     * the caller before every call; they replace n1, s_n1, w_n1, p0n1, n2 above and route the call to the LDS / scratch
     * kernels. */
    const double *nac_traj;
} sc_wm_consts;

/* sGDML force field, reference semiclassical/gdml_predictor.py:57-85 (constructor) and :96-250 (forward).
 *   xs_train, jx_alphas [n_train][n_desc]: permutation-expanded training descriptors and Jacobian-contracted
 *   coefficients (16-byte aligned: the kernels copy blocks of rows with 16-byte loads); pair_k/pair_l [n_desc]: atoms (k > l) of descriptor d in torch.tril_indices order;
 *   q = sqrt(5)/sigma; energies are returned relative to `origin` (potentials.py:699). */
typedef struct sc_gdml_model {
    int32_t n_atoms, n_desc, n_train, _pad;
    const double *xs_train, *jx_alphas;
    const int32_t *pair_k, *pair_l;
    double q, c, std, origin;
    const double *inv_mass;     /* [3 n_atoms] */
} sc_gdml_model;

/* per-trajectory scratch of the unfused RK4 step for dense, position-dependent Hessians */
typedef struct sc_dense_scratch {
    double *hess;   /* [n][4][D][D]  stage Hessians */
    double *kprev;  /* [n][2D]       slopes of (q, p) of the previous stage */
    double *ksum;   /* [n][2D]       weighted slope sums k1 + 2 k2 + 2 k3 (+ k4) */
    double *ssum;   /* [n]           weighted sums of T - V */
} sc_dense_scratch;

int         sc_version(void);
const char *sc_last_error(void);
/* ABI guard: SC_ABI_VERSION the library was compiled with; sizeof of the struct called `name` ("sc_state", ...), or
 * -1 for an unknown name; 1 if the library is a tuning build (-DSC_TUNING: reads experiment knobs from the
 * environment), 0 for the product build. */
int sc_abi_version(void);
int sc_struct_size(const char *name);
int sc_tuning_build(void);

/* number of workgroups sc_hk_step / sc_hk_correlate use for n trajectories of dimension D;
 * the caller sizes the `partials` buffers with it. */
int sc_step_grid(int64_t n, int32_t dim);
int sc_correlate_grid(int64_t n, int32_t dim);

/* In-place change of st->mono between the storage orders (the caller updates st->mono_layout afterwards);
 * `to_layout` is the order wanted, st->mono_layout the order the data is in.  D <= 64. */
int sc_mono_convert(const sc_state *st, int32_t to_layout, void *stream);

/* y (rows = 2D+4D^2+1, n) with n fastest  <->  engine layout.   reference propagators.py:329-334, 581 */
int sc_state_from_reference(const double *y, const sc_state *st, void *stream);
int sc_state_to_reference(const sc_state *st, double *y, void *stream);

/* Initial conditions sampled ON THE DEVICE, replaces the sampling half of HermanKlukPropagator.initial_conditions
 * (propagators.py:537-566: xi ~ N(0,1) of shape (2d', n), zi = z0 + iLz^T xi, probi = detLz/(2 pi)^D exp(-|xi|^2/2))
 * and, with init_state != 0, the construction of y(0) (:581-603: q, p = zi, S = 0, Mqq = Mpp = 1; also c2 = 1,
 * sgn = 1; st->mono is written ROW-MAJOR).
 *   ilz   [2d'][2D] row-major = block_diag(iLq, iLp) (:506-528), z0 [2D] = (q0, p0), prob0 = detLz/(2 pi)^D
 *   zi_t  [n][2D] out, probi [n] out, xi_out [n][2d'] out or NULL (the deviates themselves)
 * Deviates: Philox4x32-10 + Box-Muller, key = seed, counter = (trajectory index, pair index, subsequence): deviate j of the
 * trajectory with GLOBAL index first + i depends only on (seed, subsequence, first + i, j) -- not on n, the launch shape or
 * the rank that draws it -- and distinct (seed, subsequence) never share a stream.  d' <= 256, subsequence < 2^56. */
int sc_sample_initial(const sc_state *st, const double *ilz, const double *z0, int32_t dprime, double prob0,
                      uint64_t seed, uint64_t subsequence, int64_t first, int32_t init_state, double *zi_t,
                      double *probi, double *xi_out, void *stream);

/* One RK4 step of (q,p,Mqq,Mqp,Mpq,Mpp,S) followed by the HK prefactor and its sqrt-branch tracking,
 * fused per trajectory.  Replaces _rk4_step + EquationsOfMotion.f + potential.harmonic_approximation
 * (propagators.py:86-119, 313-383) and _prefactor + _track_signs_of_sqrt (propagators.py:951-1052).
 *   energy_partials[sc_step_grid()]: per-workgroup sums of T+V at the k4 stage (propagators.py:380).
 *   mode 0: step + prefactor;  mode 1: prefactor only with tracker initialisation (t = 0, propagators.py:631). */
int sc_hk_step(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk,
               double dt, int32_t mode, double *energy_partials, void *stream);

/* TWO consecutive time steps per visit of a trajectory (round 4): the same arithmetic as two calls of sc_hk_step, for the
 * separable / diagonal-width fast path with 16 < D <= 64 on the tiled storage order (sc_hk_step_multi_supported).  A workgroup
 * streams a trajectory's monodromy blocks, applies step k, stores them, eliminates -- and streams the SAME trajectory again for
 * step k + 1: these loads hit the L2 / memory-side cache instead of HBM (profiles/r4_revisit.txt), one of the four HBM transfers
 * of two time steps is gone.  Results are bit-identical to two sc_hk_step calls (tests/test_hk_multi_gpu.py).  Measured at
 * D = 60, n = 1e5: 4.38 instead of 4.55 ms per step -- the streaming phase alone drops from 4.39 to 2.94 ms per step, but the
 * elimination (FP64 VALU issue) then bounds the kernel at 3.8-4.4 ms (profiles/r4_sd_phases.txt).
 *   ms->work    [2][n][4][D]   row propagators of both sub-steps (scratch)
 *   ms->qp_mid  [n][2D], act_mid [n], c2_mid [n] complex, sgn_mid [n]: the state BETWEEN the two steps -- what sc_hk_correlate of
 *               the second time step reads (pass an sc_state whose qp / act / c2 / sgn point to them)
 *   ms->unrepaired   one int32 on the device, zeroed by the caller.  The fix-up of a weak in-block pivot (sc_state.flags) needs
 *               the blocks the determinant belongs to; for the FIRST sub-step they have moved on.  The kernel counts such
 *               determinants here: non-zero after the call = the intermediate determinants (and everything derived from them)
 *               are not reliable, redo the two steps from a saved state with sc_hk_step.  Monodromy blocks that are
 *               diagonal (separable potential from M(0) = 1) cannot have weak pivots; HermanKlukPropagator.run() takes this
 *               entry point only then.
 *   energy_partials [2][sc_step_grid()]: sums of T+V at the k4 stage of either step (two sc_energy_guard calls). */
typedef struct sc_multi_scratch {
    double *work, *qp_mid, *act_mid, *c2_mid, *sgn_mid;
    int32_t *unrepaired;
} sc_multi_scratch;
int sc_hk_step_multi_supported(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk);
int sc_hk_step_multi(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_multi_scratch *ms,
                     double dt, double *energy_partials, void *stream);

/* out[i] = <qp[i] , G_bra | qk,pk,G_ket> for i < n  (complex).  propagators.py:181-240 with a single ket. */
int sc_overlap(const sc_overlap_consts *oc, const double *qp, int64_t n, double *out, void *stream);

/* nac factor of the INITIAL points: nacq = n2 + (q0-q).rn + i/hbar (p0n1 + (p-p0).gn).  propagators.py:902-903 */
int sc_nac_initial(const sc_nac_consts *nc, const double *zi, int64_t n, double *nacq, void *stream);

/* Per-trajectory terms of C_auto and k_ic for the current state and their per-workgroup partial sums.
 * Replaces autocorrelation_qp / autocorrelation / ic_correlation (propagators.py:784-911) up to the
 * phase exp(i t E0/hbar), which the host applies.
 *   w_i      = 1 / (mc_norm * probi_i),  mc_norm = N_total (2 pi hbar)^D      propagators.py:837
 *   cq_i     = conj(vt_i) vi_i sgn_i sqrt(c2_i) exp(i S_i/hbar) w_i           propagators.py:806
 *   kq_i     = nacQ_i nacq_i cq_i / hbar^2  (skipped when nc == NULL)         propagators.py:900-909
 *   partials[sc_correlate_grid()][4] = sums of (cq.re, cq.im, kq.re, kq.im) per workgroup
 *   cq_out / kq_out (complex [n]) may be NULL. */
int sc_hk_correlate(const sc_state *st, const sc_overlap_consts *ovl_t0, const sc_nac_consts *nc,
                    const double *vi, const double *probi, const double *nacq, double mc_norm,
                    double *cq_out, double *kq_out, double *partials, void *stream);

/* slot[0..3] = sum of correlate partials, slot[4] = (sum of energy partials)/n_energy.
 * Deterministic (fixed summation order).  Either partial buffer may be NULL (its slots are left untouched). */
int sc_reduce_slot(const double *corr_partials, int32_t n_corr, const double *energy_partials, int32_t n_energy_blocks,
                   double n_energy, double *slot, void *stream);

/* The same sums into ROW *cursor of slots[.][5], then *cursor += 1 (cursor: one int64 in device memory).  With the
 * destination held on the device a launch sequence "correlate, reduce, step" has no argument that changes from step to
 * step: it is captured once in a HIP graph and replayed (HermanKlukPropagator.run(use_graph=True)). */
int sc_reduce_slot_at(const double *corr_partials, int32_t n_corr, double *slots, int64_t *cursor, void *stream);

/* The WHOLE caller loop (cli.py:401-436: nsteps times "autocorrelation, ic_correlation, step") in one launch, for
 *   - separable potentials (SC_POT_MORSE / HARMONIC_SEP / EPS_MORSE) with diagonal width matrices and D <= 12,
 *   - a constant dense Hessian (SC_POT_HARMONIC_DENSE with its step matrix lin_prop built for this dt), any width matrices
 *     (hk->real_lr for dense ones), at the small molecular shapes D <= 12 the library instantiates (methylium: D = 12, d' = 6):
 * a trajectory is loaded once into registers, runs all steps there and is written back once (sc_hk_run_supported says
 * whether the combination qualifies; everything else takes sc_hk_correlate / sc_hk_step step by step).  Arguments as for
 * sc_hk_correlate and sc_hk_step.  partials: scratch of 5 * sc_hk_run_slots(n, D) * nsteps doubles;
 * slots_out [nsteps][5]: row k = Re C, Im C, Re k, Im k summed over the trajectories for the state BEFORE step k (what
 * sc_hk_correlate + sc_reduce_slot give) and the mean <T+V> at the k4 stage of step k; elog: the energy-guard log of
 * sc_energy_guard, advanced by nsteps steps. */
int sc_hk_run_slots(int64_t n, int32_t dim);
int sc_hk_run_supported(const sc_potential *pot, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0);
int sc_hk_run(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0,
              const sc_nac_consts *nc, const double *vi, const double *probi, const double *nacq, double mc_norm,
              double dt, int32_t nsteps, double *partials, double *slots_out, double *elog, void *stream);

/* sc_hk_run for a CONSTANT dense Hessian with the monodromy blocks of the state in NORMAL-MODE coordinates (round 4).  RK4 of a
 * linear system commutes with a change of basis: with W = m^-1/2 H m^-1/2 = U diag(lambda) U^T, A = m^-1/2 U, B = m^1/2 U and
 *     Mqq~ = A^-1 Mqq A,  Mqp~ = A^-1 Mqp B,  Mpq~ = B^-1 Mpq A,  Mpp~ = B^-1 Mpp B
 * the step matrix Phi(dt) of sc_potential.lin_prop becomes 2 x 2 per mode, mode_prop[a] = (phi_qq, phi_qp, phi_pq, phi_pp)_a =
 * the RK4 polynomial of dt [[0, 1], [-lambda_a, 0]], and the prefactor is the same expression of the transformed blocks with
 * L1 A, L2 B, A^-1 R1, B^-1 R2 as the prefactor constants -- real, dense: diag = 0.  The CALLER transforms st->mono before the call and back
 * after it (HermanKlukPropagator.run does, with two batched products); (q, p, S), the correlation terms and the energy guard are
 * in the original coordinates as in sc_hk_run.  Per lane and step 8 D multiply-adds replace the 8 D^2 of the product with Phi. */
/* mono[i][p] <- left[p] . mono[i][p] . right[p] for the four blocks p = qq, qp, pq, pp of every trajectory (left, right: [4][D][D],
 * row-major state, D <= 16): the change of basis in front of and behind sc_hk_run_modal. */
int sc_mono_similarity(const sc_state *st, const double *left, const double *right, void *stream);
int sc_hk_run_modal_supported(const sc_potential *pot, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0);
int sc_hk_run_modal(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0,
                    const sc_nac_consts *nc, const double *vi, const double *probi, const double *nacq, double mc_norm,
                    double dt, int32_t nsteps, const double *mode_prop, double *partials, double *slots_out, double *elog, void *stream);

/* Walton-Manolopoulos: Filinov matrix A (eqn 50), its inverse and determinant, Gt/Gti/CQQ/M (57-78), the second
 * inverse and determinant, the trackers of sqrt(detA), sqrt(detM) and the per-trajectory terms of eqns (85), (100),
 * fused per trajectory.  Replaces WaltonManolopoulosPropagator._expand_L/_prefactor/autocorrelation_qp/
 * autocorrelation/ic_correlation (propagators.py:1132-1389, 1577-1719) after sc_hk_step has advanced the state.
 *   track: 2 initialise the trackers (t = 0), 1 track against the previous step, 0 recompute with the stored signs
 *   has_nac == 0: k_ic terms are skipped (wc->n1 etc. may be NULL)
 *   cq_out/kq_out complex [n] may be NULL; partials[sc_wm_grid()][4] as in sc_hk_correlate.
 * Three kernels behind it: a register-resident one (D <= 16, e <= 16 at the instantiated shapes), one with every matrix of
 * the trajectory in LDS, and the same with the matrices in wc->scratch for shapes beyond the LDS (no size limit other
 * than memory: sc_wm_scratch_bytes says how much scratch the call needs, 0 if none, -1 for invalid shapes).
 * sc_wm_grid: rows of `partials` the caller provides and sums (the kernels' own slots + those of the pivoted re-run). */
int sc_wm_grid(int64_t n, int32_t dim);
int64_t sc_wm_scratch_bytes(int64_t n, int32_t dim, int32_t dprime);
int sc_wm_correlate(const sc_state *st, const sc_wm_consts *wc, const double *zi, const double *probi,
                    double mc_norm, int32_t track, int32_t has_nac, double *cq_out, double *kq_out,
                    double *partials, void *stream);

/* partner atoms per atom the sGDML kernel instantiated for a molecule of n_atoms atoms holds (8, 16, 20, 24 or 32);
 * -1 beyond 48 atoms, which sc_gdml_eval / sc_gdml_stage refuse */
int sc_gdml_row_len(int32_t n_atoms);

/* E - origin [n], dE/dr [n][3N], d2E/drdr [n][3N][3N] of the sGDML model at the geometries r [n][3N].
 * Replaces GDMLPredict.forward / MolecularGDMLPotential.harmonic_approximation (gdml_predictor.py:96-250). */
int sc_gdml_eval(const sc_gdml_model *g, const double *r, int64_t n, double *energy, double *grad, double *hess,
                 void *stream);

/* One RK4 stage (stage = 0..3) of (q, p, S) on the sGDML surface: stage point from the previous slope, potential
 * evaluation, slopes, stage Hessian -> sc->hess; stage 3 also writes the new (q, p, S) and the per-workgroup sums of
 * T+V (energy_partials[sc_dense_grid()]).  With sc_dense_mono_step it replaces _rk4_step / EquationsOfMotion.f
 * (propagators.py:86-119, 313-383) for potentials whose Hessian is dense and position dependent. */
int sc_dense_grid(int64_t n);
/* The same RK4 bookkeeping of (q, p, S) for potentials the CALLER evaluates (any object with the reference's potential
 * protocol, potentials.py:41-204): per stage, sc_stage_point writes the stage positions r_out [n][D], the caller
 * evaluates V [n], grad [n][D] (and copies its Hessians into sc->hess[n][stage][D][D]), sc_stage_consume advances the
 * slopes / sums exactly as sc_gdml_stage does (energy_partials[sc_dense_grid()] at stage 3). */
int sc_stage_point(const sc_state *st, const sc_dense_scratch *sc, double dt, int32_t stage, double *r_out, void *stream);
int sc_stage_consume(const sc_state *st, const sc_dense_scratch *sc, const double *inv_mass, const double *V,
                     const double *grad, double dt, int32_t stage, double *energy_partials, void *stream);
int sc_gdml_stage(const sc_gdml_model *g, const sc_state *st, const sc_dense_scratch *sc, double dt, int32_t stage,
                  double *energy_partials, void *stream);

/* RK4 of the four monodromy blocks with the four stage Hessians hess[n][4][D][D] (each used as A[i][k] = hess[k][i];
 * the built-in potentials produce symmetric images), then the HK prefactor (diagonal or dense/rank-deficient width
 * matrices) and its branch tracking.  D <= 96: on the FP64 matrix cores; for 64 < D the RK4 sums do not fit the
 * register file and live in `mono_sums`.  D > 96: no size limit -- every matrix of a trajectory in a per-workgroup block
 * of `mono_sums`, products on the vector ALUs, pivoted LU on global memory (slow, kept for parity with the unbounded
 * reference).  `mono_sums`: sc_dense_mono_scratch_bytes(n, D, d') bytes (0: may be NULL).
 * mode: 0 = step + prefactor, 1 = prefactor and tracker initialisation only. */
int64_t sc_dense_mono_scratch_bytes(int64_t n, int32_t D, int32_t dprime);
int sc_dense_mono_step(const sc_state *st, const sc_hk_consts *hk, const double *inv_mass, const double *hess,
                       double *mono_sums, double dt, int32_t mode, void *stream);

/* Structure-exploiting HK step ("separable shortcut"): same contract as sc_hk_step for a
 * separable potential (SC_POT_MORSE / _HARMONIC_SEP / _EPS_MORSE), diagonal width matrices (hk->diag) and monodromy
 * blocks that are diagonal, held as mono_diag[n][4][D] (diagonals of Mqq, Mqp, Mpq, Mpp) instead of st->mono, which
 * is NOT touched.  Replaces the same reference code as sc_hk_step (propagators.py:86-119, 313-383, 951-1052) for this
 * special case; opt-in on the host, reported separately from the dense-state kernel. */
int sc_hk_step_diag(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, double *mono_diag,
                    double dt, int32_t mode, double *energy_partials, void *stream);

/* O(n^2) pairwise sum  sum_ij wb_i wk_j exp(rs_i + rs_j + X1_i.Y1_j + i (ib_i + ik_j + X2_i.Y2_j))  behind
 * HermanKlukPropagator.norm() (propagators.py:734-782): X1/Y1 [n][K1], X2/Y2 [n][K2] real, rs/ib/ik [n] real,
 * wb/wk [n] complex; partials[sc_pair_sum_tiles(n)][4] (re, im, 0, 0) for sc_reduce_slot. */
int64_t sc_pair_sum_tiles(int64_t n);
/* ... and over bras i of one set of trajectories (X operands, rs_i, ib, wb: ni rows) and kets j of another (Y operands,
 * rs_j, ik, wk: nj rows): a rank's shard against the gathered ensemble, for norm() across ranks. */
int64_t sc_pair_sum_rect_tiles(int64_t ni, int64_t nj);
int sc_pair_sum_rect(const double *X1, const double *Y1, int32_t K1, const double *X2, const double *Y2, int32_t K2,
                     const double *rs_i, const double *rs_j, const double *ib, const double *ik, const double *wb,
                     const double *wk, int64_t ni, int64_t nj, double *partials, void *stream);
int sc_pair_sum(const double *X1, const double *Y1, int32_t K1, const double *X2, const double *Y2, int32_t K2,
                const double *rs, const double *ib, const double *ik, const double *wb, const double *wk,
                int64_t n, double *partials, void *stream);

/* Frozen-Gaussian wavefunction on a spatial grid behind HermanKlukPropagator.wavefunction() (propagators.py:252-292,
 * 688-732):  phi[k] = fac sum_n v_n exp(-1/2 |Lx_k - Lq_n|^2 + i (p_n.x_k - pq_n)),  L = Gamma_t^(1/2).
 * LqT, PT [D][n] (trajectory index fastest), pq [n], v [n] complex, Lx, X [nx][D], phi [nx] complex. */
int sc_grid_sum(const double *LqT, const double *PT, const double *pq, const double *v, int64_t n, int32_t D,
                const double *Lx, const double *X, int32_t nx, double fac, double *phi, void *stream);

/* Walton-Manolopoulos wavefunction on a spatial grid (propagators.py:1434-1482):
 *   phi[k] = sum_n v_n exp(-1/2 dx^T CQQ_n dx + dvec_n . dx),  dx = x_k - Q_n
 * with the per-trajectory export of sc_wm_correlate; qp [n][2D] (engine state), X [nx][D], phi [nx] complex. */
int sc_wm_grid_sum(const double *qp, const double *coef, const double *cqq, const double *dvec, int64_t n, int32_t D,
                   const double *X, int32_t nx, double *phi, void *stream);

/* O(n^2) pair sum behind WaltonManolopoulosPropagator.norm() (propagators.py:1484-1575), from the per-trajectory
 * export of sc_wm_correlate and its projections cqqp [n][d'][d'] = U^T CQQ U, dvecp [n][d'] = U^T dvec (complex),
 * U [D][d'] real.  partials [sc_wm_pair_sum_tiles(n)][4] for sc_reduce_slot.  D <= 64, d' <= 16. */
int64_t sc_wm_pair_sum_tiles(int64_t n);
/* The same pair sum over bras i of one set of trajectories and kets j of another (a rank's shard against the gathered
 * ensemble: norm() across ranks); partials [sc_wm_pair_sum_rect_tiles(ni, nj)][4]. */
int64_t sc_wm_pair_sum_rect_tiles(int64_t ni, int64_t nj);
int sc_wm_pair_sum_rect(const double *qp_i, const double *coef_i, const double *cqqp_i, const double *dvecp_i, int64_t ni,
                        const double *qp_j, const double *coef_j, const double *cqq_j, const double *dvec_j,
                        const double *cqqp_j, const double *dvecp_j, int64_t nj, const double *U, int32_t D,
                        int32_t dprime, double *partials, void *stream);
int sc_wm_pair_sum(const double *qp, const double *coef, const double *cqq, const double *dvec, const double *cqqp,
                   const double *dvecp, const double *U, int64_t n, int32_t D, int32_t dprime, double *partials,
                   void *stream);

/* Energy-conservation guard on the device, reference propagators.py:385-398 (check_energy_conservation).
 * elog[4] = { <T+V>(t-dt), <T+V>(t), largest |change| seen so far, number of steps logged }.
 * The mean of this step is formed from energy_partials; the host raises the reference's RuntimeError when
 * elog[2] > 1e-2 Hartree the next time it synchronises. */
int sc_energy_guard(const double *energy_partials, int32_t n_blocks, double n_traj, double *elog, void *stream);

/* ---- multi-GPU flush -------------------------------------------------------------------------
 * Trajectories shard over the GPUs of a node, one process per GPU; every term of C_auto / k_ic already carries the weight
 * 1/(N_total P(q_i, p_i)) (propagators.py:837, 909), so the functions of the whole ensemble are the plain SUM of the
 * per-rank slot buffers -- the running mean of cli.py:453-458 over one repetition, taken across ranks.
 * sc_flush_allreduce is that sum: ONE ncclAllReduce(ncclDouble, ncclSum), in place, on `stream` (RCCL over xGMI), of
 * `count` doubles -- the caller passes the packed 4 nt sums, or the whole [nt][5] slot buffer of sc_reduce_slot_at /
 * sc_hk_run (column 4, the per-rank mean <T+V>, then holds the sum of the ranks' means).  Nothing else on the hot path
 * is a collective.
 *
 * librccl is loaded at the first sc_comm_* call (dlopen of librccl.so.1 from the loader's search path): no link-time
 * dependency, single-GPU users never load it.
 *   sc_comm_available   RCCL's version code (ncclGetVersion) if the library can be loaded, 0 otherwise
 *   sc_comm_unique_id   ncclGetUniqueId -> id_out[SC_COMM_ID_BYTES]; ONE rank calls it and hands the bytes to the others by
 *                       whatever host channel the application has (file, socket, MPI, torch's TCPStore)
 *   sc_comm_init        ncclCommInitRank on the CURRENT HIP device; collective over the nranks processes
 *   sc_comm_rank_count  rank and size of a communicator (either output may be NULL)
 *   sc_comm_destroy     ncclCommDestroy (NULL is accepted)
 * Returns SC_ERR_UNSUPPORTED when librccl cannot be loaded, SC_ERR_LAUNCH with RCCL's message on an RCCL error. */
#define SC_COMM_ID_BYTES 128
int sc_comm_available(void);
int sc_comm_unique_id(void *id_out);
int sc_comm_init(const void *id, int32_t nranks, int32_t rank, void **comm_out);
int sc_comm_rank_count(void *comm, int32_t *rank_out, int32_t *nranks_out);
int sc_comm_destroy(void *comm);
int sc_flush_allreduce(double *sums, int64_t count, void *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMICLASSICAL_HIP_H */
