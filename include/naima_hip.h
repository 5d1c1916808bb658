/* naima_hip.h -- C ABI of libnaima_hip.so: the MI355X (gfx950) implementation of
 * naima's radiative-likelihood hot path.
 *
 * The reference (zblz/naima) has NO FFI: the path sits behind Python callables
 * (SURVEY.md 8b).  This header is the boundary this build introduces; every
 * entry point names the reference function(s) it replaces (paths relative to
 * /root/reference/src/naima/).  INTEGRATION.md shows the ctypes stub a naima
 * maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a
 *     negative NH_E* code, and nh_last_error() returns a thread-local message;
 *   - all numeric data are float64, C-contiguous;
 *   - compute entry points take DEVICE pointers obtained from nh_alloc() and are
 *     asynchronous on the context's HIP stream (stream-ordered, single host
 *     thread per context, no callbacks).  nh_upload/nh_download/nh_sync move
 *     data and synchronise;
 *   - "N" is the number of walkers in the batch (emcee evaluates lnprob once
 *     per walker, core.py:450-457; here a half-ensemble is ONE call);
 *   - particle spectra travel between kernels as the per-walker weight arrays
 *       w[N][nG]  = x_i * n_w(E_i)      dlw[N][nG]: dlw[i] = ln|w[i+1]/w[i]|
 *     with x the integration variable (Lorentz factor for electrons, total
 *     energy in GeV for protons) and n the particles per unit x; the log-ratios
 *     of ADJACENT nodes (last entry unused) are what utils.py:336 needs and are
 *     assembled from small, separately accurate pieces;
 *   - emission kernels travel as TRANSPOSED tables Kt[nG][nK] with
 *     dlnKt[i][k] = ln|Kt[i+1][k]/Kt[i][k]| (last row unused); the
 *     table builders write an [nG][nE] block with row stride ld >= nE, so several
 *     seeds can sit side by side in one [nG][S*nE] table (Kt + j*nE, ld = S*nE);
 *   - spectra are returned in 1/(s eV) (intrinsic luminosity per unit energy),
 *     exactly what <Class>._spectrum() returns in the reference.
 */
#ifndef NAIMA_HIP_H
#define NAIMA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nh_ctx nh_ctx;

enum {
  NH_OK = 0,
  NH_EINVAL = -1,  /* bad argument */
  NH_EHIP = -2,    /* HIP runtime error (message has the hipError string) */
  NH_ENOMEM = -3,
  NH_ECOMM = -4    /* RCCL error */
};

/* particle distribution kinds, models.py:49-422 */
enum {
  NH_PD_POWERLAW = 0,            /* models.py:88-92   */
  NH_PD_ECPL = 1,                /* models.py:157-161 */
  NH_PD_BROKENPL = 2,            /* models.py:234-238 */
  NH_PD_ECBPL = 3,               /* models.py:330-335 */
  NH_PD_LOGPARABOLA = 4          /* models.py:402-407 */
};
/* parameter row of a particle distribution (energies in eV):
 *   [0] amplitude (1/eV)   [1] e_0      [2] alpha | alpha_1
 *   [3] e_cutoff           [4] beta     [5] e_break   [6] alpha_2   [7] unused
 * LogParabola uses [2]=alpha, [4]=beta. */
#define NH_PD_NPAR 8

/* hadronic high-energy models, radiative.py:1179-1209 */
enum { NH_PP_GEANT4 = 0, NH_PP_PYTHIA8 = 1, NH_PP_SIBYLL = 2, NH_PP_QGSJET = 3 };

/* ---- context, memory, stream ------------------------------------------- */
const char* nh_last_error(void);
int nh_version(void);
int nh_create(int device, nh_ctx** out);
int nh_destroy(nh_ctx* ctx);
/* HIP devices visible to this process (no context needed).  The reference's parallel mode asks
 * for `threads` worker processes and takes what the host gives (core.py:446-448: Pool(threads));
 * a launcher of one rank per GPU checks this first and refuses a run the node cannot hold. */
int nh_device_count(int* count);
int nh_device_info(nh_ctx* ctx, char* name, int name_len, int* compute_units,
                   double* hbm_bytes, int* clock_khz);
/* PCI bus id of the context's GPU, e.g. "0000:c1:00.0" (len >= 16): one worker per GPU is the
 * reference's Pool(threads) with GPUs for workers (core.py:446-457) -- a launcher checks that the
 * ranks' ids are distinct */
int nh_device_pci_bus_id(nh_ctx* ctx, char* out, int len);
int nh_alloc(nh_ctx* ctx, long long bytes, void** dev_out);
int nh_free(nh_ctx* ctx, void* dev);
int nh_upload(nh_ctx* ctx, void* dev_dst, const void* host_src, long long bytes);
/* host -> device on a copy stream of its own: not ordered behind the main stream's work and not
 * waited for by nh_sync (the NEXT block of stretch-move random numbers goes up while the current
 * block's launch runs; emcee draws them inside the step, StretchMove.get_proposal).  `marker` is
 * recorded behind the copy; nh_stream_wait_marker orders the main stream behind it.  `after`
 * orders the copy itself behind a point of the main stream: the destination may still be read
 * by launches queued before that point (a ring of blocks that wraps). */
int nh_upload_ahead(nh_ctx* ctx, void* dev_dst, const void* host_src, long long bytes,
                    void* marker, void* after /* marker on the main stream the copy waits for, or NULL */);
int nh_stream_wait_marker(nh_ctx* ctx, void* marker);
int nh_download(nh_ctx* ctx, void* host_dst, const void* dev_src, long long bytes);
int nh_memset(nh_ctx* ctx, void* dev, int byte, long long bytes);
/* dev[0..n) := values[0..n), n <= 8 64-bit words carried as kernel arguments: stream-ordered
 * like nh_upload, without the staging copy an upload from pageable memory costs (the
 * descriptors a run of the sampler re-points: where its chain history goes) */
int nh_set_words(nh_ctx* ctx, void* dev, const long long* values /*host*/, int n);
int nh_sync(nh_ctx* ctx);
/* pinned host staging + markers: nh_upload from a pinned buffer is truly asynchronous, so
 * the host can prepare half-step h+1 while the device runs h; a marker recorded after the
 * upload tells the host when the pinned buffer may be overwritten */
int nh_host_alloc(nh_ctx* ctx, long long bytes, void** host_out);
int nh_host_free(nh_ctx* ctx, void* host);
int nh_marker_create(nh_ctx* ctx, void** marker_out);
int nh_marker_record(nh_ctx* ctx, void* marker);
int nh_marker_wait(nh_ctx* ctx, void* marker);
int nh_marker_destroy(nh_ctx* ctx, void* marker);
/* HIP-event timing on the context's stream (used by bench.py for the roofline) */
int nh_timer_start(nh_ctx* ctx);
int nh_timer_stop(nh_ctx* ctx, double* elapsed_ms);
/* per-kernel accumulated HIP-event time since the last reset; kernel ids NH_K_* */
enum { NH_K_PDIST = 0, NH_K_INTEGRATE = 1, NH_K_SYNCHROTRON = 2, NH_K_TABLES = 3,
       NH_K_LNPROB = 4, NH_K_SSC = 5, NH_K_GLUE = 6 /* pack, move, scatter, lincomb */,
       NH_K_ROWS = 7 /* k_integrate_rows (We/Wp) */,
       NH_K_HALFSTEP = 8 /* k_half_step: the whole half-step in one launch */, NH_K_COUNT = 9 };
int nh_profile_enable(nh_ctx* ctx, int on);
/* HIP-event time of an empty kernel: the fixed cost an event pair adds per launch */
int nh_profile_calibrate(nh_ctx* ctx, int reps, double* overhead_us);
int nh_profile_read(nh_ctx* ctx, double* ms_per_kernel /*[NH_K_COUNT]*/,
                    long long* launches /*[NH_K_COUNT]*/, int reset);

/* ---- row 1: utils.py:285-355 trapz_loglog ------------------------------ */
/* out[r] = trapz_loglog(y[r][0:n], x[0:n]) for r < nrows (axis=-1 semantics:
 * zero nodes, NaN/sign-change -> log branch, |b+1| <= 1e-10 -> log branch). */
int nh_trapz_loglog(nh_ctx* ctx, const double* y, const double* x, int nrows, int n,
                    double* out);
/* the same with intervals=True (utils.py:350-351): out[row*(n-1) + i] = the term of
 * segment (x_i, x_{i+1}) */
int nh_trapz_loglog_intervals(nh_ctx* ctx, const double* y, const double* x, int nrows, int n,
                              double* out);

/* ---- rows 2,3: models.py eval statics + radiative.py:147-160,1002-1015 -- */
/* w[N][nG] = xg_i * unit_scale * f_kind(e_eV_i; params_w), dlw[i] = ln|w[i+1]/w[i]|.
 * unit_scale = mec2[eV] for electrons (1/eV -> 1/mec2, radiative.py:160) or 1e9
 * for protons (1/eV -> 1/GeV, radiative.py:1015).  n_out (optional, may be NULL)
 * receives the bare n = w/xg (what _nelec/_J return). */
int nh_particle_weights(nh_ctx* ctx, int kind, const double* params /*[N][NH_PD_NPAR]*/,
                        int N, const double* e_eV /*[nG]*/, const double* xg /*[nG]*/,
                        int nG, double unit_scale, double* w, double* dlw, double* n_out);

/* the same walkers on up to NH_MAX_GRIDS grids in ONE launch (the components of a model
 * evaluation use different particle grids) */
typedef struct { const double* e_eV; const double* xg; double* w; double* dlw;
                 double unit_scale; int nG; int pad;
                 const double* ln_e; /* ln e_eV[i], or NULL: taken per node */
                 const double* lx;   /* ln(xg[i+1]/xg[i]) (nh_grid_logratio), or NULL */
               } nh_grid;
#define NH_MAX_GRIDS 4
int nh_particle_weights_multi(nh_ctx* ctx, int kind, const double* params, int N,
                              const nh_grid* grids /*host [ngrids]*/, int ngrids);

/* lx[i] = ln(xg[i+1]/xg[i]), i < nG-1 (the abscissa ratios of utils.py:336) */
int nh_grid_logratio(nh_ctx* ctx, const double* xg, int nG, double* lx);

/* ---- the generic per-walker reduction (rows 1+6..10) --------------------- */
/* out[w*ldo + k] = scale[k] * trapz_loglog(n_w * K_k, xg)  for k < nK,
 * (summed over the nsplit planes, see below)
 * evaluated as sum_i  lx_i * (u2-u1)/ln(u2/u1),  u = w_i*Kt[i][k],
 * ln(u2/u1) = dlw[i] + dlnKt[i][k].
 * scale may be NULL (=1).  nonnegative != 0 promises Kt >= 0, w of one sign, and dlnKt
 * finite with dlnKt[i][k] >= 1e300 wherever Kt[i][k] or Kt[i+1][k] is zero (what every
 * nh_table_* builder writes; true for all emission tables except the FITPACK look-up
 * table, which rings below zero, and the Baring+99 fits): sign-change and zero-node
 * handling are then compiled out of the inner loop.  Replaces the trapz_loglog(nelec*gamint, gam) calls of
 * radiative.py:684 (IC), 949-953/966-970 (bremsstrahlung), 1530 (pion decay),
 * 165/193/1020/1053 (We, Wp). */
int nh_integrate_tables(nh_ctx* ctx, const double* w, const double* dlw, int N, int nG,
                        const double* lx, const double* Kt, const double* dlnKt, int nK,
                        const double* scale, double* out, int ldo, int nonnegative,
                        int nsplit);
/* nsplit > 1 cuts the abscissa into nsplit ranges handled by different workgroups, each
 * writing its partial sums to its own plane out + h*N*ldo (h < nsplit): the result is the
 * sum of the planes, which the consumer forms (nh_lnprob / nh_lincomb take the planes as
 * components).  It evens out the work per CU when (tiles x walkers) does not fill the
 * 256 CUs evenly; nh_integrate_tables_nsplit gives the recommended value. */
int nh_integrate_tables_nsplit(int N, int nG, int nK);

/* ---- row 5: radiative.py:282-342 Synchrotron._spectrum ------------------- */
/* out[w*ldo+k] = spectrum 1/(s eV) at photon energy E_eV[k] for field B_G[w*ldB]
 * (Gauss; ldB = 1 for a plain vector, NH_PD_NPAR when B rides in slot 7 of the packed
 * parameter rows); walker-dependent through both w/dlw and B. */
int nh_synchrotron(nh_ctx* ctx, const double* w, const double* dlw, const double* B_G, int ldB,
                   int N, const double* gam, const double* lx, int nG,
                   const double* E_eV, int nE, double* out, int ldo);

/* ---- rows 6,7: radiative.py:547-607 + G12/G34 345-367 -------------------- */
/* Kt[i][k] (1/s per unit ... as _iso/_ani_ic_on_planck return) for temperature
 * T_K; theta_rad < 0 selects the isotropic Eq.14, otherwise Eq.11.
 * scale[k] = uf*Eph_k/E_eV_k so that integrate_tables gives radiative.py:684-687 */
int nh_table_ic_planck(nh_ctx* ctx, const double* gam, int nG, const double* E_eV, int nE,
                       double T_K, double theta_rad, double* Kt, double* dlnKt, int ld);

/* ---- row 8: radiative.py:609-655 _iso_ic_on_monochromatic ---------------- */
/* seed_E_eV[ns]; ns == 1: seed_dens = energy density eV/cm3 (monochromatic);
 * ns > 1: differential photon density 1/(eV cm3), integrated with trapz_loglog
 * over the seed energies (radiative.py:638-640). */
int nh_table_ic_seed(nh_ctx* ctx, const double* gam, int nG, const double* E_eV, int nE,
                     const double* seed_E_eV, const double* seed_dens, int ns,
                     double* Kt, double* dlnKt, int ld);
/* walker-dependent seed density (SSC: examples/CrabNebula_SynSSC.py:29-31):
 * seed_dens[N][ns]; fused outer+inner reduction, out as nh_integrate_tables. */
int nh_ic_seed_walkers(nh_ctx* ctx, const double* w, const double* dlw, int N,
                       const double* gam, const double* lx, int nG,
                       const double* E_eV, int nE, const double* seed_E_eV,
                       const double* seed_dens /*[N][ns]*/, int ns, double* out, int ldo);

/* The same with the Aharonian-Atoyan kernel of radiative.py:620-637 tabulated: it depends on
 * (seed energy, gamma, photon energy) only, so a sampler builds it once -- nh_ssc_table fills
 * `table` (nh_ssc_table_bytes bytes of device memory: cfg4's grids 374 MB) -- and every step
 * runs the walkers' part alone.  Same result as nh_ic_seed_walkers. */
long long nh_ssc_table_bytes(int nG, int nE, int ns);
int nh_ssc_table(nh_ctx* ctx, const double* gam, int nG, const double* E_eV, int nE,
                 const double* seed_E_eV, int ns, void* table);
int nh_ic_seed_walkers_tab(nh_ctx* ctx, const double* w, const double* dlw, int N,
                           const double* gam, const double* lx, int nG,
                           const double* E_eV, int nE, const double* seed_E_eV,
                           const double* seed_dens /*[N][ns]*/, int ns, const void* table,
                           double* out, int ldo);

/* ---- row 10: radiative.py:838-989 Bremsstrahlung ------------------------- */
/* two tables: sigma_ee (rel/non-rel, 873-928) and sigma_ep = sigma_1 (838-849),
 * both in cm2/eV. */
int nh_table_brems(nh_ctx* ctx, const double* gam, int nG, const double* E_eV, int nE,
                   double* Kt_ee, double* dlnKt_ee, double* Kt_ep, double* dlnKt_ep, int ld);

/* ---- row 9: radiative.py:1215-1482 Kafexhiu+14, 1770-1797 LookupTable ---- */
/* Kt[i][k] = dsigma/dEgamma(Ep_i, Egamma_k) in cm2/GeV */
int nh_table_pion_analytic(nh_ctx* ctx, const double* Ep_GeV, int nG, const double* E_eV,
                           int nE, int hiE_model, int nuclear_enhancement,
                           double* Kt, double* dlnKt, int ld);
/* bicubic tensor-product B-spline (FITPACK bispev) with knots tx[ntx], ty[nty]
 * and coefficients c[(ntx-4)*(nty-4)], evaluated at (log10 Ep, log10 Egamma) */
int nh_table_pion_lut(nh_ctx* ctx, const double* Ep_GeV, int nG, const double* E_eV, int nE,
                      const double* tx, int ntx, const double* ty, int nty, const double* c,
                      double* Kt, double* dlnKt, int ld);

/* ---- row 11: core.py:64-94 lnprobmodel ----------------------------------- */
/* PionDecayKelner06._spectrum (radiative.py:1716-1767; Kelner, Aharonian & Bugayov 2006):
 * out[w*ldo + k] = differential luminosity 1/(s eV) for nh = 1 cm^-3 at photon energies
 * E_eV[k]: full calculation (Eq. 71, :1665-1684) at E >= Etrans_eV, delta-functional
 * approximation (:1693-1714) below, joined at Etrans by nhat (:1743-1748) when `mixed`
 * (the host sets it when energies lie on both sides).  params: the [N][8] particle rows of
 * nh_particle_weights (eV).  The reference's adaptive quad (epsrel 1e-3) is replaced by a
 * converged fixed Gauss-Legendre rule.  nhat_out[N], wp_TeV_out[N] (the `Wp` property,
 * :1716-1728, in TeV) may be NULL. */
int nh_pion_kelner06(nh_ctx* ctx, int kind, const double* params, int N, const double* E_eV,
                     int nE, double Etrans_eV, int mixed, double* out, int ldo,
                     double* nhat_out, double* wp_TeV_out);

/* model[w][k] = sum_j cscale[j] * comp_j[w*ldc + k]        (1/(s cm2 eV))
 * m' = model*conv[k] (SED<->differential factor, utils.py:219-282)
 * lnl[w] = -sum_{!ul} (m'-flux)^2/(2 sigma^2), sigma = err_hi if m'>flux else err_lo,
 *          + nviol*ln(1-cl[nviol]), nviol = #{ul : m' > flux}   (cl indexed by count).
 * model_out may be NULL.  ncomp <= 4. */
int nh_lnprobmodel(nh_ctx* ctx, const double* const* comps /*host array of dev ptrs*/,
                   const double* cscale /*host [ncomp]*/, int ncomp, int ldc, int N, int nE,
                   const double* conv, const double* flux, const double* err_lo,
                   const double* err_hi, const int* ul, const double* cl,
                   double* model_out, double* lnl);

/* ---- device-resident step loop ------------------------------------------- */
/* A lazy per-walker scalar: value[w] = a * tf(b * base[w*stride] + c); base NULL
 * means the constant a.  It lets the parameter transforms a naima model function
 * writes (10**pars[0] / u.eV, pars[3] * u.uG ...) be folded into the kernel that
 * consumes them, with pars living in HBM. */
enum { NH_TF_ID = 0, NH_TF_POW10 = 1, NH_TF_EXP = 2, NH_TF_LOG = 3, NH_TF_LOG10 = 4,
       NH_TF_SQRT = 5, NH_TF_SQUARE = 6, NH_TF_RECIP = 7 };
typedef struct { const double* base; long long stride; double a, b, c; int tf; int pad; } nh_lazy;
#define NH_MAX_LAZY 8
/* out[w*ld + j] = value_j[w], j < ncols <= NH_MAX_LAZY: builds the [N][NH_PD_NPAR]
 * parameter rows of nh_particle_weights (and B[N]) on the device */
int nh_pack_rows(nh_ctx* ctx, const nh_lazy* cols /*host [ncols]*/, int ncols, int N,
                 double* out, int ld);
enum { NH_OP_ADD = 0, NH_OP_SUB, NH_OP_MUL, NH_OP_DIV, NH_OP_POW, NH_OP_MAX, NH_OP_MIN,
       NH_OP_LT, NH_OP_LE, NH_OP_GT, NH_OP_GE };
int nh_ew_binary(nh_ctx* ctx, int op, const nh_lazy* x, const nh_lazy* y, int N, double* out);

/* a spectrum component: ptr[w*ld + k] * scale */
typedef struct { const double* ptr; long long ld; double scale; } nh_comp;
#define NH_MAX_COMP 8
/* out[w*ldo + k] = rowfac[w] * colfac[k] * sum_j comps[j]   (colfac, rowfac may be NULL):
 * rowfac carries a per-walker physical factor (target density n0 / nh, seed energy
 * density) that the reference applies as a scalar, radiative.py:684-687, 949-987, 1534 */
int nh_lincomb(nh_ctx* ctx, const nh_comp* comps /*host*/, int ncomp, const double* colfac,
               const nh_lazy* rowfac /*host*/, int N, int m, double* out, int ldo);

/* priors of core.py:34-58 on lazy scalars; lp[w] = sum of the terms */
enum { NH_PRIOR_UNIFORM = 0, NH_PRIOR_NORMAL = 1, NH_PRIOR_LOGUNIFORM = 2, NH_PRIOR_VALUE = 3 };
typedef struct { nh_lazy x; double p0, p1; int kind; int pad; } nh_prior;
#define NH_MAX_PRIOR 16
int nh_priors(nh_ctx* ctx, const nh_prior* terms /*host*/, int nterms, int N, double* lp);

/* core.py:97-121 in one launch: model = sum comps; lnl as nh_lnprobmodel;
 * p = lp[w] (NULL = 0) + sum of the prior terms (nterms may be 0);
 * total[w] = isinf(p) ? p : lnl + p. */
int nh_lnprob(nh_ctx* ctx, const nh_comp* comps /*host*/, int ncomp, int N, int nE,
              const double* conv, const double* flux, const double* err_lo,
              const double* err_hi, const int* ul, const double* cl, const double* lp,
              const nh_prior* terms /*host*/, int nterms,
              double* model_out /*[N][nE] or NULL*/, double* total);

/* stretch move on a device-resident ensemble coords[N][ndim], logp[N].  The host ships the
 * random numbers of many half-steps as one block, slice h (stride 3*ns doubles) =
 * { z[ns] | lnU'[ns] (float64) | S[ns] | partner[ns] (int32) }: S = active walkers,
 * partner = each one's partner in the complementary half.  cursor[0] (device int) selects
 * the slice; accept advances it (advance != 0), so a captured graph replays unchanged.
 * propose writes the block [lo, lo+nloc) of the proposals TRANSPOSED (qT[d][j]: pars[d] is
 * a contiguous vector over walkers); accept needs all ns new log-probabilities and
 * optionally leaves the slice's S in sel[ns] for nh_scatter_rows. */
int nh_move_propose(nh_ctx* ctx, const double* coords, const double* blk, const int* cursor,
                    int ns, int ndim, int lo, int nloc, double* qT, double* factors);
int nh_move_accept(nh_ctx* ctx, double* coords, double* logp, const double* blk, int* cursor,
                   const double* newlp, int ns, int ndim, int* accepted, int* naccepted,
                   int* sel, int advance);

/* nh_move_accept on gathered ROWS { lnprob | blob 0 | blob 1 ... } (width doubles each, row j
 * = proposal j of the slice): the blobs of an accepted proposal go to cur[b][walker] (m[b]
 * values each, consecutive in the row).  advance as nh_move_accept. */
int nh_move_accept_rows(nh_ctx* ctx, double* coords, double* logp, const double* blk, int* cursor,
                        const double* rows, int width, int ns, int ndim, int* accepted,
                        int* naccepted, int* sel, int advance, int nblobs,
                        double* const* cur /*host [nblobs]*/, const int* m /*host [nblobs]*/);

/* ---- the two launches that bracket a model evaluation in the device step loop ------
 * Slice protocol of these two: cursor[0] = index of the slice whose proposals are being
 * evaluated (-1 right after a block upload).
 *
 * nh_step_front (one block per proposed walker): proposes block [lo, lo+nloc) of slice
 * cursor[0]+1 into qT/factors, evaluates `packs` (the nh_pack_rows requests the model
 * made on its first evaluation; their lazy columns read qT or are constants) for the
 * proposed walkers, then the particle weights of nh_particle_weights_multi(kind, params,
 * nloc, grids) -- params must be one of the pack outputs -- and `moms`: single-row
 * reductions (We, Wp: nh_integrate_tables with nK = 1) over one of those grids.  When the
 * slice just accepted closed an ensemble step (cursor odd) the chain history row is
 * appended first.  The last block to finish advances the cursor.  `done` is a zeroed
 * device int the launch uses to find that block. */
typedef struct { nh_lazy cols[NH_MAX_LAZY]; int ncols; int ld; double* out; } nh_pack;
#define NH_MAX_PACK 4
/* chain history, DEVICE-resident (the host rewrites it between runs): row n receives
 * coords[N][ndim] and logp[N], n is incremented, while n < cap.  coords == NULL: off. */
typedef struct { double* coords; double* logp; long long n; long long cap; } nh_hist;
typedef struct { int grid; int pad; const double* Kt; const double* dlnKt; double* out; } nh_moment;
#define NH_MAX_MOMENT 4
int nh_step_front(nh_ctx* ctx, const double* coords, const double* logp, const double* blk,
                  int* cursor, int* done, int ns, int ndim, int lo, int nloc, double* qT,
                  double* factors, const nh_pack* packs /*host*/, int npacks, int kind,
                  const double* params, const nh_grid* grids /*host*/, int ngrids,
                  const nh_moment* moms /*host*/, int nmoms, nh_hist* hist /*device or NULL*/);

/* nh_lnprob followed, in the same launch, by the accept of nh_move_accept for walkers
 * [lo, lo+N) of slice cursor[0] (single-rank loop: every walker's new log-probability is
 * the one this launch just computed).  The cursor is not advanced (nh_step_front does). */
typedef struct { double* coords; double* logp; const double* blk; const int* cursor; int ns;
                 int ndim; int lo; int pad; int* accepted; int* naccepted; int* sel; } nh_accept;
int nh_lnprob_accept(nh_ctx* ctx, const nh_comp* comps /*host*/, int ncomp, int N, int nE,
                     const double* conv, const double* flux, const double* err_lo,
                     const double* err_hi, const int* ul, const double* cl, const double* lp,
                     const nh_prior* terms /*host*/, int nterms, double* model_out,
                     double* total, const nh_accept* mv /*host*/);

/* nh_synchrotron whose workgroups go on to evaluate nh_lnprob (+ nh_lnprob_accept's move
 * when mv != NULL) for their own walker: for models where the synchrotron spectrum is the
 * last component the likelihood waits for, the step loop saves a launch.  comps[syn_comp]
 * must be this launch's output (it is taken from LDS); nE <= 64. */
int nh_synchrotron_lnprob(nh_ctx* ctx, const double* w, const double* dlw, const double* B_G,
                          int ldB, int N, const double* gam, const double* lx, int nG,
                          const double* E_eV, int nE, double* out, int ldo,
                          const nh_comp* comps /*host*/, int ncomp, int syn_comp,
                          const double* conv, const double* flux, const double* err_lo,
                          const double* err_hi, const int* ul, const double* cl,
                          const double* lp, const nh_prior* terms /*host*/, int nterms,
                          double* total, const nh_accept* mv /*host or NULL*/);

/* the same for a table reduction that is the last producer (pi0 decay, a single-seed IC
 * model): nh_integrate_tables (one plane) whose workgroups finish their walkers' nh_lnprob
 * (+ accept when mv != NULL).  The spectrum has nK energies; comps[loc_comp] must be this
 * launch's output.  Shapes that cannot carry the epilogue (more than 64 columns, ...) are
 * run as the two launches. */
int nh_integrate_tables_lnprob(nh_ctx* ctx, const double* w, const double* dlw, int N, int nG,
                               const double* lx, const double* Kt, const double* dlnKt, int nK,
                               const double* scale, double* out, int ldo, int nonnegative,
                               const nh_comp* comps /*host*/, int ncomp, int loc_comp,
                               const double* conv, const double* flux, const double* err_lo,
                               const double* err_hi, const int* ul, const double* cl,
                               const double* lp, const nh_prior* terms /*host*/, int nterms,
                               double* total, const nh_accept* mv /*host or NULL*/);
/* dst[idx[lo+j]][0:m] = src[j][0:m] where accepted[lo+j] (accepted NULL = all) */
int nh_scatter_rows(nh_ctx* ctx, double* dst, int ldd, const double* src, int lds,
                    const int* idx, const int* accepted, int lo, int nloc, int m);
int nh_copy(nh_ctx* ctx, void* dev_dst, const void* dev_src, long long bytes);

/* side streams: independent pieces of one model evaluation (the radiative components)
 * run concurrently; under capture they become branches of the graph.  fork: side stream
 * `side` (0..3) starts after everything issued so far on the main stream and becomes
 * the current stream; wait: stream `waiter` waits for `producer`'s work so far (-1 =
 * main); join: the main stream waits for every forked side stream and becomes current
 * again (nh_sync, nh_download and nh_graph_end join implicitly). */
int nh_stream_fork(nh_ctx* ctx, int side);
/* as nh_stream_fork, but the side stream waits only for `marker` (nh_marker_record on the
 * main stream right after the launch whose output the branch consumes) */
int nh_stream_fork_at(nh_ctx* ctx, int side, void* marker);
int nh_stream_switch(nh_ctx* ctx, int side);
int nh_stream_wait(nh_ctx* ctx, int waiter, int producer);
int nh_stream_join(nh_ctx* ctx);

/* ---- the general electron path: a particle grid PER WALKER --------------------------------
 * Eemin / Eemax as per-walker values (fit parameters; the reference takes any keyword as
 * per-call state, radiative.py:280, 430): walker w integrates over
 *   gamma = logspace(log10(Eemin_w/mec2), log10(Eemax_w/mec2), max(10, int(nEed * decades)))
 * (radiative.py:147-154), built with its weights in the workgroup's LDS; the emission kernel is
 * evaluated at every (node, photon energy) -- no walker-independent table exists.
 *   what = 0: Synchrotron._spectrum (radiative.py:282-342) with B_G per walker,
 *             out[w*ldo + k] in 1/(s eV)
 *   what = 1: InverseCompton on nseed thermal seed fields (radiative.py:547-607), seed s at
 *             out[w*ldo + s*nE + k] = trapz_loglog(nelec sigma_s, gamma); the caller applies
 *             uf * Eph / E (radiative.py:684-687); seed_theta[s] < 0: isotropic.  T and theta
 *             are lazy scalars: a seed temperature / angle may itself be a fit parameter (a
 *             walker with T <= 0 gets the NaN the reference's arithmetic gives)
 *   what = 2: We = trapz_loglog(gamma nelec, gamma mec2) over the walker's grid, erg
 *             (radiative.py:162-195: We, compute_We with per-walker limits); out[w*ldo]
 *   what = 3: Bremsstrahlung (radiative.py:838-989): out[w*ldo + k] = trapz_loglog(nelec
 *             sigma_ee, gamma), out[w*ldo + nE + k] the same with sigma_1 (electron-ion), per
 *             eV; the caller applies n0 c and the abundance weights (radiative.py:949-987)
 * The limits arrive in the unit the caller's Quantity carries, with that unit's value in erg
 * beside them: the kernel forms gamma_min = (Eemin / mec2[erg]) * unit_erg exactly as the host
 * path does (what astropy reduces Eemin / mec2 to), so both paths take log10 of the same double.
 * nEed is a lazy scalar too (nodes per decade per walker).
 * nmax: grid nodes the workgroup's LDS is sized for (4 nmax doubles); a walker that needs more
 * gets NaN and status[0] (device ints, zeroed by the caller) receives the largest count asked
 * for.  status[1] counts evaluations whose nEed * decades lay within 1e-9 of an integer: there
 * int() may differ between this log10 and numpy's in the last place -- reported, not silent. */
int nh_general_electron(nh_ctx* ctx, int kind, const double* rows /*[N][NH_PD_NPAR]*/, int N,
                        const nh_lazy* Eemin /*host*/, double Eemin_unit_erg,
                        const nh_lazy* Eemax /*host*/, double Eemax_unit_erg,
                        const nh_lazy* nEed /*host*/, int what, const nh_lazy* B_G /*host, what = 0*/,
                        const nh_lazy* seed_T_K /*host*/, const nh_lazy* seed_theta /*host*/,
                        int nseed, const double* E_eV, int nE, double* out, int ldo, int nmax,
                        int* status /*device, 2 ints*/);
/* the same over every walker's own grid for ONE monochromatic (ns = 1: seed_dens its energy
 * density, eV/cm3) or tabulated (seed_dens in 1/(eV cm3) at the ns energies seed_E, eV) isotropic
 * seed field that all walkers share -- InverseCompton._calc_specic's inner trapz_loglog over the
 * seed's energies (radiative.py:609-655) at every (node, photon energy):
 * out[w*ldo + k] = trapz_loglog(nelec sigma_seed, gamma); the caller applies Eph / E
 * (radiative.py:684-687) */
int nh_general_electron_seed(nh_ctx* ctx, int kind, const double* rows /*[N][NH_PD_NPAR]*/, int N,
                             const nh_lazy* Eemin /*host*/, double Eemin_unit_erg,
                             const nh_lazy* Eemax /*host*/, double Eemax_unit_erg,
                             const nh_lazy* nEed /*host*/, const double* seed_E /*device*/,
                             const double* seed_dens /*device*/, int ns,
                             const double* E_eV /*device*/, int nE, double* out, int ldo, int nmax,
                             int* status);
/* The same with a photon density PER WALKER, seed_dens[w * seed_ld + s] (seed_ld >= ns): a
 * synchrotron-self-Compton seed -- each walker's own synchrotron photons,
 * examples/CrabNebula_SynSSC.py:29-45 -- combined with Eemin / Eemax / nEed per walker
 * (InverseCompton takes any keyword per call, radiative.py:430; inner integral :609-655). */
int nh_general_electron_seed_rows(nh_ctx* ctx, int kind, const double* rows, int N,
                                  const nh_lazy* Eemin, double Eemin_unit_erg, const nh_lazy* Eemax,
                                  double Eemax_unit_erg, const nh_lazy* nEed, const double* seed_E,
                                  const double* seed_dens, long long seed_ld, int ns,
                                  const double* E_eV, int nE, double* out, int ldo, int nmax,
                                  int* status);

/* The same for protons: Epmin / Epmax / nEpd per walker (radiative.py:1002-1055, 1495-1536).
 * Walker w integrates over Ep = logspace(log10 Epmin_w, log10 Epmax_w, max(10, int(nEpd_w *
 * decades))) GeV, decades = log10(Epmax / Epmin) (count_mode 0: the spectrum's grid,
 * radiative.py:1002-1009) or log10 Epmax - log10 Epmin (count_mode 1: compute_Wp's,
 * :1047-1053); the limits arrive in the caller's unit with that unit in GeV beside them.
 *   what = 0: out[w*ldo + k] = trapz_loglog(dsigma/dE(Ep, E_k) J, Ep), the Kafexhiu+14 cross
 *             section evaluated at every (node, energy) (hiE: NH_PP_*, nuc: nuclear enhancement);
 *             the caller applies nh c (radiative.py:1530-1536)
 *   what = 1: the same with the look-up table's FITPACK spline (tx, ty, cf as for
 *             nh_table_pion_lut) evaluated at every (node, energy)
 *   what = 2: out[w*ldo] = Wp = trapz_loglog(Ep J, Ep), GeV
 * nmax / status as for nh_general_electron. */
int nh_general_proton(nh_ctx* ctx, int kind, const double* rows /*[N][NH_PD_NPAR]*/, int N,
                      const nh_lazy* Epmin /*host*/, double Epmin_unit_GeV,
                      const nh_lazy* Epmax /*host*/, double Epmax_unit_GeV,
                      const nh_lazy* nEpd /*host*/, int count_mode, int what, int hiE, int nuc,
                      const double* tx, int ntx, const double* ty, int nty, const double* cf,
                      const double* E_eV, int nE, double* out, int ldo, int nmax,
                      int* status /*device, 2 ints*/);

/* ---- ONE launch per half-step ---------------------------------------------------------
 * nh_step_front + every table reduction of the model + its synchrotron component +
 * nh_lnprob (+ the accept of nh_lnprob_accept when do_accept) for the proposed walkers
 * [lo, lo+nloc) of slice cursor[0]+1, one workgroup per walker: emcee's
 * StretchMove.get_proposal / RedBlueMove.propose around core.py:97-121 around
 * radiative.py:282-342, 657-710, 1495-1536.  The particle weights stay in LDS (they are
 * also written to grids[].w/.dlw when write_weights), every spectrum is written to its
 * `out` as the separate entry points would.
 * Slices: nh_half_step_begin_block tells the plan that a new block of moves sits in `blk`;
 * the k-th launch after it proposes, evaluates and (do_accept) accepts slice first_slice + k -- the plan
 * counts its own launches on the device, nothing has to be advanced by the caller.  The
 * history row of a closed ensemble step (row steps_before + k/2 - 1 of hist, hist->n is not
 * used) is written by the NEXT launch of the same block of moves; after the last half-step
 * of a block, and whenever the chain may be read, the caller writes it with nh_hist_append
 * (writing a row twice is harmless).  The descriptor is copied once (nh_half_step_create);
 * a launch takes no other argument, so it can be captured into a hipGraph and replayed. */
/* KD: the emission table of nh_integrate_tables in the interleaved layout the kernel streams,
 * KD[(i*nK + k)*2 + {0,1}] = {Kt[i][k], dlnKt[i][k]} (nh_table_interleave).  Where column k
 * changes sign between nodes i and i+1 the second entry is NaN: that segment takes the
 * reference's log branch (NaN exponent b, utils.py:336-345) -- decided once per table, the
 * sign pattern does not depend on the walker.  A table declared `nonnegative` to the half-step
 * plan is built WITH lx = ln(x[i+1]/x[i]) of its grid: its second entries are dlnKt / lx (the
 * kernel then integrates a segment as (u2 - u1) / (dl / lx), utils.py:336-339 without the
 * multiplication); other tables with lx = NULL. */
typedef struct { int grid; int nK; int ldo; int nonnegative;
                 const double* KD; const double* reserved; const double* scale /*[nK] or NULL*/;
                 double* out /*[nloc][ldo]*/; } nh_hs_table;
int nh_table_interleave(nh_ctx* ctx, const double* Kt, const double* dlnKt,
                        const double* lx /*[nG-1] device, or NULL*/, int nG, int nK,
                        double* KD);
typedef struct { int grid /* -1: no synchrotron component */; int nE; int ldo;
                 int bcol /* column of the particle rows that carries B [G], or -1 */; int ldB;
                 int n1 /* 0, or: energies [n1, nE) go to out2 -- two Synchrotron.flux calls of one
                           model evaluation (CrabNebula_SynSSC.py:29, 45) as ONE component */;
                 const double* E_eV; const double* B /* [nloc*ldB] when bcol < 0 */;
                 double* out /*[nloc][ldo]*/; double* out2 /*[nloc][ldo2] == out + nloc * ldo, or NULL*/; int ldo2; int pad2;
               } nh_hs_syn;
#define NH_HS_MAX_TAB 4
/* a blob the model function returns besides its flux (core.py:103-113; emcee keeps the blobs
 * of the ACCEPTED position of every walker): kind 0 = the model spectrum itself (the sum of
 * `comps`, nE values), kind 1 = lazy(result of single-row reduction `mom`) (We, Wp).  On
 * accept the launch writes the row into cur[walker]; with the coordinates it appends the
 * ensemble's rows to the history at *hist (a device word holding the history's base, 0: off). */
typedef struct { int kind; int mom; int m; int pad; nh_lazy lazy;
                 double* cur /*[N][m]*/; double* const* hist /*device*/; } nh_hs_blob;
#define NH_HS_MAX_BLOB 4
typedef struct {
  double* coords; double* logp; const double* blk;
  int* cursor; /* out: the slice of the launch, for nh_move_accept(advance = 0) after it */
  int* reserved;
  int ns, ndim, lo, nloc;
  double* qT; double* factors; nh_hist* hist /* device, or NULL */;
  int* accepted; int* naccepted; int* sel;
  int do_accept;      /* 1: single rank, the launch accepts; 0: total[] only (sharded loop) */
  int write_weights;  /* also store w / dlw in grids[].w / .dlw */
  nh_pack packs[NH_MAX_PACK]; int npacks;
  int kind; const double* params;
  nh_grid grids[NH_MAX_GRIDS]; int ngrids;
  nh_moment moms[NH_MAX_MOMENT]; int nmoms;
  nh_hs_table tab[NH_HS_MAX_TAB]; int ntab;
  nh_hs_syn syn;
  nh_comp comps[NH_MAX_COMP]; int ncomp; int nE;
  const double* conv; const double* flux; const double* err_lo; const double* err_hi;
  const int* ul; const double* cl; const double* lp /* [nloc] or NULL */;
  nh_prior terms[NH_MAX_PRIOR]; int nterms;
  double* model_out /* [nloc][nE] or NULL */; double* total /* [nloc] */;
  nh_hs_blob blobs[NH_HS_MAX_BLOB]; int nblobs;
  /* do_accept = 0 (sharded loop) with blobs: total[] is an exchange buffer of rows of
   * send_width doubles, row j = { lnprob | blob 0 | blob 1 ... } of walker j: ONE all-gather
   * per half-step carries the log-probabilities and the blobs; nh_move_accept_rows takes the
   * gathered rows.  0: total[] holds the log-probabilities only. */
  int send_width;
} nh_hs_desc;
typedef struct nh_halfstep_plan nh_halfstep_plan;
int nh_half_step_create(nh_ctx* ctx, const nh_hs_desc* desc /*host*/, nh_halfstep_plan** out);
int nh_half_step_begin_block(nh_ctx* ctx, nh_halfstep_plan* plan, int first_slice,
                             int steps_before);
/* slice >= 0: the launch works on that slice of the block of moves (a captured graph replays
 * it for the same slice); slice < 0: the slice the plan's own launch counter says is next */
int nh_half_step_launch(nh_ctx* ctx, nh_halfstep_plan* plan, int slice);
/* Which launches of a half-step open and close a span of the context's device clock
 * (nh_clock_read).  Default: every launch of the plan is a span by itself.  A half-step of
 * several launches -- a model that asks for its synchrotron spectrum twice with the SSC seed
 * integral in between, examples/CrabNebula_SynSSC.py:29-45 -- opens with its first launch
 * (1, 0) and closes with its last (0, 1). */
int nh_half_step_span(nh_halfstep_plan* plan, int open, int close);
int nh_half_step_info(const nh_halfstep_plan* plan, int* threads, int* blocks,
                      long long* lds_bytes);
/* workgroups per walker of the plan's launches: 1, or K = 2, 4, 8 where a launch holds fewer
 * walkers than the device has compute units and the model's work items are most of a launch
 * (each of the K workgroups repeats the prologue and takes every K-th work item; the last to
 * arrive sums the partial spectra in index order and evaluates the likelihood: the results do
 * not depend on arrival order).  NH_HS_SPLIT=<K> in the environment caps K (1: never split). */
int nh_half_step_split(const nh_halfstep_plan* plan, int* split);
/* How the plan's launches integrate the synchrotron component (radiative.py:282-342): *form = 0 no
 * such component, 1 the direct form (nh_syn.h), 2 the log-domain form on the grid's comb
 * (nh_syn2.h: a log-uniform particle grid whose table fits in LDS beside the model's). */
int nh_half_step_syn_form(const nh_halfstep_plan* plan, int* form);
/* NaN log-probabilities the accepts of the separate kernels (nh_lnprob / ..._lnprob with a move,
 * nh_move_accept, nh_move_accept_rows) have met since the last reset: emcee raises
 * ValueError("Probability function returned NaN") at the first one (EnsembleSampler.
 * compute_log_prob; reference call site core.py:128); a launch rejects the proposal -- NaN
 * compares false -- and counts.  The one-launch kernels count per plan (below). */
int nh_nan_count(nh_ctx* ctx, int reset, int* count);
/* The device span clock: what the step loop's launches -- the kernels that replace emcee's
 * EnsembleSampler.sample loop around core.py:97-121 (reference call sites core.py:128, 450-457)
 * -- have spent ON the device since the last reset, measured on those launches themselves: the
 * first workgroup of a span's first kernel stamps the GPU's constant-rate wall clock, the last
 * workgroup out of the span's last kernel adds the difference (a span = one launch of the
 * resident loop with its epilogue, or the launches of one half-step of the per-launch loop).
 * out[3] = { ticks inside closed spans, closed spans, ticks per millisecond }.  Spans lie inside
 * the host interval around their launch calls and do not overlap, so (host time of a region) -
 * (span time) >= 0 by construction: bench.py's region_overhead_us.  Synchronises the stream;
 * reading adds nothing to any launch. */
int nh_clock_read(nh_ctx* ctx, int reset, long long* out);
/* proposals whose log-probability was NaN since the plan was created (or the last reset):
 * emcee raises ValueError("Probability function returned NaN") on the first one
 * (EnsembleSampler.compute_log_prob; reference call site core.py:128), a launch rejects the
 * proposal and counts it here for the caller to act on.  Synchronises the stream. */
int nh_half_step_nan_count(nh_ctx* ctx, nh_halfstep_plan* plan, int reset, int* count);
/* The same with, beside it, the proposals the prior forbade (core.py:99-101, 115-119: the
 * launch evaluates none of their integrals; the reference evaluates the model and discards the
 * result).  A negative *nan / *forbidden on entry SETS that counter to -value - 1 first (a
 * replayed block of moves must not count its proposals twice). */
int nh_half_step_counts(nh_ctx* ctx, nh_halfstep_plan* plan, int reset, int* nan, int* forbidden);
int nh_half_step_destroy(nh_ctx* ctx, nh_halfstep_plan* plan);

/* ---- a whole block of moves in ONE launch: the half-step kernel with resident workgroups ----
 * Replaces the sequence of nh_half_step_launch calls for slices [slice0, slice0 + nslices) of
 * the block of moves in `blk` (whole ensemble steps: slice0 and nslices even, at most 32 steps)
 * -- emcee's EnsembleSampler.sample loop over StretchMove / RedBlueMove.propose around
 * core.py:97-121 (reference call sites core.py:128, 450-457).  The workgroups stay resident
 * for the whole launch; the ensemble-wide barrier between half-steps becomes a per-walker
 * hand-off: a walker of slice h + 1 waits only for its own and its partner's record of the
 * previous slices (data-tagged 8-byte granules written write-through by the workgroup that
 * moved the walker; see nh_persist.hip).  Everything walker-independent (grid nodes, table
 * of exponentials, data columns, priors) is loaded into LDS once per launch.
 * Needs a plan with do_accept = 1, one workgroup per walker, lo = 0, nloc = ns, ndim <= 15, no
 * separately evaluated prior (lp == NULL) and every likelihood component produced inside the
 * launch; nh_half_step_run_create says no otherwise (the caller keeps launching per half-step).
 * The launch reads coords / logp as they are and leaves them at the state after the last
 * slice; naccepted is advanced; the spectra / weights / parameter-row outputs of the plan's
 * descriptor are NOT written.  Chain history: hist_coords [cap][N][ndim], hist_logp [cap][N]
 * and hist_blobs[b] [cap][N][m_b] (device; hist_coords NULL: none kept) receive rows hist_row0,
 * hist_row0 + 1, ... for the steps of the launch; the blobs' current values (blobs[].cur) end at
 * the last step's.  A launch whose workgroups cannot all be resident would wait for ever: the
 * grid is sized by the occupancy query, every wait is bounded, and a time-out aborts the
 * launch and latches an error that nh_half_step_run_status reports (0 = every launch so far
 * found its records). */
typedef struct nh_halfstep_run nh_halfstep_run;
int nh_half_step_run_create(nh_ctx* ctx, nh_halfstep_plan* plan, nh_halfstep_run** out);
int nh_half_step_run(nh_ctx* ctx, nh_halfstep_plan* plan, nh_halfstep_run* run, int slice0,
                     int nslices, double* hist_coords, double* hist_logp,
                     double* const* hist_blobs /*host array of device pointers, or NULL*/,
                     long long hist_row0, long long hist_cap);
int nh_half_step_run_status(nh_ctx* ctx, nh_halfstep_run* run, int* status);
/* What launch number `launch` of the loop (1, 2, ...) left behind, read from page-locked host
 * memory its epilogue wrote -- no stream operation, no synchronisation: *done = 0 while it has not
 * ended (wait != 0: sleeps until it has), else its status (0 = it found every record; -1 = the
 * report has been overwritten: only the last two launches have one) and the plan's counters of
 * NaN log-probabilities and of proposals the prior forbids (core.py:99-119) as they stood behind
 * it.  The sampler queues launch n + 1 only once launch n - 1 is known to have ended well, so a
 * launch that gave up costs the replay of two blocks of moves at most.  One-GPU loops only. */
int nh_half_step_run_report(nh_ctx* ctx, nh_halfstep_run* run, int launch, int wait, int* done,
                            int* status, int* nan_count, int* forbidden,
                            int* before /*[2]: the two counters as the launch found them, or NULL*/);
/* The resident loop's own copies of the plan's emission tables with their columns SORTED by the
 * first grid row in which they are non-zero (an inverse-Compton or pi0 table is zero below the
 * kinematic threshold gamma = E / mec2, Ep = E ...: rows that contribute exact zeros to
 * trapz_loglog, utils.py:347-348).  kds[t] (device): the interleaved table [nG][nK][2] of plan
 * table t with permuted columns (nh_table_interleave of the permuted Kt / dlnKt), followed by a
 * trailer of ints { row0[8] | perm[nK] }: row0[tile] = first row in which any of the tile's 64
 * columns is non-zero (segments below it are not walked), perm[p] = the column of the spectrum
 * that position p holds.  NULL keeps the plan's table.  The caller keeps the buffers alive.  The
 * trailers are read back and checked (rows in range, the column order a permutation). */
int nh_half_step_run_tables(nh_ctx* ctx, nh_halfstep_plan* plan, nh_halfstep_run* run,
                            const double* const* kds /*host*/, int ntab);
/* ---- the resident loop over an ensemble SHARED by the GPUs of a node (2 .. 8 ranks, one process
 * per GPU).  Replaces, for walkers sharded as in the per-launch loop (rank r proposes positions
 * [lo, lo + nloc) of every half-step; the reference's analogue is Pool(threads) over a fixed
 * ensemble, core.py:446-457), the all-gather between the launches of a half-step: a mover stores
 * its walker's record into EVERY rank's ring (system-scope stores; xGMI for the others), consumers
 * poll their local ring as on one GPU.  No collective and no launch per half-step.
 *   create_shared: plan with lo / nloc = this rank's block (do_accept not needed), rings in
 *                  fine-grained device memory;
 *   export / attach: the 64-byte hipIpc handle of a rank's rings, to be carried to every other
 *                  rank by the caller's control plane and attached there (all before the first run);
 *   probe:         `rounds` tagged exchanges with every peer inside ONE launch -- status 0 only if
 *                  stores another GPU makes while the kernel runs reach its polling loads (call on
 *                  every rank at the same time; bounded like every wait of the loop);
 *   hist_flags:    where nh_half_step_run keeps who-moved-what of the following launches,
 *                  [hist_cap][N] ints, rows hist_row0 ... of every launch: -1 = another rank
 *                  moved the walker in that step, 0 / 1 = this rank did and rejected / accepted
 *                  (history rows and blobs are written by the mover only, on its own GPU; blob
 *                  rows of rejected moves are NOT filled);
 *   counters:      nacc_own[N] (moves this rank accepted; the plan's naccepted is not touched) and
 *                  curstamp[N] (-1, or the step count at which this rank last accepted a move of
 *                  the walker: whoever holds the largest stamp holds the walker's current blobs in
 *                  its blobs[].cur); reset bit 0 / 1 clears the first / second afterwards.
 * After a shared launch coords / logp hold the whole ensemble on every rank. */
int nh_half_step_run_create_shared(nh_ctx* ctx, nh_halfstep_plan* plan, int rank, int nrank,
                                   nh_halfstep_run** out);
int nh_half_step_run_export(nh_ctx* ctx, nh_halfstep_run* run, void* handle64);
int nh_half_step_run_attach(nh_ctx* ctx, nh_halfstep_run* run, int peer, const void* handle64);
int nh_half_step_run_probe(nh_ctx* ctx, nh_halfstep_run* run, int rounds, int* status,
                           double* us_per_round);
int nh_half_step_run_hist_flags(nh_halfstep_run* run, int* flags);
int nh_half_step_run_counters(nh_ctx* ctx, nh_halfstep_run* run, int* nacc_own, int* curstamp,
                              int reset);
int nh_half_step_run_info(const nh_halfstep_run* run, int* grid, int* threads,
                          long long* lds_bytes);
/* How the loop evaluates Synchrotron._spectrum's integrand (radiative.py:300-340): mode 1 = in
 * the log domain on the grid's comb (csrc/nh_syn2.h: one exponent per node, ln Gtilde from a
 * table of `pieces` degree-5 pieces of `nodes_per_piece` grid steps; taken when the particle
 * grid is log-uniform, as radiative.py:147-154 makes it), 0 = the direct form (any grid; no
 * synchrotron component). */
int nh_half_step_run_syn_info(const nh_halfstep_run* run, int* mode, int* nodes_per_piece, int* pieces);
/* Where the loop's table work items (the trapz_loglog over the particle grid of radiative.py:684,
 * 1534-1536) read their rows: in_registers = 1 when the loop runs the instance of a table-only
 * model whose items keep their rows of {K, dlnK} in vector registers for the whole launch
 * (workgroups of 512 threads; an emission table does not depend on the walker), `nodes_max` the
 * rows per lane that instance can hold; 0 = streamed from the L2s every half-step. */
/* workgroups per walker of the resident loop's launches (the plan's split) and whether they
 * divide the grid's ROWS between them (table-only models: core.py:450-457's one evaluation per
 * walker on two compute units, each with half of radiative.py:1495-1536's proton grid) */
int nh_half_step_run_split_info(const nh_halfstep_run* run, int* split, int* rows);
/* *deep = 1 when the loop's launches keep TWO walkers of a workgroup in flight (ensembles of more
 * walkers per half-step than resident workgroups: a workgroup's next walker has its records,
 * proposal and parameter packs made while the current one's work items run, and its weights
 * start while the current one's likelihood -- core.py:64-121 -- is still being summed) */
int nh_half_step_run_pipeline_info(const nh_halfstep_run* run, int* deep);
int nh_half_step_run_table_info(const nh_halfstep_run* run, int* in_registers, int* nodes_max);
/* NH_HS_DEBUG=1: out[256][64][8] wall-clock stamps (100 MHz) of the last launch, per
 * (workgroup, slice handled): start | records in | packs done | weights done | own items
 * done | all items done | spectra summed | record published; then [64][4][16]: for workgroup
 * 0, per slice handled, when each of its waves reached barrier 1 | 2 | 3 (| unused) */
int nh_half_step_run_stamps(nh_ctx* ctx, const nh_halfstep_run* run, long long* out);
int nh_half_step_run_destroy(nh_ctx* ctx, nh_halfstep_run* run);
/* diagnostics (plans created under NH_HS_DEBUG=1): per-phase 100 MHz wall-clock stamps of the
 * first 8 workgroups of the last launch, out[8][16], followed by 4 x 16 per-wave figures of
 * workgroup 0, then start[1024] and end[1024] stamps of every workgroup (2304 values in all);
 * all zero otherwise */
int nh_half_step_stamps(nh_ctx* ctx, const nh_halfstep_plan* plan, long long* out);
/* row `row` (-1: hist->n - 1) of the device-resident history := coords[N][ndim] / logp[N] */
int nh_hist_append(nh_ctx* ctx, const double* coords, const double* logp, long long N, int ndim,
                   const nh_hist* hist /*device*/, long long row);
/* the same for the blobs a half-step plan keeps: row `row` of every blob history := cur */
int nh_half_step_append_blobs(nh_ctx* ctx, const nh_halfstep_plan* plan, long long row);

/* host-side generator of the move's random numbers: a worker thread fills a ring of
 * (page-locked) blocks ahead of the consumer, in exactly the slice layout above.  The
 * stream (xoshiro256** from `seed`; per step: permutation, z, partners, ln U') does not
 * depend on how many steps are taken at once, nor on the rank. */
typedef struct nh_moves nh_moves;
int nh_moves_create(unsigned long long seed, int N, double a, int ksteps_per_block, int depth,
                    int pinned, nh_moves** out);
/* up to `want` consecutive steps, contiguous in host memory (2*got slices); depth >= 3.
 * A used-up block goes back to the producer one block late: an asynchronous copy out of
 * the MOST RECENT take may still be pending at the next take, all earlier ones must be done. */
int nh_moves_take(nh_moves* m, int want, const void** host_ptr, int* got);
int nh_moves_destroy(nh_moves* m);

/* capture everything launched on the context's stream into a hipGraph, replay it */
int nh_graph_begin(nh_ctx* ctx);
int nh_graph_end(nh_ctx* ctx, void** graph_exec_out);
int nh_graph_launch(nh_ctx* ctx, void* graph_exec);
int nh_graph_destroy(nh_ctx* ctx, void* graph_exec);

/* ---- multi-GPU: one all-gather per half-step -------------- */
#define NH_UNIQUE_ID_BYTES 128
/* NH_OK when librccl could be loaded and every entry point resolved (no communicator is made).
 * Every rank asks this first and the answers are min-reduced over the control plane, so that no
 * rank enters the blocking ncclCommInitRank while another one never will. */
int nh_comm_available(void);
int nh_comm_unique_id(char* id_out /*[NH_UNIQUE_ID_BYTES]*/);
int nh_comm_init(nh_ctx* ctx, int rank, int nranks, const char* id);
int nh_comm_destroy(nh_ctx* ctx);
/* what the live communicator says of itself (ncclCommCount, ncclCommUserRank, ncclCommCuDevice):
 * the workers the reference's Pool(threads) would report, core.py:446-457 */
int nh_comm_info(nh_ctx* ctx, int* nranks, int* rank, int* device);
/* recv[r*count .. (r+1)*count) = send of rank r (device buffers, float64 count) */
int nh_comm_allgather(nh_ctx* ctx, const double* send, double* recv, long long count);

#ifdef __cplusplus
}
#endif
#endif /* NAIMA_HIP_H */
