/*
 * fluca_hip.h -- C-ABI of libflucahip.so: the MI355X (gfx950) replacement for the
 * pressure-Poisson path of Fluca's PCABF preconditioner.
 *
 * Plain C, no PETSc / torch / C++ types.  Every function returns int: 0 = success,
 * negative = -(PETSC_ERR_* class) so a PETSc-side caller can SETERRQ(-rc).
 * Device pointers are raw HIP device addresses owned by the caller unless stated.
 * A handle is driven by one host thread; different handles (one per GPU / process, or several
 * per process on several host threads) are independent: nothing of a solve lives in a global
 * (tests/test_gpu_config5.py drives eight handles from eight threads).  The only process-wide
 * state are the tuning knobs below (atomics) and the lazily loaded RCCL entry points (guarded).
 * All arithmetic is IEEE fp64 (PetscScalar = double).
 *
 * Reference interfaces replaced (file:line relative to thecasterian/fluca):
 *   fl_poisson_create    PCSetUp_ABF: S = D((-T)Ainv G - (-R)), Ainv = ID         fluca/src/ns/utils/abfpc/abfpc.c:113-182
 *                        + D / Gst row definitions                                 fluca/src/ns/impl/linearcn/cnlinearcart3d.c:2314-2600
 *                        + Mesh_Cart decomposition (N, nRanks, ownership l[])      fluca/include/fluca/private/meshcartimpl.h:8-17
 *   fl_poisson_apply     MatMult(S, x, y) inside KSPSolve(kspS)                    abfpc.c:77,180
 *   fl_poisson_solve     KSPSolve(abf->kspS, abf->Srhs, p)                         abfpc.c:77
 *   fl_poisson_rhs       MatMult(D, Vstar, Srhs); VecAYPX(Srhs, -1, contrhs)       abfpc.c:75-76
 *   fl_poisson_project   stage 2: v = v* - G p ; V = V* + (-T)Gp - (-R)p           abfpc.c:80-101
 *   fl_poisson_gst_bc    ComputeStaggeredPressureGradientBoundaryConditionVector   cnlinearcart3d.c:2602-2805
 *   fl_pressure_update   p = phalf + 1.5 dp ; phalf += dp (first step p0 + 2dp)    cnlinearcart3d.c:2846-2854
 *   fl_ksp_opts          -ns_abf_schur_ksp_* / -ns_abf_schur_pc_type options       abfpc.c:42,206,248-249
 *   fl_bc                NSBoundaryConditionType                                   fluca/include/flucansbc.h:5-11
 *   fl_abf_apply         PCApply_ABF, both stages                                          abfpc.c:48-111
 *   fl_momentum_*        A = I + dt C - (mu dt / 2 rho) L and KSPSolve(kspA)                   cnlinearcart3d.c:425-632,873-1294,2930-2941; abfpc.c:72
 *   fl_ibm_*             no reference counterpart (THEORY_GUIDE.md:130-132 is a TODO); specified in DESIGN.md
 *
 * Array layouts (x fastest, block-contiguous per rank, exactly one rank's OWNED part):
 *   cell  (i,j,k) -> (k*ny + j)*nx + i                     nx,ny,nz = decomp.len (or grid.n)
 *   x-face(i,j,k) -> (k*ny + j)*fx + i ,  i in [0,fx)      fx = nx+1 on the last rank of a non-periodic axis, else nx
 *   y-face(i,j,k) -> (k*fy + j)*nx + i ,  fy likewise      (DMStag: the extra face belongs to the last rank)
 *   z-face(i,j,k) -> (k*ny + j)*nx + i ,  k in [0,fz)
 *   Face f of an axis lies between cells f-1 and f.
 */
#ifndef FLUCA_HIP_H
#define FLUCA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: -(PETSC_ERR_*) ------------------------------------------------------------ */
#define FL_SUCCESS 0
#define FL_ERR_MEM (-55)            /* PETSC_ERR_MEM */
#define FL_ERR_SUP (-56)            /* PETSC_ERR_SUP */
#define FL_ERR_ARG_SIZ (-60)        /* PETSC_ERR_ARG_SIZ */
#define FL_ERR_ARG_WRONG (-62)      /* PETSC_ERR_ARG_WRONG */
#define FL_ERR_ARG_OUTOFRANGE (-63) /* PETSC_ERR_ARG_OUTOFRANGE */
#define FL_ERR_ARG_WRONGSTATE (-73) /* PETSC_ERR_ARG_WRONGSTATE */
#define FL_ERR_LIB (-76)            /* PETSC_ERR_LIB (RCCL) */
#define FL_ERR_ARG_NULL (-85)       /* PETSC_ERR_ARG_NULL */
#define FL_ERR_NOT_CONVERGED (-91)  /* PETSC_ERR_NOT_CONVERGED */
#define FL_ERR_GPU (-97)            /* PETSC_ERR_GPU */

/* ---- types ----------------------------------------------------------------------------------- */

/* = NSBoundaryConditionType, flucansbc.h:5-11.  Index order of bc[6]: 0 left(-x) 1 right 2 down(-y) 3 up 4 back(-z) 5 front
 * (MeshCartGetBoundaryIndex, fluca/src/mesh/impl/cart/cart.c:564-591). */
typedef enum { FL_BC_NONE = 0, FL_BC_VELOCITY = 1, FL_BC_PRESSURE_OUTLET = 2, FL_BC_PERIODIC = 3, FL_BC_SYMMETRY = 4 } fl_bc;

/* GLOBAL grid.  xf[d]: n[d]+1 face coordinates (arrc[i][iprev]); xc[d]: n[d] cell centres (arrc[i][ielem]) or NULL for
 * midpoints (cart.c:136).  Host pointers, copied at create. */
typedef struct fl_grid {
  int64_t       n[3];
  const double *xf[3];
  const double *xc[3];
} fl_grid;

/* One rank's block of the m x n x p rank grid (DMStagCreate3d ownership, cart.c:88-104).  NULL = whole grid on one GPU. */
typedef struct fl_decomp {
  int     ranks[3];
  int     coord[3];
  int64_t lo[3], len[3];
} fl_decomp;

typedef enum { FL_KSP_CG = 0, FL_KSP_BCGS = 1, FL_KSP_CHEBYSHEV = 2, FL_KSP_GMRES = 3 } fl_ksp_type; /* -ksp_type cg|bcgs|chebyshev|gmres (gmres: fl_momentum_solve only) */
typedef enum { FL_PC_NONE = 0, FL_PC_JACOBI = 1, FL_PC_MG = 2 } fl_pc_type;                     /* -pc_type none|jacobi|mg (mg: fl_ksp_cg only) */
typedef enum { FL_NORM_PRECONDITIONED = 0, FL_NORM_UNPRECONDITIONED = 1, FL_NORM_NATURAL = 2, FL_NORM_NONE = 3 } fl_norm_type; /* -ksp_norm_type */

/* KSPConvergedReason values */
#define FL_CONVERGED_RTOL 2
#define FL_CONVERGED_ATOL 3
#define FL_CONVERGED_ITS 4
#define FL_DIVERGED_ITS (-3)
#define FL_DIVERGED_DTOL (-4)
#define FL_DIVERGED_BREAKDOWN (-5)
#define FL_DIVERGED_INDEFINITE_PC (-8)
#define FL_DIVERGED_NANORINF (-9)
#define FL_DIVERGED_INDEFINITE_MAT (-10)

/* Every field must hold a defined value: start from fl_ksp_opts_default (or a zeroed struct) and set what differs.  The struct grows at
 * its END between ABI versions (FL_ABI_VERSION / fl_abi_version below): a caller compiled against an older header hands over a shorter
 * struct, so check fl_abi_version() == FL_ABI_VERSION once at start-up. */
typedef struct fl_ksp_opts {
  int     type;             /* fl_ksp_type */
  int     pc;               /* fl_pc_type */
  int     norm_type;        /* fl_norm_type */
  int     remove_nullspace; /* constant null space attached to S (abfpc.c:173-177): y -= mean(y) after every PC apply */
  int     maxit;            /* -ksp_max_it   (PETSc default 10000) */
  double  rtol, atol, dtol; /* -ksp_rtol 1e-5, -ksp_atol 1e-50, -ksp_divtol 1e5 */
  double  emin, emax;       /* Chebyshev bounds of the preconditioned operator; 0,0 = Gershgorin bound * (0.1, 1.1) */
  int     variant;          /* 0 (the only value the product accepts): CG = fused kernels, q = S p formed twice and never stored, x updated
                               every second iteration (60 B/cell/iteration); BiCGStab = M S p and M S s formed where they are needed and never stored
                               (120 B/cell/iteration).  Other values select superseded implementations kept for A/B measurements (CG 1 = one kernel
                               per BLAS-1 / SpMV step, 2 = q stored and read back, 72 B/cell; BiCGStab: products stored, 152 B/cell): they exist in
                               a -DFL_KBENCH_VARIANTS build of the library only, here the solve returns FL_ERR_SUP. */
  int     check_every;      /* host polls the device-side convergence flag every this many iterations (0 = default 16);
                             * < 0 with FL_NORM_NONE (Chebyshev): never -- exactly maxit steps, no statistics (smoother use) */
  int     profile;          /* n > 0: bracket the kernels of every n-th pair of CG iterations (every Chebyshev launch) with HIP events ->
                               stats.kernel_ms (k_cg_A / the Chebyshev kernel), stats.kernel2_ms (k_cg_Bq); 1 = every iteration */
  double *history;          /* optional host array, receives the monitored norm of iterations 0..iters */
  int     nhistory;
  int     mg_levels;        /* FL_PC_MG: number of grid levels, 0 = coarsen as far as possible (-pc_mg_levels) */
  int     mg_smooth_its;    /* FL_PC_MG: Chebyshev-Jacobi steps before and after the coarse correction, 0 = 3 (-mg_levels_ksp_max_it) */
  int     gmres_restart;    /* FL_KSP_GMRES: -ksp_gmres_restart, 0 = 30 (PETSc's default) */
  int     cg_single_reduction; /* FL_KSP_CG with FL_PC_JACOBI / FL_PC_NONE: -ksp_cg_single_reduction (on the reference's sub-KSP:
                               -ns_abf_schur_ksp_cg_single_reduction, prefix built at abfpc.c:206): all inner products of an iteration in
                               ONE reduction -- one all-reduce and one scalar kernel per iteration on several ranks instead of two, for
                               72 instead of 60 B/cell/iteration (W = A p kept by recurrence).  0 (default) = off, as in PETSc.
                               With FL_PC_MG: FL_ERR_SUP (not built).  Other Krylov types ignore it, as PETSc ignores an option of another type. */
  int     initial_guess_nonzero; /* -ksp_initial_guess_nonzero (KSPSetInitialGuessNonzero), fl_momentum_solve / fl_abf_apply's kspA only (ABI 6): the
                               array handed over as x holds the initial guess.  Convergence as KSPConvergedDefault does it then: against the norm of
                               the RIGHT-HAND SIDE (in the KSP's norm), not of the initial residual.  fl_poisson_solve: FL_ERR_SUP (its kernels
                               start from zero).  The reference never sets it on kspA (PCApply_ABF's x holds no guess there); the host mirror
                               offers it for the first PCApply_ABF of a time step, where the previous velocity is one (include/fluca_host.h). */
} fl_ksp_opts;

typedef struct fl_ksp_stats {
  int    iters;
  int    reason; /* KSPConvergedReason */
  double rnorm0, rnorm;
  double seconds;         /* wall time of the solve measured with HIP events on the handle's stream */
  double kernel_ms;       /* profile=1: mean duration of the dominant kernel (HIP events), else 0 */
  int    kernel_launches; /* number of launches averaged in kernel_ms */
  double kernel2_ms;      /* profile=1, CG: mean duration of the second kernel of an iteration (k_cg_Bq: r-update, and the x-updates on
                             every second iteration), else 0 */
  int    kernel2_launches;
} fl_ksp_stats;

typedef struct fl_poisson  fl_poisson;
typedef struct fl_ibm      fl_ibm;
typedef struct fl_momentum fl_momentum;

/* ---- life cycle ------------------------------------------------------------------------------ */

/* bc[6] as fl_bc.  kappa = dt/rho (MatScale at cnlinearcart3d.c:2890,2907).  device = HIP device ordinal. */
int fl_poisson_create(const fl_grid *grid, const int bc[6], double kappa, const fl_decomp *decomp, int device, fl_poisson **out);
int fl_poisson_destroy(fl_poisson *h);
/* Run on a caller-owned hipStream_t (pass as void*); NULL = the handle's own stream. */
int fl_poisson_set_stream(fl_poisson *h, void *hip_stream);
int fl_poisson_synchronize(fl_poisson *h);
/* every rank of the handle's communicator has reached this call when it returns (MPI_Barrier on the object's comm) */
int fl_poisson_barrier(fl_poisson *h);
/* in-place sum of n <= 8 HOST doubles over the ranks of the handle's communicator (MPIU_Allreduce(..., MPI_SUM, comm) of a few scalars; a
 * flag "does any rank ..." is a sum of 0 / 1).  Blocking; a no-op on one rank.  Every rank must call it. */
int fl_poisson_allreduce_sum(fl_poisson *h, double *host_vals, int n);
/* sizes of this rank's arrays: out[0]=cells, out[1..3]=x,y,z faces */
int fl_poisson_sizes(const fl_poisson *h, int64_t out[4]);
void fl_ksp_opts_default(fl_ksp_opts *o); /* PETSc defaults + cg/jacobi/preconditioned norm */
const char *fl_version(void);
#define FL_ABI_VERSION 6
int fl_abi_version(void); /* the FL_ABI_VERSION the library was built with */

/* ---- device memory for hosts that have no allocator of their own (the C host mirror, a PETSc host without HIP Vecs) - */
int fl_current_device(int *device);                      /* the calling thread's current HIP device (hipGetDevice) */
int fl_malloc(int device, size_t bytes, void **dev_out); /* zero-initialised */
int fl_free(int device, void *dev);
int fl_memcpy_h2d(int device, void *dev, const void *host, size_t bytes);
int fl_memcpy_d2h(int device, void *host, const void *dev, size_t bytes);
/* Page-locked host memory, and a copy out of it that is ordered on the handle's stream like a kernel launch (no device-wide wait; dev may be read by
 * kernels enqueued before and after).  The host buffer must stay untouched until fl_poisson_upload_fence (waits for the LAST upload of the handle, hence
 * for all of them) or fl_poisson_synchronize has returned.  host need not come from fl_malloc_host, but only page-locked memory copies asynchronously. */
int fl_malloc_host(size_t bytes, void **host_out);
int fl_free_host(void *host);
int fl_poisson_upload(fl_poisson *h, void *dev, const void *host, size_t bytes);
int fl_poisson_upload_fence(fl_poisson *h);

/* y = a x + b z (z_dev may be NULL; y may alias x or z) and result = sum x y over all ranks (blocking; every rank passes its
 * owned entries) -- for hosts that run an outer iteration over device vectors and have no vector library of their own. */
int fl_vec_lincomb(fl_poisson *h, int64_t n, double a, const double *x_dev, double b, const double *z_dev, double *y_dev);
int fl_vec_dot(fl_poisson *h, int64_t n, const double *x_dev, const double *y_dev, double *result);
/* VecMDot / VecMAXPY: out[i] = x . y_i (one host wait for all k) and x += sum_i alpha_i y_i; ys_dev is a HOST array of k device
 * pointers.  Classical Gram-Schmidt of the outer GMRES reads its work vector once per pass instead of once per basis vector. */
int fl_vec_mdot(fl_poisson *h, int64_t n, const double *x_dev, const double *const *ys_dev, int k, double *out);
int fl_vec_maxpy(fl_poisson *h, int64_t n, double *x_dev, const double *alphas, const double *const *ys_dev, int k);

/* Build-specific tuning knobs (no reference counterpart; results never change beyond round-off).  ONE table (fluca_amd/csrc/fl_knobs.h); a knob's
 * initial value is its default or, if set, the environment variable FLUCA_<NAME IN CAPITALS>, which the library reads once per process.  Names:
 *   "cheb_fuse"  0 = one kernel launch per Chebyshev step; 1 (default) = two steps per sweep over memory wherever no convergence
 *                test sits between them (KSP_NORM_NONE sweeps, the multigrid smoother) and the grid is large enough to gain;
 *                2 = the same on every grid where it is legal.  On several ranks the ranks vote once per handle: fused only where all agree.
 *   "schur_var_fused" 1 (default) = fl_abf_schur_apply / the Schur solve with schurainv DIAG or ROWSUM form S p in one pass on one rank
 *                (fl_schur_var.hip); 0 = the composition of projection, scaling, face interpolation and divergence (A/B, tests; several ranks always)
 *   "cheb_zero3" 1 (default) = a smoother call that starts from a zero guess (the pre-smoother of a multigrid cycle) runs its first THREE steps in
 *                one sweep, the outer CG's residual update included, where "cheb_fuse" applies and the handle has one rank; 0 = first step and
 *                pair as separate launches (A/B, tests).  Same arithmetic either way.
 *   "placement"  0 (default since round 3) = one plain allocation per vector; 1 = the first solve on a handle whose padded vectors are
 *                >= 256 MiB runs the placement search of fl_poisson_tune_placement by itself (about 0.15 s, once per handle; worth
 *                1 - 2 % of the CG iteration rate at 512^3).  A failure inside the search never fails the solve: plain allocations.
 *                "placement_vmm" (1): its arenas live in chunk-mapped virtual memory; "placement_verbose" (0): it narrates on stderr.
 *   "cg_xbatch"  1 (default) = the CG solver updates x every second iteration (both updates of the pair at once, while the older
 *                direction is still in its buffer); 0 = one update per iteration.  The same x bit for bit.
 *   "mg_prolong" 1 (default) = tri-linear prolongation of the FL_PC_MG cycle; 0 = piecewise constant.  (This one and the next change the
 *                preconditioner, i.e. iteration counts -- not the converged answer.)
 *   "mg_flexible" 1 (default) = the CG around the FL_PC_MG cycle forms beta in the Polak-Ribiere way (flexible CG, KSPFCG with
 *                -ksp_fcg_mmax 1): robust against the cycle not being a symmetric operator; 0 = KSPCG's beta.
 *   "mg_coarse"  1 (default) = a coarsest multigrid level of at most 4096 cells on one rank is solved by ONE workgroup (no launches per CG iteration,
 *                no host poll inside the cycle); 0 = through the public Jacobi-PCG like every other size.  Same algorithm, other summation order.
 *   "mg_post_smooth" 0 (default) = as many smoothing steps after the coarse correction as before it (fl_ksp_opts.mg_smooth_its); n > 0 = n steps
 *                (V(3,2) is 3 % faster than V(3,3) at 512^3 and one iteration longer: profiles/r05_mg_nu.txt).
 *   "overlap"    1 (default) = on several ranks the exchange of the new residual runs behind the update kernel on a second stream; 0 = after it.
 *   "comm_loopback" 1 = a handle created on ONE rank sends the ghost layers of its periodic axes to itself through the communicator instead of
 *                copying them (exercises RCCL on a one-GPU box; looked at by fl_poisson_create).  "comm_trace" 1 / 2 = exchanges narrated on stderr.
 *   "ghost_width" 0 (default) = automatic: two ghost layers where a neighbouring rank exists, else one; 1 / 2 = forced (looked at by fl_poisson_create).
 *   "allreduce"  0 (default) = scalar reductions through ncclAllReduce / the host callback; 1 = the one-shot all-reduce through peer-mapped buffers
 *                (fl_poisson_comm_init_oneshot below).
 * Returns FL_ERR_ARG_WRONG for an unknown name.  Process-wide atomics: handles on several host threads may read them while they run; set a knob
 * before the solves it should affect.  (Switches between a shipped code path and a superseded one -- round 1's kernels, stored-q CG, ... -- are
 * compile-time constants in this library; a -DFL_KBENCH_VARIANTS build turns them into knobs of the same table for A/B measurements.) */
int fl_tuning_set(const char *name, int value);
int fl_tuning_get(const char *name, int *value);

/* Placement of the solver vectors in HBM (opt-in).  A kernel that streams five gigabyte-sized vectors in lock step runs up to 13 %
 * slower when all of them sit in one physically contiguous block of memory (what back-to-back allocations give) than when two or
 * three of them come from a different block (measurements: profiles/r02_placement.md).  This call -- or the first solve of a large
 * handle when the tuning knob "placement" is 1 -- builds a scratch arena (16 GiB + 8 vectors) out of 256 MiB chunks of physical
 * memory mapped into one reserved address range (HIP virtual memory management), slides a window of five packed vectors through
 * it with the CG kernel pair as the probe (about 20 positions), and then RELEASES every chunk the fastest window does not touch:
 * the vectors stay on the physical memory that was measured.  MEMORY: the handle afterwards holds five padded vectors plus at most
 * two chunks of slack (5.8 - 6.3 GB at 512^3; an unplaced handle holds 5.75 GB); the arena (up to three while searching, 25 GiB
 * each at 512^3) exists only during the call.  Where the address-range calls are not available the arena is one plain allocation and
 * is kept whole, as in round 2.  Deterministic, idempotent, never changes results.  max_tries >= 1 (kept from the earlier
 * interface, unused).
 * probe_ms_out (may be NULL): {probe time with all vectors in one block, probe time of the vectors the handle ended up with};
 * {0, 0} if the handle is too small to be placed or memory is short. */
int fl_poisson_tune_placement(fl_poisson *h, int max_tries, double probe_ms_out[2]);
/* bytes of device memory the handle holds for its padded solver vectors */
int fl_poisson_vector_bytes(fl_poisson *h, int64_t *bytes_out);

/* ---- operator -------------------------------------------------------------------------------- */
int fl_poisson_apply(fl_poisson *h, const double *x_dev, double *y_dev);  /* y = S x */
int fl_poisson_diagonal(fl_poisson *h, double *d_dev);                    /* MatGetDiagonal(S) */
int fl_poisson_solve(fl_poisson *h, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats);
/* Upper bound of the spectrum of M S (M = 1/diag for pc = FL_PC_JACOBI, identity otherwise) from the 1-D operator tables:
 * what the Chebyshev solver / smoother multiplies by (0.1, 1.1) when emin = emax = 0.  Host-only, no GPU work. */
int fl_poisson_gershgorin(const fl_poisson *h, int pc, double *bound);
/* b = contrhs - D V   (contrhs_dev may be NULL = 0) */
int fl_poisson_rhs(fl_poisson *h, const double *Vx_dev, const double *Vy_dev, const double *Vz_dev, const double *contrhs_dev, double *b_dev);
/* v_d -= kappa (G p)_d at cell centres (any v*_dev may be NULL), V_d -= kappa (Gst p)_d on faces (any V*_dev may be NULL) */
int fl_poisson_project(fl_poisson *h, const double *p_dev, double *vx_dev, double *vy_dev, double *vz_dev, double *Vx_dev, double *Vy_dev, double *Vz_dev);
/* boundary = 0..5; pb_dev: boundary pressures on that boundary's faces of this rank (plane, x fastest); writes
 * coeff*pb into the boundary faces of V_dev (face array of the boundary's axis), INSERT_VALUES semantics.  No-op (success)
 * unless bc[boundary] is PRESSURE_OUTLET and this rank touches the boundary. */
int fl_poisson_gst_bc(fl_poisson *h, int boundary, const double *pb_dev, double *V_dev);
/* Building blocks of the reference's boundary-condition vectors (Compute*BoundaryConditionVector_Private,
 * cnlinearcart3d.c:219-423, 648-871, 1296-1511, 1749-1932, 2142-2312, 2602-2805): each is "a value per boundary face"
 * (plane_dev: this rank's part of boundary 0..5, in-plane axes in x,y,z order, the first fastest) either written into the
 * boundary faces of a face array of the boundary's axis (DMStagVecSetValuesStencil INSERT_VALUES) or added to the cells
 * next to the boundary (ADD_VALUES).  No-op on ranks that do not touch the boundary and on periodic axes. */
int fl_boundary_set_faces(fl_poisson *h, int boundary, double coeff, const double *plane_dev, double *face_dev);
int fl_boundary_add_cells(fl_poisson *h, int boundary, double coeff, const double *plane_dev, double *cell_dev);
int fl_boundary_add_faces(fl_poisson *h, int boundary, double coeff, const double *plane_dev, double *face_dev); /* ADD_VALUES into the boundary faces */
/* first != 0: p = p0 + 2 dp, phalf = p0 + dp ; else p = phalf + 1.5 dp, phalf += dp */
int fl_pressure_update(fl_poisson *h, int first, const double *dp_dev, const double *p0_dev, double *phalf_dev, double *p_dev);

/* ---- DMStag vectors <-> the arrays above (what a PETSc-side caller needs around every other call) ---------------
 * The reference keeps p on sdm (1 dof per element), v on vdm (3 dof per element), V on Sdm (1 dof per face) and v0interp on Vdm
 * (3 dof per face) -- fluca/src/mesh/impl/cart/cart.c:88-116, cnlinearcart3d.c:896-905.  `what`: 0 = cells (DMSTAG_ELEMENT),
 * 1 / 2 / 3 = the x / y / z faces (DMSTAG_LEFT / DMSTAG_DOWN / DMSTAG_BACK).  All on the device, on the handle's stream.
 * LOCAL: the array DMStagVecGetArray[Read] returns for a local vector (or its device copy): arr[k][j][i][slot] over the ghosted
 * box; fill the struct from DMStagGetGhostCorners, DMStagGetCorners and DMStagGetEntriesPerElement, `slot` from
 * DMStagGetLocationSlot(dm, loc, c, &slot).  "to" writes the owned entries only (INSERT_VALUES), ghosts are left alone.
 * GLOBAL: the array of a global vector of this rank (what PCApply_ABF's sub-vectors are): dof[4] = DMStagGetDOF (strata 0 and
 * 1 must be 0), `comp` = the dof index c within the location.  The partial elements PETSc appends behind the last element of a
 * non-periodic axis are handled; fl_dmstag_global_entries returns the local size such a vector must have.
 * Ordering per PETSc's DMStag (DMSetUp_Stag_3d); checked against an independent enumeration, not against PETSc (absent here). */
typedef struct fl_dmstag_local {
  int64_t gstart[3]; /* DMStagGetGhostCorners: first element of the ghosted box (may be -1) */
  int64_t gsize[3];  /*                        its extents */
  int64_t start[3];  /* DMStagGetCorners: first owned element (must equal fl_decomp.lo) */
  int     entries;   /* DMStagGetEntriesPerElement */
} fl_dmstag_local;
int fl_layout_from_dmstag_local(fl_poisson *h, const fl_dmstag_local *D, int what, int slot, const double *local_dev, double *out_dev);
int fl_layout_to_dmstag_local(fl_poisson *h, const fl_dmstag_local *D, int what, int slot, const double *in_dev, double *local_dev);
int fl_layout_from_dmstag_global(fl_poisson *h, const int dof[4], int what, int comp, const double *global_dev, double *out_dev);
int fl_layout_to_dmstag_global(fl_poisson *h, const int dof[4], int what, int comp, const double *in_dev, double *global_dev);
int fl_dmstag_global_entries(const fl_poisson *h, const int dof[4], int64_t *entries);

/* ---- multi-GPU: one process per GPU, halo exchange + scalar all-reduce ----------------------- */
#define FL_UNIQUE_ID_BYTES 128
/* rank 0 calls this, the host broadcasts the bytes (torch.distributed / MPI), every rank calls ..._comm_init_rccl */
int fl_comm_unique_id(void *out128);
int fl_poisson_comm_init_rccl(fl_poisson *h, const void *id128, int rank, int nranks);
/* Host-staged transport for tests / hosts without RCCL (e.g. gloo): the library stages faces through pinned host memory.
 * exchange: nmsg messages; peer[m] = rank to swap with; send[m]/recv[m] host buffers of nbytes[m]; tag[m] disambiguates
 * two messages between the same pair.  allreduce: in-place sum of n doubles. */
typedef int (*fl_exchange_fn)(void *ctx, int nmsg, const int *peer, const int *sendtag, const int *recvtag, void *const *send, void *const *recv, const int64_t *nbytes);
typedef int (*fl_allreduce_fn)(void *ctx, double *vals, int n);
int fl_poisson_comm_init_host(fl_poisson *h, fl_exchange_fn xchg, fl_allreduce_fn allred, void *ctx, int rank, int nranks);
/* One-shot all-reduce (opt-in; tuning knob "allreduce" = 1): the scalar reductions of a solve -- 8 doubles, twice per CG iteration -- written
 * straight into the peers' mailboxes (fine-grained device memory every rank has mapped) by one single-wave kernel per rank, instead of
 * ncclAllReduce's rendezvous; the sums are added in rank order, the same bits on every rank.  Set-up, after fl_poisson_comm_init_*: every rank
 * calls ..._oneshot_handle (its mailbox as a 64-byte hipIpcMemHandle_t, and as an address for handles of the same process), the host gathers
 * them, every rank calls ..._oneshot_attach with the whole list (either nranks x 64 bytes of IPC handles, or nranks addresses).  At most 8 ranks.
 * A wait that sees no peer for about two seconds gives up: the sums are NaN (the solve ends with KSP_DIVERGED_NANORINF) and ..._oneshot_error
 * reports 1. */
#define FL_IPC_HANDLE_BYTES 64
int fl_poisson_comm_oneshot_handle(fl_poisson *h, void *ipc_handle64, void **address);
int fl_poisson_comm_oneshot_attach(fl_poisson *h, const void *handles, void *const *addresses);
int fl_poisson_comm_oneshot_error(fl_poisson *h, int *error);
/* What the handle's communicator IS, for a caller (bench.py) that must prove which wire carried its halos: transport 0 none, 1 RCCL, 2 host
 * callbacks; rank / nranks as RCCL itself reports them (ncclCommUserRank / ncclCommCount) for transport 1, as given at init otherwise;
 * neighbours = ranks this one exchanges ghost layers with; halo_bytes = bytes this rank SENDS per ghost exchange of one cell vector. */
typedef struct fl_comm_info {
  int     transport, rank, nranks, loopback, neighbours, messages;
  int64_t halo_bytes;
} fl_comm_info;
int fl_poisson_comm_info(fl_poisson *h, fl_comm_info *out);

/* The ghost-exchange plan of one rank (host-only, no GPU): what fl_poisson_* does before every stencil application, the
 * analogue of DMGlobalToLocal on the reference's star-stencil DMStag (cart.c:66,91).  For each message: swap with `peer`;
 * send the owned cell layer at boundary `send_boundary` (0..5, -1 = nothing), receive into the ghost layer at
 * `recv_boundary` (-1 = nothing).  Messages between one pair of ranks must be matched in array order (RCCL) or by tag. */
typedef struct fl_halo_msg {
  int peer, send_boundary, recv_boundary, sendtag, recvtag;
} fl_halo_msg;
int fl_halo_plan(const fl_decomp *d, const int periodic[3], fl_halo_msg out[12]); /* returns the number of messages (<= 12) */

/* Host-only helper = DMStag's default ownership split (N/m cells each, the first N%m ranks get one more). */
int fl_decomp_default(const int64_t n[3], const int ranks[3], int rank, fl_decomp *out);
/* rank of the neighbour across boundary 0..5 of this block, -1 if physical (non-periodic) boundary */
int fl_decomp_neighbor(const fl_decomp *d, const int periodic[3], int boundary);

/* ---- momentum block of the Jacobian (first "next" row) --------------- */
/* Matrix-free A = I + dt C - (mu dt / 2 rho) L on the cell-centred velocity, replacing the AIJ matrix the reference
 * re-assembles every step (NSFormJacobian_CNLinear_Cart3d_Internal, cnlinearcart3d.c:2930-2941):
 *   L  ComputeVelocityLaplacianOperator_Private   cnlinearcart3d.c:425-632   (1-D rows: cartdiscret.c:167-303)
 *   C  ComputeConvectionOperator_Private          cnlinearcart3d.c:873-1294  (1-D rows: cartdiscret.c:305-371)
 * Velocity vectors are component-major: v[c*cells + cell], c = 0,1,2 (the reference's DMStag vector interleaves the
 * components per element -- a fixed permutation).  Grid, boundary conditions, decomposition, stream and communicator
 * are those of `grid_from`, which must outlive the momentum handle; every local block needs >= 2 cells per axis and
 * a VELOCITY / SYMMETRY wall needs >= 3 cells along its axis (the reference's one-sided rows read cell i+-2). */
int fl_momentum_create(fl_poisson *grid_from, fl_momentum **out);
int fl_momentum_destroy(fl_momentum *m);
/* V0_dev[d]: face-normal velocity of the previous step on the d-faces (sol0's NS_FIELD_FACE_NORMAL_VELOCITY);
 * v0interp_dev[c*3+d]: component c of cnl->v0interp on the d-faces.  Face layouts as at the top of this file.
 * Copies the twelve fields (the caller may reuse its buffers) and sets cI = 1, cC = dt, cL = -mu dt / (2 rho). */
int fl_momentum_set_state(fl_momentum *m, double dt, double rho, double mu, const double *const V0_dev[3], const double *const v0interp_dev[9]);
/* The same state, handed over together with the cell-centred velocity it was interpolated from: v0_dev = sol0's NS_FIELD_VELOCITY
 * (3 * cells, component-major), v0interp = B v0 + vbc (cnlinearcart3d.c:2826-2829; vbc is non-zero on boundary faces only, :1749-1932).
 * The operator then forms v0interp on the inner faces from v0 inside its kernel (bit-identical to fl_momentum_interp_faces) and reads the
 * nine stored fields only on the faces at the ends of this rank's block: 96 B per cell and product instead of 144.  A v0interp that is
 * NOT B v0 on inner faces gives a different operator than fl_momentum_set_state -- use that entry point then. */
int fl_momentum_set_state_v0(fl_momentum *m, double dt, double rho, double mu, const double *const V0_dev[3], const double *const v0interp_dev[9], const double *v0_dev);
/* A = cI I + cC C + cL L with explicit coefficients (other time integrators; L or C alone in the parity tests) */
int fl_momentum_set_coefficients(fl_momentum *m, double cI, double cC, double cL);
int fl_momentum_apply(fl_momentum *m, const double *v_dev, double *y_dev); /* y = A v      (MatMult) */
int fl_momentum_diagonal(fl_momentum *m, double *d_dev);                   /* MatGetDiagonal(A) */
/* KSPSolve(abf->kspA, momrhs, vstar), abfpc.c:72, zero initial guess, pc JACOBI or NONE; remove_nullspace is ignored (A is non-singular).
 * opts->type: FL_KSP_BCGS (left-preconditioned KSPBCGS, preconditioned norm), FL_KSP_GMRES (PETSc's default type for kspA; restart
 * opts->gmres_restart) or FL_KSP_CHEBYSHEV (KSPCHEBYSHEV's three-term recurrence fused into the product: 144 B per cell and step where a
 * BiCGStab iteration moves 552 -- the method of choice while the operator is diffusion-dominated, e.g. nu dt / h^2 > 1; interval from
 * opts->emin / emax = -ksp_chebyshev_eigenvalues, or with PCJACOBI fl_momentum_chebyshev_interval). */
int fl_momentum_solve(fl_momentum *m, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats);
/* V_d = rhs_d + (T v)_d on the d-faces: MatMult(abf->negT, vstar, Vstar); VecAYPX(Vstar, -1, interprhs), abfpc.c:73-74.
 * T = ComputeFaceNormalVelocityInterpolationOperator_Private, cnlinearcart3d.c:1934-2140.  rhs_dev (or any entry) may
 * be NULL = 0.  A VELOCITY / SYMMETRY wall face has no T row: it receives rhs alone (the boundary-condition vector). */
int fl_momentum_face_interp(fl_momentum *m, const double *v_dev, const double *const rhs_dev[3], double *const V_dev[3]);
/* V_d = rhs_d + alpha (T v)_d; rhs_dev[d] may alias V_dev[d] (MatMultAdd(negT, ...) of the Rhie-Chow boundary terms, cnlinearcart3d.c:3033: alpha = -1) */
int fl_momentum_face_interp_scaled(fl_momentum *m, double alpha, const double *v_dev, const double *const rhs_dev[3], double *const V_dev[3]);
/* out[c*3+d] = vbc[c*3+d] + (B v)_c on the d-faces: cnl->v0interp of NSStep_CNLinear_Cart3d_Internal
 * (cnlinearcart3d.c:2826-2829), B = ComputeFaceVelocityInterpolationOperator_Private (cnlinearcart3d.c:1513-1747).
 * vbc_dev (or any entry) may be NULL = 0; out may be handed straight to fl_momentum_set_state. */
int fl_momentum_interp_faces(fl_momentum *m, const double *v_dev, const double *const vbc_dev[9], double *const out_dev[9]);
/* The same rows on the faces at the two ends of each axis of this rank's block only (the inner entries of out are left alone): everything
 * fl_momentum_set_state_v0 reads of v0interp while the operator forms the inner faces from v0 itself (k_mom3) -- a ninth of the work on a 512^3 block.
 * Where the operator will read whole fields anyway (a block with ny <= 8, FLUCA_MOM_KERNEL=2) this IS fl_momentum_interp_faces.
 * A state set with fl_momentum_set_state needs the whole fields: use fl_momentum_interp_faces there. */
int fl_momentum_interp_faces_ends(fl_momentum *m, const double *v_dev, const double *const vbc_dev[9], double *const out_dev[9]);
/* The cell-wise part of momrhs, NSFormFunction_CNLinear_Cart3d_Internal (cnlinearcart3d.c:2976-2998):
 *   momrhs = v0 + (mu dt / 2 rho) L v0 - kappa G p + vbc
 * p_dev: phalf (or p0 on the first step), may be NULL; vbc_dev (3*cells, may be NULL): the boundary-condition vectors of
 * L, C and G combined by the caller, (mu dt/2 rho)(vbcL(t) + vbcL(t+dt)) - dt vbcC - vbcG -- they are zero for periodic
 * boundaries and homogeneous walls.  kappa is the handle's dt/rho. */
int fl_momentum_rhs(fl_momentum *m, double dt, double rho, double mu, const double *v0_dev, const double *p_dev, const double *vbc_dev, double *momrhs_dev);

/* ---- the whole preconditioner application ---------------------------------------------------- */
/* MatMult of the 3 x 3 block Jacobian the preconditioner belongs to (MatNest of cnlinearcart3d.c:2885-2941), for an outer
 * Krylov method that keeps its vectors on the device:
 *   fv = A v + kappa G p ;  fV = V - T v - R p,  -R = (-T)(kappa G) + kappa Gst ;  fp = D V */
int fl_abf_jacobian_mult(fl_momentum *m, const double *v_dev, const double *const V_dev[3], const double *p_dev, double *fv_dev, double *const fV_dev[3], double *fp_dev);
/* PCApply_ABF, abfpc.c:48-111, with the reference's default upperainv = schurainv = ID (abfpc.c:328-329):
 *   v* = A^-1 momrhs ; V* = interprhs + T v* ; p = S^-1 (contrhs - D V*) ; v = v* - kappa G p ; V = V* - kappa Gst p
 * v_dev (3*cells, component-major), V_dev[3] (faces) and p_dev (cells) are outputs; interprhs_dev / contrhs_dev may be
 * NULL = 0.  stats[0] = KSPSolve(kspA), stats[1] = KSPSolve(kspS).  A non-converged inner solve is reported in stats,
 * not as an error (PETSc's behaviour without -ksp_error_if_not_converged). */
int fl_abf_apply(fl_momentum *m, const fl_ksp_opts *momentum_opts, const fl_ksp_opts *schur_opts, const double *momrhs_dev, const double *const interprhs_dev[3], const double *contrhs_dev, double *v_dev,
                 double *const V_dev[3], double *p_dev, fl_ksp_stats stats[2]);
/* PCABFAinvType (flucans.h:99-103): the approximation of A^-1 inside the Schur complement (PCABFSetSchurComplementAinvType,
 * -pc_abf_schur_ainv_type) and inside the upper-triangular solve (PCABFSetUpperTriangularAinvType, -pc_abf_upper_ainv_type).
 * ID is the default (abfpc.c:328-329) and the fused matrix-free path; DIAG / ROWSUM make S a variable-coefficient operator
 * that follows A (abfpc.c:155-165): applied as a composition of kernels and solved by flexible GMRES preconditioned with
 * the ID Schur solve. */
enum { FL_ABF_AINV_ID = 0, FL_ABF_AINV_DIAG = 1, FL_ABF_AINV_ROWSUM = 2 };
int fl_abf_set_ainv_types(fl_momentum *m, int schur_type, int upper_type);
/* y = S p with the current schurainv type: MatMult(abf->S) */
int fl_abf_schur_apply(fl_momentum *m, const double *p_dev, double *y_dev);
/* MatGetRowSum(A) into 3*cells doubles (component-major), like fl_momentum_diagonal */
int fl_momentum_rowsum(fl_momentum *m, double *out_dev);
/* Gershgorin radius of the Jacobi-scaled momentum operator: max over the rows of (sum of |a_ij|, j != i) / |a_ii|, over all ranks.  Every
 * eigenvalue of D^-1 A lies in the disc of this radius around 1.  One product-sized launch and a host wait per state (cached). */
int fl_momentum_gershgorin(fl_momentum *m, double *radius);
/* The interval FL_KSP_CHEBYSHEV on kspA uses when opts->emin = emax = 0 (PCJACOBI): emax = 1 + g, emin = max(1 - g, 0.9 / mean_i a_ii) -- the
 * second is where the spectrum of a viscous-dominated A = I + dt C - (mu dt / 2 rho) L ends (fl_momentum.hip says why). */
int fl_momentum_chebyshev_interval(fl_momentum *m, double *emin, double *emax);

/* ---- immersed boundary (build-defined; no reference function) -------------------------------- */
typedef enum { FL_DELTA_PESKIN4 = 0, FL_DELTA_ROMA3 = 1 } fl_delta_kind;
/* markers X,Y,Z (device, length L) are binned per cell at create / update.  Any tensor-product grid: on a stretched axis the delta
 * function is evaluated in index space (marker position -> continuous cell-centre index, piecewise linear through the centres) and
 * spreading divides by the volume of the target cell; on uniform axes that is the textbook delta_h. */
int fl_ibm_create(fl_poisson *grid_from, int kind, int64_t L, const double *X_dev, const double *Y_dev, const double *Z_dev, fl_ibm **out);
int fl_ibm_update(fl_ibm *m, const double *X_dev, const double *Y_dev, const double *Z_dev);
/* U[c*L + l] = sum_x u[c*ncell + x] delta_h(x - X_l) h^3 */
int fl_ibm_interp(fl_ibm *m, int ncomp, const double *u_dev, double *U_dev);
/* f[c*ncell + x] += sum_l F[c*L + l] delta_h(x - X_l) dV_l */
int fl_ibm_spread(fl_ibm *m, int ncomp, const double *F_dev, const double *dV_dev, double *f_dev);
int fl_ibm_destroy(fl_ibm *m);

#ifdef __cplusplus
}
#endif
#endif /* FLUCA_HIP_H */
