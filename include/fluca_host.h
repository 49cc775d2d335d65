/*
 * fluca_host.h -- host-side C mirror of the part of Fluca's Mesh / NS plugin surface that stands in front of the
 * pressure-Poisson path.  Same names, argument meaning and error behaviour as the reference; PETSc types replaced by
 * plain C (no MPI_Comm: the rank/size of the one-process-per-GPU job are set explicitly; no Vec: raw device pointers).
 * Everything numerical goes through the C-ABI of include/fluca_hip.h.
 *
 * Reference interfaces mirrored (file:line relative to thecasterian/fluca):
 *   MeshCartCreate3d / MeshSetFromOptions / MeshSetUp / MeshCartSetUniformCoordinates / MeshCartGet*   fluca/include/flucameshcart.h:29-56,
 *                                                                                                       fluca/src/mesh/impl/cart/cart.c:13-151,458-591
 *   struct _MeshOps, MeshRegister                                                                       fluca/include/fluca/private/meshimpl.h:16-25, flucamesh.h:43-44
 *   NSCreate / NSSetType / NSRegister / NSSetMesh / NSSet{Density,Viscosity,TimeStepSize} /
 *   NSSetBoundaryCondition / NSSetFromOptions / NSSetUp / NSDestroy                                     fluca/include/flucans.h:28-60,91-92
 *   struct _NSOps: all nine slots in the reference's order (setfromoptions, setup, step, formjacobian,
 *   formfunction, destroy, view, viewsolution, loadsolution); NSFormJacobian / NSFormFunction / NSView /
 *   NSViewSolution / NSLoadSolution dispatch through them                                               fluca/include/fluca/private/nsimpl.h:21-31, cnlinear.c:177-185,
 *                                                                                                       nsbasic.c:301-323,353-374, nssol.c:130-203
 *   struct _MeshOps: all eight slots (setfromoptions, setup, destroy, view, load, createglobalvector,
 *   creatematrix, getnumberboundaries); MeshView / MeshLoad / MeshCreateGlobalVector / MeshCreateMatrix  fluca/include/fluca/private/meshimpl.h:16-25, meshbasic.c:93-127, cart.c:171-260
 *   PetscViewer as far as these slots need it: FlucaViewer (ASCII here, the CGNS one in fluca_cgns.h)   fluca/src/viewer/impl/cgns/flucacgns.c
 *   The structs behind Mesh / NS / FlucaViewer, for code that registers a type: include/fluca_host_impl.h
 *   NSBoundaryCondition, NSBoundaryConditionFunction                                                    fluca/include/flucansbc.h:5-22
 *   PCApply_ABF without the momentum solve (NSPressureCorrection)                                       fluca/src/ns/utils/abfpc/abfpc.c:73-101
 *   PCApply_ABF in full (NSApplyPreconditioner) and the A block of NSFormJacobian (NSSetPreviousState)  abfpc.c:48-111, cnlinearcart3d.c:2930-2941
 *   pressure update of NSStep_CNLinear_Cart3d_Internal (NSUpdatePressure)                               fluca/src/ns/impl/linearcn/cnlinearcart3d.c:2846-2854
 *   options -cart_grid_x.. -cart_ranks_x.. -cart_boundary_type_x.. -ns_density -ns_viscosity
 *           -ns_time_step_size -ns_max_steps -ns_abf_schur_ksp_{type,rtol,atol,max_it,norm_type} -ns_abf_schur_pc_type   cart.c:21-43, nsopts.c:177-198, abfpc.c:206,248-249
 *           (-ns_abf_schur_pc_type mg [-ns_abf_schur_pc_mg_levels N -ns_abf_schur_mg_levels_ksp_max_it NU]: the build's own multigrid)
 *
 * Every function returns FlErrorCode: 0 = success, otherwise the positive PETSC_ERR_* value the reference would raise.
 */
#ifndef FLUCA_HOST_H
#define FLUCA_HOST_H

#include "fluca_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int FlErrorCode;
#define FL_DECIDE (-1) /* PETSC_DECIDE */

typedef struct _p_Mesh *Mesh;
typedef struct _p_NS   *NS;

typedef const char *MeshType;
#define MESHCART "cart"
typedef const char *NSType;
#define NSCNLINEAR "cnlinear"

typedef enum { MESHCART_BOUNDARY_NONE, MESHCART_BOUNDARY_PERIODIC } MeshCartBoundaryType;
typedef enum { MESHCART_LEFT, MESHCART_RIGHT, MESHCART_DOWN, MESHCART_UP, MESHCART_BACK, MESHCART_FRONT } MeshCartBoundaryLocation;

/* ---- Viewer: what PetscViewer is to the view / load slots below ----
 * An abstract sink / source of named fields with its own ops table (include/fluca_host_impl.h).  Two types exist:
 * FLUCAVIEWERASCII (here; PETSCVIEWERASCII as far as NSView / MeshView print) and FLUCAVIEWERCGNS (fluca_cgns.h =
 * PETSCVIEWERFLUCACGNS, flucacgns.c).  A FlucaViewerCGNS handle IS a FlucaViewer. */
typedef struct _p_FlucaViewer *FlucaViewer;
typedef const char            *FlucaViewerType;
#define FLUCAVIEWERASCII "ascii"
#define FLUCAVIEWERCGNS  "flucacgns"
FlErrorCode FlucaViewerASCIIOpen(const char *filename, FlucaViewer *viewer); /* NULL or "stdout": standard output */
FlErrorCode FlucaViewerGetType(FlucaViewer viewer, FlucaViewerType *type);
FlErrorCode FlucaViewerDestroy(FlucaViewer *viewer);

/* ---- Mesh ---- */
/* MeshDMType (flucamesh.h:20-25): which of the mesh's four DMs a vector lives on */
typedef enum { MESH_DM_SCALAR, MESH_DM_VECTOR, MESH_DM_STAG_SCALAR, MESH_DM_STAG_VECTOR } MeshDMType;
/* meshimpl.h:16-25, same slots in the same order.  Vec -> a device array (`device` = the HIP device the NS runs on; the
 * reference's Vec carries its own communicator and memory space), Mat -> nothing: every operator of the path is
 * matrix-free, the Cart type leaves creatematrix NULL and MeshCreateMatrix answers PETSC_ERR_SUP. */
struct _MeshOps {
  FlErrorCode (*setfromoptions)(Mesh, int, char **);
  FlErrorCode (*setup)(Mesh);
  FlErrorCode (*destroy)(Mesh);
  FlErrorCode (*view)(Mesh, FlucaViewer);
  FlErrorCode (*load)(Mesh, FlucaViewer);
  FlErrorCode (*createglobalvector)(Mesh, MeshDMType, int device, double **vec_dev, int64_t *n);
  FlErrorCode (*creatematrix)(Mesh, MeshDMType, MeshDMType, void **mat);
  FlErrorCode (*getnumberboundaries)(Mesh, int *);
};
FlErrorCode MeshRegister(const char name[], FlErrorCode (*create)(Mesh));
FlErrorCode MeshCreate(Mesh *mesh);
FlErrorCode MeshSetType(Mesh mesh, MeshType type);
FlErrorCode MeshSetRank(Mesh mesh, int rank, int size); /* stands in for the MPI_Comm of MeshCreate(comm, ...) */
FlErrorCode MeshCartCreate3d(MeshCartBoundaryType bndx, MeshCartBoundaryType bndy, MeshCartBoundaryType bndz, int64_t M, int64_t N, int64_t P, int m, int n, int p, const int64_t *lx, const int64_t *ly, const int64_t *lz, Mesh *mesh);
FlErrorCode MeshSetFromOptions(Mesh mesh, int argc, char **argv);
FlErrorCode MeshSetUp(Mesh mesh);
FlErrorCode MeshCartSetUniformCoordinates(Mesh mesh, double xmin, double xmax, double ymin, double ymax, double zmin, double zmax);
FlErrorCode MeshCartSetCoordinates(Mesh mesh, const double *xf, const double *yf, const double *zf); /* = the coordLoaded path, cart.c:131-140 */
FlErrorCode MeshCartGetGlobalSizes(Mesh mesh, int64_t *M, int64_t *N, int64_t *P);
/* cart.c:432-456; MeshSetFromOptions honours -cart_refine_{x,y,z} <factor> and -cart_refine <n> (cart.c:37-52: sizes and ownership ranges times factor^n) */
FlErrorCode MeshCartSetRefinementFactor(Mesh mesh, int64_t refine_x, int64_t refine_y, int64_t refine_z);
FlErrorCode MeshCartGetRefinementFactor(Mesh mesh, int64_t *refine_x, int64_t *refine_y, int64_t *refine_z);
FlErrorCode MeshCartGetNumRanks(Mesh mesh, int *m, int *n, int *p);
FlErrorCode MeshCartGetCorners(Mesh mesh, int64_t *x, int64_t *y, int64_t *z, int64_t *m, int64_t *n, int64_t *p);
FlErrorCode MeshCartGetIsFirstRank(Mesh mesh, int *fx, int *fy, int *fz);
FlErrorCode MeshCartGetIsLastRank(Mesh mesh, int *lx, int *ly, int *lz);
FlErrorCode MeshCartGetBoundaryIndex(Mesh mesh, MeshCartBoundaryLocation loc, int *index);
FlErrorCode MeshCartGetCoordinateArraysRead(Mesh mesh, const double **xf, const double **yf, const double **zf); /* cart.c:467-510; GLOBAL face coordinates, borrowed */
FlErrorCode MeshGetRank(Mesh mesh, int *rank, int *size);
FlErrorCode MeshGetNumberBoundaries(Mesh mesh, int *nb);
/* meshbasic.c:93-104: "Mesh Object: type: cart" + the type's view (ASCII: this rank's sizes and index ranges, cart.c:182-201;
 * CGNS: Base / Zone / coordinates / CellInfo of a new file, cartcgns.c:8-118).  viewer NULL = standard output. */
FlErrorCode MeshView(Mesh mesh, FlucaViewer viewer);
/* meshbasic.c:114-127: CGNS viewers only (PETSC_ERR_ARG_WRONG otherwise); before MeshSetUp.  Takes the global sizes and the
 * face coordinates from the file (cartcgns.c:120-158: boundary types NONE, coordLoaded); MeshSetUp then uses them (cart.c:131-140) */
FlErrorCode MeshLoad(Mesh mesh, FlucaViewer viewer);
/* meshbasic.c: MeshCreateGlobalVector / MeshCreateMatrix.  The array is this rank's block of the DM (cells: N, 3N component-major;
 * faces: x-, y-, z-face arrays one after the other, DMStag ownership), zeroed, freed with fl_free. */
FlErrorCode MeshCreateGlobalVector(Mesh mesh, MeshDMType type, int device, double **vec_dev, int64_t *n);
FlErrorCode MeshCreateMatrix(Mesh mesh, MeshDMType rtype, MeshDMType ctype, void **mat);
FlErrorCode MeshDestroy(Mesh *mesh);

/* ---- NS ---- */
typedef enum { NS_BC_NONE, NS_BC_VELOCITY, NS_BC_PRESSURE_OUTLET, NS_BC_PERIODIC, NS_BC_SYMMETRY } NSBoundaryConditionType;
/* PURITY CONTRACT (mirror only): a callback is a function of (t, x) and of what ctx pointed to when the condition was SET.  The mirror keeps the
 * boundary values of the two most recent times (a step's t + dt is the next step's t) and evaluates a plane again only when the time, the function
 * pointer or the ctx pointer differ, or after NSSetBoundaryCondition on that boundary -- which is how a caller that changes the data behind ctx
 * between steps says so.  -ns_keep_boundary_values false restores the reference's behaviour (every use evaluates the callback). */
typedef FlErrorCode (*NSBoundaryConditionFunction)(int dim, double t, const double x[], double val[], void *ctx);
typedef struct {
  NSBoundaryConditionType     type;
  NSBoundaryConditionFunction velocity;
  void                       *ctx_velocity;
  NSBoundaryConditionFunction pressure;
  void                       *ctx_pressure;
} NSBoundaryCondition;

/* The composite vector of ns->x / ns->r / ns->sol (a Vec over the DMComposite of vdm, Sdm, sdm in the reference): device
 * arrays v (3*cells, component-major), V[3] (x-, y-, z-faces), p (cells). */
typedef struct {
  double *v, *V[3], *p;
} NSVec;
typedef enum { NS_INIT_JACOBIAN, NS_UPDATE_JACOBIAN } NSFormJacobianType; /* flucans.h:65-68 */
/* ns->J: the reference's 3x3 MATNEST [A 0 kG; -T I -R; 0 D 0] is the fl_momentum handle here (fl_abf_jacobian_mult = MatMult(J)) */
typedef fl_momentum *NSMat;

/* nsimpl.h:21-31, the same nine slots in the same order (filled for NSCNLINEAR as at cnlinear.c:177-185) */
struct _NSOps {
  FlErrorCode (*setfromoptions)(NS, int, char **);
  FlErrorCode (*setup)(NS);
  FlErrorCode (*step)(NS);
  FlErrorCode (*formjacobian)(NS, const NSVec *x, NSMat J, NSFormJacobianType type);
  FlErrorCode (*formfunction)(NS, const NSVec *x, NSVec *f);
  FlErrorCode (*destroy)(NS);
  FlErrorCode (*view)(NS, FlucaViewer);
  FlErrorCode (*viewsolution)(NS, FlucaViewer);
  FlErrorCode (*loadsolution)(NS, FlucaViewer);
};
FlErrorCode NSRegister(const char name[], FlErrorCode (*create)(NS));
FlErrorCode NSCreate(NS *ns);
FlErrorCode NSSetType(NS ns, NSType type);
FlErrorCode NSGetType(NS ns, NSType *type);
FlErrorCode NSSetMesh(NS ns, Mesh mesh);
FlErrorCode NSSetDevice(NS ns, int device);
FlErrorCode NSSetDensity(NS ns, double rho);
FlErrorCode NSSetViscosity(NS ns, double mu);
FlErrorCode NSSetTimeStepSize(NS ns, double dt);
FlErrorCode NSSetMaxSteps(NS ns, int64_t max_steps);
FlErrorCode NSSetMaxTime(NS ns, double max_time); /* -ns_max_time; NSSolve stops at whichever of the two comes first (nsbasic.c:333-334) */
FlErrorCode NSGetMaxTime(NS ns, double *max_time);
FlErrorCode NSGetMaxSteps(NS ns, int64_t *max_steps);
FlErrorCode NSGetDensity(NS ns, double *rho);
FlErrorCode NSGetViscosity(NS ns, double *mu);
FlErrorCode NSGetTimeStepSize(NS ns, double *dt);
FlErrorCode NSSetTime(NS ns, double t);
FlErrorCode NSSetTimeStep(NS ns, int64_t step);
/* -ns_error_if_step_failed (default true, nsbasic.c:46): a failed step makes NSStep return PETSC_ERR_NOT_CONVERGED (91) */
/* -ns_keep_boundary_values (mirror only; default true): the values a VELOCITY callback returned on a boundary are kept for the two most recent times, so
 * that the several boundary-condition vectors of a step and the next step's t (= this step's t + dt) do not sweep the callback again.  That presumes what
 * the reference's own repeated sweeps presume within a step -- a callback is a function of (t, x) for a fixed context.  A caller that changes what ctx
 * points to between steps passes false (values are then kept within a step only) or sets the boundary condition again. */
FlErrorCode NSSetErrorIfStepFailed(NS ns, int flg);
FlErrorCode NSGetErrorIfStepFailed(NS ns, int *flg);
/* NSConvergedReason (flucans.h:13-18): 0 ITERATING, 1 CONVERGED_TIME, 2 CONVERGED_ITS, -1 DIVERGED_NONLINEAR_SOLVE */
FlErrorCode NSGetConvergedReason(NS ns, int *reason);
FlErrorCode NSSetBoundaryCondition(NS ns, int index, NSBoundaryCondition bc);
FlErrorCode NSGetBoundaryCondition(NS ns, int index, NSBoundaryCondition *bc);
FlErrorCode NSSetFromOptions(NS ns, int argc, char **argv);
FlErrorCode NSSetUp(NS ns);
FlErrorCode NSStep(NS ns);
/* nsbasic.c:301-323: PetscUseTypeMethod(ns, formjacobian / formfunction) inside the NS_FormJacobian / NS_FormFunction events.
 * CNLinear is linear (SNESSetPicard, nsbasic.c:248): formfunction writes the right-hand side b of J x = b into f (momrhs,
 * interprhs, contrhs = 0; x is not read), formjacobian refreshes the A block of J from ns->sol0 (INIT: wires the constant blocks
 * first).  NSStep calls both through the table, so a type registered with NSRegister that replaces one of them is honoured. */
FlErrorCode NSFormJacobian(NS ns, const NSVec *x, NSMat J, NSFormJacobianType type);
FlErrorCode NSFormFunction(NS ns, const NSVec *x, NSVec *f);
FlErrorCode NSGetJacobian(NS ns, NSMat *J);                /* ns->J */
FlErrorCode NSGetSolverVectors(NS ns, NSVec *x, NSVec *r); /* ns->x, ns->r (the right-hand side the last step solved with) */
/* nsbasic.c:353-374: ASCII viewers only, "NS Object: type: ...", parameters, step and time, then the type's view; NULL = stdout */
FlErrorCode NSView(NS ns, FlucaViewer viewer);
/* nssol.c:130-150: VecView of the field links Velocity, FaceNormalVelocity, Pressure, then PetscTryTypeMethod(viewsolution)
 * (CNLinear: PressureHalfStep, cnlinear.c:146-153).  With a FlucaViewerCGNS: FlowSolution<step> of the current step and
 * time; writes the mesh first if the file is new.  viewer NULL = stdout; an ASCII viewer prints the fields' names and sizes (not 10^8 numbers).
 * Collective over the ranks. */
FlErrorCode NSViewSolution(NS ns, FlucaViewer viewer);
/* nssol.c:174-203: FlucaVecLoad of the same fields from the LAST FlowSolution of the file, PetscUseTypeMethod(loadsolution),
 * then step and time from the viewer's output sequence.  After NSSetUp; the mesh sizes must match. */
FlErrorCode NSLoadSolution(NS ns, FlucaViewer viewer);
FlErrorCode NSGetTimeStep(NS ns, int64_t *step);
FlErrorCode NSGetTime(NS ns, double *t);
FlErrorCode NSDestroy(NS *ns);

/* the Schur-complement half of PCABF: PCABFGetSubKSPs(pc, NULL, &kspS) of the reference */
FlErrorCode NSGetPoisson(NS ns, fl_poisson **poisson);
FlErrorCode NSGetSchurKSPOptions(NS ns, fl_ksp_opts **opts);
/* needs_nullspace = no boundary is a PRESSURE_OUTLET (nsbasic.c:214-231) */
FlErrorCode NSGetNeedsNullSpace(NS ns, int *needs);
/* sizes of this rank's arrays: cells, x-, y-, z-faces */
FlErrorCode NSGetLocalSizes(NS ns, int64_t out[4]);

/* PCApply_ABF minus KSPSolve(kspA): given the intermediate velocities v* (cells, any may be NULL) and V* (faces) on the
 * device, Srhs = contrhs - D V*, dp = S^-1 Srhs, v = v* - G dp, V = V* - Gst dp  (abfpc.c:73-101, Ainv = ID) */
FlErrorCode NSPressureCorrection(NS ns, double *vstar_dev[3], double *Vstar_dev[3], const double *contrhs_dev, double *dp_dev, fl_ksp_stats *stats);
/* NSStep / NSSolve (nsbasic.c:276-350) with the CNLinear step of cnlinearcart3d.c:2807-2863 (+ NSFormJacobian :2930-2941,
 * NSFormFunction :2945-3060) on device arrays, all four boundary types.  The outer KSP of ns->snes is -ns_ksp_type gmres (default as in the reference: right-preconditioned
 * with PC_ABF, -ns_ksp_rtol 1e-5 on the unpreconditioned norm, nssol.c:24-25, -ns_ksp_gmres_restart 30), richardson
 * (x += PCApply_ABF(f - J x)) or preonly.  NSGetSolutionArrays hands out the device arrays of ns->sol (velocity 3*cells
 * component-major, face-normal velocity per axis, pressure) so that the caller can set the initial condition. */
FlErrorCode NSSolve(NS ns);
/* Immersed boundary by explicit direct forcing -- build-defined, the reference has none (THEORY_GUIDE.md:130-132): every
 * step adds spread(U_target - interp(v0)) to momrhs.  kind: fl_delta_kind; L markers X,Y,Z with volumes dV (device arrays
 * owned by the caller, uniform grid spacing required); Utarget_dev: 3*L target velocities or NULL for a body at rest. */
FlErrorCode NSSetImmersedBoundary(NS ns, int kind, int64_t L, const double *X_dev, const double *Y_dev, const double *Z_dev, const double *dV_dev, const double *Utarget_dev);
FlErrorCode NSGetSolutionArrays(NS ns, double **v_dev, double *V_dev[3], double **p_dev);
FlErrorCode NSGetPressureHalfStep(NS ns, double **phalf_dev); /* cnl->phalf, the vector named "PressureHalfStep" (cnlinear.c:54) */
FlErrorCode NSGetMesh(NS ns, Mesh *mesh);
FlErrorCode NSGetDevice(NS ns, int *device);
FlErrorCode NSSetTimeStepAndTime(NS ns, int64_t step, double t); /* the last thing NSLoadSolution does (nssol.c:199-201) */
FlErrorCode NSBarrier(NS ns);                                    /* MPI_Barrier on the object's communicator */
/* nsmon.c:5-45.  NSSolve calls NSMonitor before every step and once after the last (nsbasic.c:337-345); NSDestroy cancels. */
FlErrorCode NSMonitorSet(NS ns, FlErrorCode (*monitor)(NS, void *), void *ctx, FlErrorCode (*ctxdestroy)(void **));
FlErrorCode NSMonitorCancel(NS ns);
FlErrorCode NSMonitor(NS ns);
FlErrorCode NSGetLinearSolveInfo(NS ns, int *its, double *rnorm, int *reason);
/* norm of the right-hand side the last outer solve started from (the reference norm of its rtol test, nssol.c:24-25: the
 * unpreconditioned norm) and the residual norm it stopped at: rnorm / rnorm0 is what -ns_ksp_rtol is compared with */
FlErrorCode NSGetLinearSolveResidualNorms(NS ns, double *rnorm0, double *rnorm);
/* 1 when libroctx64 was found: NSSetUp, NSStep, NSFormFunction and NSFormJacobian then emit roctx ranges under the names of the
 * reference's PetscLogEvents (nspkg.c:21-24); rocprofv3 --marker-trace shows them */
FlErrorCode FlucaTraceEnabled(void);
/* kspA / kspS iterations summed over the outer iterations of the last step */
FlErrorCode NSGetInnerIterations(NS ns, int *momentum_its, int *schur_its);
/* The A block of NSFormJacobian (cnlinearcart3d.c:2930-2941): hands over sol0's face-normal velocity V0 (3 face arrays)
 * and cnl->v0interp (9 face arrays, component c on the faces of axis d at [c*3+d]); A = I + dt C - (mu dt / 2 rho) L is
 * applied matrix-free from then on.  Call once per time step, before NSApplyPreconditioner. */
FlErrorCode NSSetPreviousState(NS ns, const double *const V0_dev[3], const double *const v0interp_dev[9]);
/* the whole PCApply_ABF (abfpc.c:48-111): v* = A^-1 momrhs, V* = interprhs + T v*, p = S^-1(contrhs - D V*),
 * v = v* - G p, V = V* - Gst p.  v: 3*cells component-major.  stats[0] = kspA, stats[1] = kspS (may be NULL).
 * -ns_abf_momentum_guess_previous (mirror only, NSStep with -ns_ksp_type preonly | richardson): the FIRST PCApply_ABF of a time step starts kspA from the
 * previous velocity instead of from zero (fl_ksp_opts.initial_guess_nonzero; convergence against || M momrhs || as KSPConvergedDefault does with a non-zero
 * guess): momrhs - A v0 = O(dt) momrhs, so the solve needs the iterations of a reduction by rtol / O(dt) only.  The answer meets the same test.
 * -ns_abf_momentum_guess_extrapolate: the same with the guess 2 v^n - v^(n-1) (O(dt^2) away from the answer in a smooth flow; the first step uses v^n).
 * Options: -ns_abf_momentum_ksp_type bcgs|gmres|chebyshev (gmres = PETSc's default type for kspA, restart
 * -ns_abf_momentum_ksp_gmres_restart 30; chebyshev: -ns_abf_momentum_ksp_chebyshev_eigenvalues emin,emax or the Gershgorin disc,
 * -ns_abf_momentum_ksp_norm_type preconditioned|unpreconditioned|none), -ns_abf_momentum_pc_type jacobi|none (ilu, PETSc's default PC, has no
 * matrix-free form: PETSC_ERR_SUP), -ns_abf_momentum_ksp_{rtol,atol,divtol,max_it}. */
FlErrorCode NSApplyPreconditioner(NS ns, const double *momrhs_dev, const double *const interprhs_dev[3], const double *contrhs_dev, double *v_dev, double *const V_dev[3], double *p_dev, fl_ksp_stats stats[2]);
FlErrorCode NSGetMomentumKSPOptions(NS ns, fl_ksp_opts **opts);
FlErrorCode NSGetImmersedBoundary(NS ns, fl_ibm **ibm);
FlErrorCode NSGetMomentum(NS ns, fl_momentum **momentum);
/* p, phalf update of the time step, then ++step, t += dt (cnlinearcart3d.c:2846-2854, nsbasic.c:288-291) */
FlErrorCode NSUpdatePressure(NS ns, const double *dp_dev, const double *p0_dev, double *phalf_dev, double *p_dev);
/* Gst boundary vector: evaluates the PRESSURE_OUTLET callbacks at the boundary face centres at time t (host), writes
 * coeff * p_b into the boundary faces of V_dev[axis] (cnlinearcart3d.c:2602-2805) */
FlErrorCode NSComputeStaggeredPressureGradientBC(NS ns, double t, double *V_dev[3]);

#ifdef __cplusplus
}
#endif
#endif
