/*
 * abfpc_hip.c -- the Schur-complement path of Fluca's PCABF preconditioner on libflucahip.so (MI355X).
 *
 * Companion of fluca/src/ns/utils/abfpc/abfpc.c in thecasterian/fluca: it provides PCSetUp_ABF_HIP / PCApply_ABF_HIP, which
 * replace PCSetUp_ABF (abfpc.c:113-182) and PCApply_ABF (abfpc.c:48-111) when both A-inverse types are ID (the reference's
 * default, abfpc.c:328-329).  What moves to the GPU is everything behind the momentum solve:
 *     Srhs = contrhs - D V*            (abfpc.c:75-76)      fl_poisson_rhs
 *     p    = S^-1 Srhs                 (abfpc.c:77)         fl_poisson_solve  -- S is never assembled (abfpc.c:150-171 goes away)
 *     v    = v* - G p,  V = V* - Gst p (abfpc.c:80-101)     fl_poisson_project
 * KSPSolve(kspA) and V* = interprhs + T v* stay PETSc calls on the sub-matrices, as in the reference.
 *
 * How a maintainer wires it in (five edits, nothing else changes):
 *   1. abfpc.c, struct PC_ABF: add the member      struct PCABFHip *hip;
 *   2. abfpc.c, before PCCreate_ABF:               #include "abfpc_hip.c"      (this file sees PC_ABF that way)
 *   3. abfpc.c, PCCreate_ABF:                      abf->hip = NULL;  and compose "PCABFSetNS_C" -> PCABFSetNS_ABF
 *                                                  pc->ops->apply = PCApply_ABF_HIP;  pc->ops->setup = PCSetUp_ABF_HIP;
 *   4. abfpc.c, PCReset_ABF and PCDestroy_ABF:     PetscCall(PCABFHipDestroy_Private(&abf->hip));
 *   5. nsbasic.c:262, after PCABFSetFields(...):   PetscCall(PCABFSetNS(pc, ns));     (declare both in flucans.h)
 * Build: -I<fluca_amd>/include, link -lflucahip.  Options: the -ns_abf_schur_ksp_* / -ns_abf_schur_pc_type options keep
 * working -- they are read off abf->kspS and translated into fl_ksp_opts -- plus -pc_abf_hip <bool> (default true).
 *
 * This file is written against PETSc >= 3.23 (fluca/CMakeLists.txt:9-11) and the reference's public headers; PETSc is not
 * installed where libflucahip.so was developed, so it has NOT been compiled against PETSc there.  It IS parsed on every build by
 * tools/check_contrib.sh (gcc -fsyntax-only against the real fluca_hip.h and a declarations-only stand-in for the PETSc names it
 * uses; the first run found a member access on the opaque NS that a real build would have refused).  Everything it needs from
 * the library is exercised by tests/test_gpu_layout.py (the DMStag orderings) and tests/test_gpu_poisson.py (the three calls).
 */
#include <fluca_hip.h>
#include <petscdmstag.h>

typedef struct PCABFHip {
  NS           ns;                 /* borrowed: grid coordinates, boundary conditions, dt / rho */
  PetscBool    enabled;
  fl_poisson  *flh;
  int          device;
  PetscReal    kappa;              /* dt / rho the handle was created with */
  PetscInt     ncell, nface[3];
  /* device arrays in the library's layout */
  double      *d_V[3], *d_v[3], *d_p, *d_Srhs, *d_contrhs;
  /* device staging for host Vecs (unused when the Vecs live on the device) */
  double      *d_stage;
  PetscInt     stagelen;
} PCABFHip;

#define FLCall(pc, call) \
  do { \
    int flrc_ = (call); \
    PetscCheck(flrc_ == 0, PetscObjectComm((PetscObject)(pc)), (PetscErrorCode)(-flrc_), "libflucahip: %s failed with %d", #call, flrc_); \
  } while (0)

static PetscErrorCode PCABFHipFree_Private(PCABFHip *hip)
{
  PetscInt d;

  PetscFunctionBegin;
  for (d = 0; d < 3; ++d) {
    if (hip->d_V[d]) (void)fl_free(hip->device, hip->d_V[d]);
    if (hip->d_v[d]) (void)fl_free(hip->device, hip->d_v[d]);
    hip->d_V[d] = hip->d_v[d] = NULL;
  }
  if (hip->d_p) (void)fl_free(hip->device, hip->d_p);
  if (hip->d_Srhs) (void)fl_free(hip->device, hip->d_Srhs);
  if (hip->d_contrhs) (void)fl_free(hip->device, hip->d_contrhs);
  if (hip->d_stage) (void)fl_free(hip->device, hip->d_stage);
  hip->d_p = hip->d_Srhs = hip->d_contrhs = hip->d_stage = NULL;
  hip->stagelen = 0;
  if (hip->flh) (void)fl_poisson_destroy(hip->flh);
  hip->flh = NULL;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCABFHipDestroy_Private(PCABFHip **hip)
{
  PetscFunctionBegin;
  if (!*hip) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PCABFHipFree_Private(*hip));
  PetscCall(PetscFree(*hip));
  *hip = NULL;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCABFSetNS_ABF(PC pc, NS ns)
{
  PC_ABF *abf = (PC_ABF *)pc->data;

  PetscFunctionBegin;
  if (!abf->hip) {
    PetscCall(PetscNew(&abf->hip));
    abf->hip->enabled = PETSC_TRUE;
  }
  abf->hip->ns = ns; /* borrowed: the NS owns the SNES that owns this PC */
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCABFSetNS(PC pc, NS ns)
{
  PetscFunctionBegin;
  PetscValidHeaderSpecific(pc, PC_CLASSID, 1);
  PetscTryMethod(pc, "PCABFSetNS_C", (PC, NS), (pc, ns));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* The handle: global 1-D coordinates, boundary types, this rank's block of the DMStag decomposition. */
static PetscErrorCode PCABFHipCreateHandle_Private(PC pc, DM sdm)
{
  PC_ABF             *abf = (PC_ABF *)pc->data;
  PCABFHip           *hip = abf->hip;
  MPI_Comm            comm;
  PetscMPIInt         rank, size;
  PetscInt            N[3], R[3], x[3], m[3], d, i, iprev, ielem, nb, b;
  const PetscScalar **arrc[3];
  PetscReal          *xf[3], *xc[3], dt, rho;
  fl_grid             grid;
  fl_decomp           dec;
  int                 bc[6];

  PetscFunctionBegin;
  PetscCall(PetscObjectGetComm((PetscObject)pc, &comm));
  PetscCallMPI(MPI_Comm_rank(comm, &rank));
  PetscCallMPI(MPI_Comm_size(comm, &size));
  PetscCall(DMStagGetGlobalSizes(sdm, &N[0], &N[1], &N[2]));
  PetscCall(DMStagGetNumRanks(sdm, &R[0], &R[1], &R[2]));
  PetscCall(DMStagGetCorners(sdm, &x[0], &x[1], &x[2], &m[0], &m[1], &m[2], NULL, NULL, NULL));

  /* GLOBAL coordinates of every axis: each rank knows faces x[d]..x[d]+m[d] and the centres of its cells (the arrays of
     MeshCartGetCoordinateArraysRead, cart.c:475-481); ranks that share a range hold identical numbers, so a MAX over the
     communicator of arrays initialised to -inf assembles the whole axis */
  PetscCall(DMStagGetProductCoordinateArraysRead(sdm, &arrc[0], &arrc[1], &arrc[2]));
  PetscCall(DMStagGetProductCoordinateLocationSlot(sdm, DMSTAG_LEFT, &iprev));
  PetscCall(DMStagGetProductCoordinateLocationSlot(sdm, DMSTAG_ELEMENT, &ielem));
  for (d = 0; d < 3; ++d) {
    PetscCall(PetscMalloc2(N[d] + 1, &xf[d], N[d], &xc[d]));
    for (i = 0; i <= N[d]; ++i) xf[d][i] = PETSC_MIN_REAL;
    for (i = 0; i < N[d]; ++i) xc[d][i] = PETSC_MIN_REAL;
    for (i = x[d]; i <= x[d] + m[d]; ++i) xf[d][i] = PetscRealPart(arrc[d][i][iprev]);
    for (i = x[d]; i < x[d] + m[d]; ++i) xc[d][i] = PetscRealPart(arrc[d][i][ielem]);
    PetscCallMPI(MPIU_Allreduce(MPI_IN_PLACE, xf[d], (PetscMPIInt)(N[d] + 1), MPIU_REAL, MPIU_MAX, comm));
    PetscCallMPI(MPIU_Allreduce(MPI_IN_PLACE, xc[d], (PetscMPIInt)N[d], MPIU_REAL, MPIU_MAX, comm));
    grid.n[d]  = N[d];
    grid.xf[d] = xf[d];
    grid.xc[d] = xc[d];
  }
  PetscCall(DMStagRestoreProductCoordinateArraysRead(sdm, &arrc[0], &arrc[1], &arrc[2]));

  /* boundary types: NSBoundaryConditionType and fl_bc share their values (flucansbc.h:5-11); boundary index order
     left, right, down, up, back, front (MeshCartGetBoundaryIndex, cart.c:564-591) */
  {
    Mesh mesh;
    PetscCall(NSGetMesh(hip->ns, &mesh)); /* borrowed (nsopts.c) */
    PetscCall(MeshGetNumberBoundaries(mesh, &nb));
  }
  PetscCheck(nb == 6, comm, PETSC_ERR_SUP, "libflucahip handles 3-D Cartesian meshes");
  for (b = 0; b < 6; ++b) {
    NSBoundaryCondition c;
    PetscCall(NSGetBoundaryCondition(hip->ns, b, &c));
    bc[b] = (int)c.type;
  }

  /* this rank's block: DMStag numbers ranks x fastest (rank = i + j R0 + k R0 R1) */
  dec.ranks[0] = (int)R[0];
  dec.ranks[1] = (int)R[1];
  dec.ranks[2] = (int)R[2];
  dec.coord[0] = (int)(rank % R[0]);
  dec.coord[1] = (int)((rank / R[0]) % R[1]);
  dec.coord[2] = (int)(rank / (R[0] * R[1]));
  for (d = 0; d < 3; ++d) {
    dec.lo[d]  = x[d];
    dec.len[d] = m[d];
  }

  PetscCall(NSGetTimeStepSize(hip->ns, &dt));
  PetscCall(NSGetDensity(hip->ns, &rho));
  hip->kappa = dt / rho; /* the MatScale of G and Gst, cnlinearcart3d.c:2890,2907 */
  FLCall(pc, fl_current_device(&hip->device));
  FLCall(pc, fl_poisson_create(&grid, bc, (double)hip->kappa, size > 1 ? &dec : NULL, hip->device, &hip->flh));
  for (d = 0; d < 3; ++d) PetscCall(PetscFree2(xf[d], xc[d]));

  if (size > 1) {
    /* halo exchange + all-reduce inside the library over RCCL: rank 0 makes the id, MPI carries it */
    char id[FL_UNIQUE_ID_BYTES];
    if (rank == 0) FLCall(pc, fl_comm_unique_id(id));
    PetscCallMPI(MPI_Bcast(id, FL_UNIQUE_ID_BYTES, MPI_BYTE, 0, comm));
    FLCall(pc, fl_poisson_comm_init_rccl(hip->flh, id, (int)rank, (int)size));
  }

  {
    int64_t sizes[4];
    FLCall(pc, fl_poisson_sizes(hip->flh, sizes));
    hip->ncell = (PetscInt)sizes[0];
    for (d = 0; d < 3; ++d) {
      hip->nface[d] = (PetscInt)sizes[1 + d];
      FLCall(pc, fl_malloc(hip->device, sizeof(double) * (size_t)sizes[1 + d], (void **)&hip->d_V[d]));
      FLCall(pc, fl_malloc(hip->device, sizeof(double) * (size_t)sizes[0], (void **)&hip->d_v[d]));
    }
    FLCall(pc, fl_malloc(hip->device, sizeof(double) * (size_t)sizes[0], (void **)&hip->d_p));
    FLCall(pc, fl_malloc(hip->device, sizeof(double) * (size_t)sizes[0], (void **)&hip->d_Srhs));
    FLCall(pc, fl_malloc(hip->device, sizeof(double) * (size_t)sizes[0], (void **)&hip->d_contrhs));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* -ns_abf_schur_ksp_* / -ns_abf_schur_pc_type, as set on abf->kspS, into fl_ksp_opts */
static PetscErrorCode PCABFHipOptions_Private(PC pc, fl_ksp_opts *o)
{
  PC_ABF     *abf = (PC_ABF *)pc->data;
  PC          spc;
  PetscBool   is;
  PetscReal   rtol, atol, dtol;
  PetscInt    maxit;
  KSPNormType nt;

  PetscFunctionBegin;
  fl_ksp_opts_default(o);
  PetscCall(KSPGetTolerances(abf->kspS, &rtol, &atol, &dtol, &maxit));
  o->rtol  = (double)rtol;
  o->atol  = (double)atol;
  o->dtol  = (double)dtol;
  o->maxit = (int)maxit;
  o->type  = FL_KSP_CG; /* PETSc's default for kspS would be GMRES; S is symmetric positive semi-definite on uniform grids */
  PetscCall(PetscObjectTypeCompare((PetscObject)abf->kspS, KSPBCGS, &is));
  if (is) o->type = FL_KSP_BCGS;
  PetscCall(PetscObjectTypeCompare((PetscObject)abf->kspS, KSPCHEBYSHEV, &is));
  if (is) o->type = FL_KSP_CHEBYSHEV;
  PetscCall(KSPGetPC(abf->kspS, &spc));
  o->pc = FL_PC_JACOBI;
  PetscCall(PetscObjectTypeCompare((PetscObject)spc, PCNONE, &is));
  if (is) o->pc = FL_PC_NONE;
  PetscCall(PetscObjectTypeCompare((PetscObject)spc, PCMG, &is));
  if (is) o->pc = FL_PC_MG;
  PetscCall(KSPGetNormType(abf->kspS, &nt));
  switch (nt) {
  case KSP_NORM_UNPRECONDITIONED:
    o->norm_type = FL_NORM_UNPRECONDITIONED;
    break;
  case KSP_NORM_NATURAL:
    o->norm_type = FL_NORM_NATURAL;
    break;
  case KSP_NORM_NONE:
    o->norm_type = FL_NORM_NONE;
    break;
  default:
    o->norm_type = FL_NORM_PRECONDITIONED;
  }
  o->remove_nullspace = abf->nullspace ? 1 : 0; /* the constant null space of abfpc.c:173-177 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* Device address of a Vec's array: the Vec's own when it lives on the device, otherwise a staged copy. */
static PetscErrorCode PCABFHipStageIn_Private(PC pc, Vec v, PetscInt offset, const double **dev)
{
  PCABFHip          *hip = ((PC_ABF *)pc->data)->hip;
  const PetscScalar *a;
  PetscMemType       mt;
  PetscInt           n;

  PetscFunctionBegin;
  PetscCall(VecGetLocalSize(v, &n));
  PetscCall(VecGetArrayReadAndMemType(v, &a, &mt));
  if (PetscMemTypeDevice(mt)) *dev = (const double *)a;
  else {
    FLCall(pc, fl_memcpy_h2d(hip->device, hip->d_stage + offset, a, sizeof(double) * (size_t)n));
    *dev = hip->d_stage + offset;
  }
  PetscCall(VecRestoreArrayReadAndMemType(v, &a));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCABFHipStageOutBegin_Private(PC pc, Vec v, PetscInt offset, PetscScalar **a, PetscMemType *mt, double **dev)
{
  PCABFHip *hip = ((PC_ABF *)pc->data)->hip;

  PetscFunctionBegin;
  PetscCall(VecGetArrayWriteAndMemType(v, a, mt));
  *dev = PetscMemTypeDevice(*mt) ? (double *)*a : hip->d_stage + offset;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCABFHipStageOutEnd_Private(PC pc, Vec v, PetscScalar **a, PetscMemType mt, const double *dev)
{
  PCABFHip *hip = ((PC_ABF *)pc->data)->hip;
  PetscInt  n;

  PetscFunctionBegin;
  PetscCall(VecGetLocalSize(v, &n));
  FLCall(pc, fl_poisson_synchronize(hip->flh));
  if (!PetscMemTypeDevice(mt)) FLCall(pc, fl_memcpy_d2h(hip->device, *a, dev, sizeof(double) * (size_t)n));
  PetscCall(VecRestoreArrayWriteAndMemType(v, a));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetUp_ABF_HIP(PC pc)
{
  PC_ABF   *abf = (PC_ABF *)pc->data;
  PetscBool isnest, usehip = PETSC_TRUE;
  PetscInt  m, n;
  IS       *rowis, *colis;

  PetscFunctionBegin;
  PetscCall(PetscOptionsGetBool(((PetscObject)pc)->options, ((PetscObject)pc)->prefix, "-pc_abf_hip", &usehip, NULL));
  if (!abf->hip || !abf->hip->ns || !usehip || abf->schurainv != PC_ABF_AINV_ID || abf->upperainv != PC_ABF_AINV_ID) {
    if (abf->hip) abf->hip->enabled = PETSC_FALSE;
    PetscCall(PCSetUp_ABF(pc)); /* the reference's own path: assembled S */
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  abf->hip->enabled = PETSC_TRUE;
  PetscCall(PetscObjectTypeCompare((PetscObject)pc->pmat, MATNEST, &isnest));
  PetscCheck(isnest, PetscObjectComm((PetscObject)pc), PETSC_ERR_ARG_WRONG, "Only Pmat of MATNEST type is supported");

  /* A and -T are still needed: the momentum solve and V* = interprhs + T v* stay with PETSc.  D, G, -R and S are not. */
  PetscCall(MatDestroy(&abf->A));
  PetscCall(MatDestroy(&abf->negT));
  PetscCall(VecDestroy(&abf->vstar));
  PetscCall(VecDestroy(&abf->Vstar));
  PetscCall(MatNestGetSize(pc->pmat, &m, &n));
  PetscCall(PetscMalloc2(m, &rowis, n, &colis));
  PetscCall(MatNestGetISs(pc->pmat, rowis, colis));
  PetscCall(MatCreateSubMatrix(pc->mat, rowis[abf->vidx], colis[abf->vidx], MAT_INITIAL_MATRIX, &abf->A));
  PetscCall(MatCreateSubMatrix(pc->mat, rowis[abf->Vidx], colis[abf->vidx], MAT_INITIAL_MATRIX, &abf->negT));
  PetscCall(PetscFree2(rowis, colis));
  PetscCall(KSPSetOperators(abf->kspA, abf->A, abf->A));

  {
    MatNullSpace nullspace;
    PetscCall(MatNullSpaceDestroy(&abf->nullspace));
    PetscCall(MatGetNullSpace(pc->mat, &nullspace));
    if (nullspace) PetscCall(MatNullSpaceCreate(PetscObjectComm((PetscObject)pc->mat), PETSC_TRUE, 0, NULL, &abf->nullspace));
  }

  /* S = -kappa D Gst depends on the grid, the boundary types and dt / rho only: the handle survives every PCSetUp of a run
     with a fixed time step; a new dt / rho re-creates it */
  if (abf->hip->flh) {
    PetscReal dt, rho;
    PetscCall(NSGetTimeStepSize(abf->hip->ns, &dt));
    PetscCall(NSGetDensity(abf->hip->ns, &rho));
    if (dt / rho != abf->hip->kappa) PetscCall(PCABFHipFree_Private(abf->hip));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCApply_ABF_HIP(PC pc, Vec b, Vec x)
{
  PC_ABF       *abf = (PC_ABF *)pc->data;
  PCABFHip     *hip = abf->hip;
  PetscInt      m, n, d, nV, nv, np;
  IS           *rowis, *colis;
  Vec           momrhs, interprhs, contrhs, v, V, p;
  DM            sdm, Sdm, vdm;
  PetscInt      dofS[4], dofv[4];
  int           idofS[4], idofv[4];
  const double *gV, *gv, *gc;
  double       *ov, *oV, *op;
  PetscScalar  *av, *aV, *ap;
  PetscMemType  mv, mV, mp;
  fl_ksp_opts   opts;
  fl_ksp_stats  stats;

  PetscFunctionBegin;
  if (!hip || !hip->enabled) {
    PetscCall(PCApply_ABF(pc, b, x));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCall(MatNestGetSize(pc->pmat, &m, &n));
  PetscCall(PetscMalloc2(m, &rowis, n, &colis));
  PetscCall(MatNestGetISs(pc->pmat, rowis, colis));
  PetscCall(VecGetSubVector(b, rowis[abf->vidx], &momrhs));
  PetscCall(VecGetSubVector(b, rowis[abf->Vidx], &interprhs));
  PetscCall(VecGetSubVector(b, rowis[abf->pidx], &contrhs));
  PetscCall(VecGetSubVector(x, colis[abf->vidx], &v));
  PetscCall(VecGetSubVector(x, colis[abf->Vidx], &V));
  PetscCall(VecGetSubVector(x, colis[abf->pidx], &p));
  if (!abf->vstar) PetscCall(MatCreateVecs(abf->A, &abf->vstar, NULL));
  if (!abf->Vstar) PetscCall(MatCreateVecs(abf->negT, NULL, &abf->Vstar));

  PetscCall(VecGetDM(p, &sdm));
  PetscCall(VecGetDM(V, &Sdm));
  PetscCall(VecGetDM(v, &vdm));
  PetscCheck(sdm && Sdm && vdm, PetscObjectComm((PetscObject)pc), PETSC_ERR_ARG_WRONGSTATE, "The solution sub-vectors carry no DM");
  if (!hip->flh) PetscCall(PCABFHipCreateHandle_Private(pc, sdm));
  PetscCall(DMStagGetDOF(Sdm, &dofS[0], &dofS[1], &dofS[2], &dofS[3])); /* 0,0,1,0  (cart.c:108) */
  PetscCall(DMStagGetDOF(vdm, &dofv[0], &dofv[1], &dofv[2], &dofv[3])); /* 0,0,0,3  (cart.c:107) */
  for (d = 0; d < 4; ++d) {
    idofS[d] = (int)dofS[d];
    idofv[d] = (int)dofv[d];
  }
  PetscCall(VecGetLocalSize(V, &nV));
  PetscCall(VecGetLocalSize(v, &nv));
  PetscCall(VecGetLocalSize(p, &np));
  {
    int64_t want;
    FLCall(pc, fl_dmstag_global_entries(hip->flh, idofS, &want));
    PetscCheck((PetscInt)want == nV && np == hip->ncell && nv == 3 * hip->ncell, PetscObjectComm((PetscObject)pc), PETSC_ERR_ARG_SIZ, "DMStag vector sizes do not match the library's block");
  }
  if (hip->stagelen < nV + nv + np) { /* one staging buffer for [V | v | p], only touched when the Vecs are host Vecs */
    if (hip->d_stage) FLCall(pc, fl_free(hip->device, hip->d_stage));
    FLCall(pc, fl_malloc(hip->device, sizeof(double) * (size_t)(nV + nv + np), (void **)&hip->d_stage));
    hip->stagelen = nV + nv + np;
  }

  /* Stage 1: the lower triangular solve (abfpc.c:71-77) */
  PetscCall(KSPSolve(abf->kspA, momrhs, abf->vstar));
  PetscCall(MatMult(abf->negT, abf->vstar, abf->Vstar));
  PetscCall(VecAYPX(abf->Vstar, -1., interprhs));
  PetscCall(PCABFHipStageIn_Private(pc, abf->Vstar, 0, &gV));
  PetscCall(PCABFHipStageIn_Private(pc, abf->vstar, nV, &gv));
  PetscCall(PCABFHipStageIn_Private(pc, contrhs, nV + nv, &gc));
  for (d = 0; d < 3; ++d) {
    FLCall(pc, fl_layout_from_dmstag_global(hip->flh, idofS, 1 + (int)d, 0, gV, hip->d_V[d])); /* LEFT, DOWN, BACK faces */
    FLCall(pc, fl_layout_from_dmstag_global(hip->flh, idofv, 0, (int)d, gv, hip->d_v[d]));     /* component d of a cell */
  }
  FLCall(pc, fl_vec_lincomb(hip->flh, (int64_t)np, 1., gc, 0., NULL, hip->d_contrhs)); /* sdm: one dof per element, already x fastest */
  FLCall(pc, fl_poisson_rhs(hip->flh, hip->d_V[0], hip->d_V[1], hip->d_V[2], hip->d_contrhs, hip->d_Srhs));
  PetscCall(PCABFHipOptions_Private(pc, &opts));
  FLCall(pc, fl_poisson_solve(hip->flh, hip->d_Srhs, hip->d_p, &opts, &stats));
  {
    PetscBool errorifnot;
    PetscCall(KSPGetErrorIfNotConverged(abf->kspS, &errorifnot));
    PetscCheck(stats.reason > 0 || !errorifnot, PetscObjectComm((PetscObject)pc), PETSC_ERR_NOT_CONVERGED, "KSPSolve(kspS) on the GPU did not converge, reason %d after %d iterations", stats.reason, stats.iters);
  }

  /* Stage 2: the upper triangular solve (abfpc.c:79-101):  v = v* - kappa G p,  V = V* - kappa Gst p */
  FLCall(pc, fl_poisson_project(hip->flh, hip->d_p, hip->d_v[0], hip->d_v[1], hip->d_v[2], hip->d_V[0], hip->d_V[1], hip->d_V[2]));

  PetscCall(PCABFHipStageOutBegin_Private(pc, V, 0, &aV, &mV, &oV));
  PetscCall(PCABFHipStageOutBegin_Private(pc, v, nV, &av, &mv, &ov));
  PetscCall(PCABFHipStageOutBegin_Private(pc, p, nV + nv, &ap, &mp, &op));
  for (d = 0; d < 3; ++d) {
    FLCall(pc, fl_layout_to_dmstag_global(hip->flh, idofS, 1 + (int)d, 0, hip->d_V[d], oV));
    FLCall(pc, fl_layout_to_dmstag_global(hip->flh, idofv, 0, (int)d, hip->d_v[d], ov));
  }
  FLCall(pc, fl_vec_lincomb(hip->flh, (int64_t)np, 1., hip->d_p, 0., NULL, op));
  PetscCall(PCABFHipStageOutEnd_Private(pc, V, &aV, mV, oV));
  PetscCall(PCABFHipStageOutEnd_Private(pc, v, &av, mv, ov));
  PetscCall(PCABFHipStageOutEnd_Private(pc, p, &ap, mp, op));

  PetscCall(VecRestoreSubVector(b, rowis[abf->vidx], &momrhs));
  PetscCall(VecRestoreSubVector(b, rowis[abf->Vidx], &interprhs));
  PetscCall(VecRestoreSubVector(b, rowis[abf->pidx], &contrhs));
  PetscCall(VecRestoreSubVector(x, colis[abf->vidx], &v));
  PetscCall(VecRestoreSubVector(x, colis[abf->Vidx], &V));
  PetscCall(VecRestoreSubVector(x, colis[abf->pidx], &p));
  PetscCall(PetscFree2(rowis, colis));
  PetscFunctionReturn(PETSC_SUCCESS);
}
