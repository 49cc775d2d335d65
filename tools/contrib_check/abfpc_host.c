/*
 * NOT the reference's abfpc.c.  The handful of declarations abfpc.c has in scope at the point where INTEGRATION.md section 2
 * tells a maintainer to write `#include "abfpc_hip.c"`: the public Fluca types and prototypes the binding calls (names and
 * argument lists as in fluca/include/flucans.h:34-50,98-107, flucansbc.h:5-22, flucamesh.h:38 -- declarations restated, no
 * code), struct PC_ABF with the members the binding touches (abfpc.c:6-31 plus the `hip` member of step 1), and the two
 * reference functions it falls back to.  Parsed by tools/check_contrib.sh with gcc -fsyntax-only; never compiled to an object.
 */
#include <petscdmstag.h> /* tools/contrib_check/petscdmstag.h: declarations only, NOT PETSc */

typedef struct _p_Mesh *Mesh;
typedef struct _p_NS   *NS;
typedef enum { NS_BC_NONE, NS_BC_VELOCITY, NS_BC_PRESSURE_OUTLET, NS_BC_PERIODIC, NS_BC_SYMMETRY } NSBoundaryConditionType;
typedef PetscErrorCode (*NSBoundaryConditionFunction)(PetscInt, PetscReal, const PetscReal[], PetscScalar[], void *);
typedef struct {
  NSBoundaryConditionType     type;
  NSBoundaryConditionFunction velocity;
  void                       *ctx_velocity;
  NSBoundaryConditionFunction pressure;
  void                       *ctx_pressure;
} NSBoundaryCondition;
typedef enum { PC_ABF_AINV_ID, PC_ABF_AINV_DIAG, PC_ABF_AINV_ROWSUM } PCABFAinvType;
PetscErrorCode NSGetMesh(NS, Mesh *);
PetscErrorCode NSGetDensity(NS, PetscReal *);
PetscErrorCode NSGetTimeStepSize(NS, PetscReal *);
PetscErrorCode NSGetBoundaryCondition(NS, PetscInt, NSBoundaryCondition *);
PetscErrorCode MeshGetNumberBoundaries(Mesh, PetscInt *);

struct PCABFHip;
typedef struct {
  PetscInt         vidx, Vidx, pidx;
  PCABFAinvType    schurainv, upperainv;
  KSP              kspA, kspS;
  Mat              A, negT, G, D, negR, S;
  MatNullSpace     nullspace;
  Vec              Adiag, vstar, Vstar, Srhs, invA2Gp, negRp;
  struct PCABFHip *hip; /* INTEGRATION.md section 2, step 1 */
} PC_ABF;
static PetscErrorCode PCApply_ABF(PC pc, Vec b, Vec x);
static PetscErrorCode PCSetUp_ABF(PC pc);

#include "../../contrib/abfpc_hip.c"

/* bodies of the two reference functions live in abfpc.c; here they only have to exist for the parser */
static PetscErrorCode PCApply_ABF(PC pc, Vec b, Vec x)
{
  (void)pc; (void)b; (void)x;
  return PETSC_SUCCESS;
}
static PetscErrorCode PCSetUp_ABF(PC pc)
{
  (void)pc;
  return PETSC_SUCCESS;
}
/* step 3 of the wiring: the ops table takes the two new functions */
PetscErrorCode PCCreate_ABF_wiring_check(PC pc)
{
  pc->ops->apply = PCApply_ABF_HIP;
  pc->ops->setup = PCSetUp_ABF_HIP;
  return PCABFHipDestroy_Private(&((PC_ABF *)pc->data)->hip);
}
