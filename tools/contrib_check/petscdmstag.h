/*
 * NOT PETSc.  Declarations only, written for ONE purpose: to let `gcc -fsyntax-only` parse contrib/abfpc_hip.c in an image
 * that has no PETSc (tools/check_contrib.sh).  It lists exactly the PETSc / DMStag / MPI names that file uses, with the
 * argument lists of PETSc's public manual pages (>= 3.23) as far as they are remembered; nothing here is an implementation,
 * nothing here is linked, and a clean parse proves only what INTEGRATION.md section 2 says it proves: no typos, every call
 * into fluca_hip.h has the declared argument count and compatible types, every name the file relies on is accounted for.
 * It does not prove that the file compiles against a real PETSc, let alone that it runs.
 */
#ifndef FLUCA_CONTRIB_CHECK_PETSC_DECLS_H
#define FLUCA_CONTRIB_CHECK_PETSC_DECLS_H
#include <stddef.h>
#include <stdint.h>

typedef int    PetscErrorCode;
typedef int    PetscInt;
typedef int    PetscMPIInt;
typedef double PetscReal;
typedef double PetscScalar;
typedef int    PetscClassId;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef enum { PETSC_MEMTYPE_HOST = 0, PETSC_MEMTYPE_DEVICE = 1, PETSC_MEMTYPE_HIP = 5 } PetscMemType;
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;

typedef struct _p_PetscOptions *PetscOptions;
typedef struct _p_PetscObject {
  PetscOptions options;
  char        *prefix;
} *PetscObject;
typedef struct _p_Vec          *Vec;
typedef struct _p_Mat          *Mat;
typedef struct _p_KSP          *KSP;
typedef struct _p_DM           *DM;
typedef struct _p_IS           *IS;
typedef struct _p_MatNullSpace *MatNullSpace;
/* what petsc/private/pcimpl.h (included by abfpc.c) lets the reference touch */
typedef struct _p_PC *PC;
struct _PCOps {
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*apply)(PC, Vec, Vec);
};
struct _p_PC {
  struct _p_PetscObject hdr;
  struct _PCOps         ops[1];
  Mat                   mat, pmat;
  void                 *data;
};

typedef enum { KSP_NORM_DEFAULT = -1, KSP_NORM_NONE = 0, KSP_NORM_PRECONDITIONED = 1, KSP_NORM_UNPRECONDITIONED = 2, KSP_NORM_NATURAL = 3 } KSPNormType;
typedef enum { MAT_INITIAL_MATRIX, MAT_REUSE_MATRIX, MAT_IGNORE_MATRIX, MAT_INPLACE_MATRIX } MatReuse;
typedef enum { DMSTAG_NULL_LOCATION, DMSTAG_BACK_DOWN_LEFT, DMSTAG_BACK_DOWN, DMSTAG_BACK_DOWN_RIGHT, DMSTAG_BACK_LEFT, DMSTAG_BACK, DMSTAG_BACK_RIGHT, DMSTAG_BACK_UP_LEFT, DMSTAG_BACK_UP, DMSTAG_BACK_UP_RIGHT, DMSTAG_DOWN_LEFT, DMSTAG_DOWN, DMSTAG_DOWN_RIGHT, DMSTAG_LEFT, DMSTAG_ELEMENT, DMSTAG_RIGHT } DMStagStencilLocation;

#define PETSC_SUCCESS           0
#define PETSC_ERR_SUP           56
#define PETSC_ERR_ARG_SIZ       60
#define PETSC_ERR_ARG_WRONG     62
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_NOT_CONVERGED 91
#define PETSC_MIN_REAL          (-1.7976931348623157e308)
#define PC_CLASSID              1
#define MPI_IN_PLACE            ((void *)1)
#define MPI_BYTE                ((MPI_Datatype)1)
#define MPIU_REAL               ((MPI_Datatype)2)
#define MPIU_MAX                ((MPI_Op)1)
#define KSPBCGS                 "bcgs"
#define KSPCHEBYSHEV            "chebyshev"
#define PCNONE                  "none"
#define PCMG                    "mg"
#define MATNEST                 "nest"

PetscErrorCode PetscErrorDecl_(MPI_Comm, PetscErrorCode, const char *, ...) __attribute__((format(printf, 3, 4)));
#define PetscFunctionBegin      do { } while (0)
#define PetscFunctionReturn(x)  return x
#define PetscCall(...)          do { PetscErrorCode ierr_q_ = (__VA_ARGS__); if (ierr_q_) return ierr_q_; } while (0)
#define PetscCallMPI(...)       do { int ierr_m_ = (__VA_ARGS__); if (ierr_m_) return (PetscErrorCode)ierr_m_; } while (0)
#define PetscCheck(cond, comm, ierr, ...) do { if (!(cond)) return PetscErrorDecl_(comm, ierr, __VA_ARGS__); } while (0)
#define PetscValidHeaderSpecific(h, ck, arg) do { (void)(h); } while (0)
#define PetscRealPart(a)        (a)
#define PetscMemTypeDevice(m)   (((m) & 0x1) == 0x1)
PetscErrorCode PetscMallocDecl_(size_t, void *);
PetscErrorCode PetscFreeDecl_(void *);
#define PetscNew(b)             PetscMallocDecl_(sizeof(**(b)), (b))
#define PetscFree(a)            PetscFreeDecl_(a)
#define PetscMalloc2(m1, r1, m2, r2) (PetscMallocDecl_((size_t)(m1) * sizeof(**(r1)), (r1)) || PetscMallocDecl_((size_t)(m2) * sizeof(**(r2)), (r2)))
#define PetscFree2(a, b)        (PetscFreeDecl_(a) || PetscFreeDecl_(b))
PetscErrorCode PetscObjectQueryFunctionDecl_(PetscObject, const char[], void (**)(void));
#define PetscTryMethod(obj, A, B, C) \
  do { \
    PetscErrorCode(*f_tm_) B = NULL; \
    PetscCall(PetscObjectQueryFunctionDecl_((PetscObject)(obj), A, (void (**)(void))&f_tm_)); \
    if (f_tm_) PetscCall((*f_tm_)C); \
  } while (0)

MPI_Comm       PetscObjectComm(PetscObject);
PetscErrorCode PetscObjectGetComm(PetscObject, MPI_Comm *);
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char[], PetscBool *);
PetscErrorCode PetscOptionsGetBool(PetscOptions, const char[], const char[], PetscBool *, PetscBool *);
int            MPI_Comm_rank(MPI_Comm, int *);
int            MPI_Comm_size(MPI_Comm, int *);
int            MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm);
int            MPIU_Allreduce(const void *, void *, PetscMPIInt, MPI_Datatype, MPI_Op, MPI_Comm);

PetscErrorCode VecGetSubVector(Vec, IS, Vec *);
PetscErrorCode VecRestoreSubVector(Vec, IS, Vec *);
PetscErrorCode VecGetLocalSize(Vec, PetscInt *);
PetscErrorCode VecGetDM(Vec, DM *);
PetscErrorCode VecDestroy(Vec *);
PetscErrorCode VecAYPX(Vec, PetscScalar, Vec);
PetscErrorCode VecGetArrayReadAndMemType(Vec, const PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayReadAndMemType(Vec, const PetscScalar **);
PetscErrorCode VecGetArrayWriteAndMemType(Vec, PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayWriteAndMemType(Vec, PetscScalar **);

PetscErrorCode MatNestGetSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatNestGetISs(Mat, IS[], IS[]);
PetscErrorCode MatCreateSubMatrix(Mat, IS, IS, MatReuse, Mat *);
PetscErrorCode MatDestroy(Mat *);
PetscErrorCode MatCreateVecs(Mat, Vec *, Vec *);
PetscErrorCode MatMult(Mat, Vec, Vec);
PetscErrorCode MatGetNullSpace(Mat, MatNullSpace *);
PetscErrorCode MatNullSpaceCreate(MPI_Comm, PetscBool, PetscInt, const Vec[], MatNullSpace *);
PetscErrorCode MatNullSpaceDestroy(MatNullSpace *);

PetscErrorCode KSPSolve(KSP, Vec, Vec);
PetscErrorCode KSPSetOperators(KSP, Mat, Mat);
PetscErrorCode KSPGetTolerances(KSP, PetscReal *, PetscReal *, PetscReal *, PetscInt *);
PetscErrorCode KSPGetPC(KSP, PC *);
PetscErrorCode KSPGetNormType(KSP, KSPNormType *);
PetscErrorCode KSPGetErrorIfNotConverged(KSP, PetscBool *);

PetscErrorCode DMStagGetGlobalSizes(DM, PetscInt *, PetscInt *, PetscInt *);
PetscErrorCode DMStagGetNumRanks(DM, PetscInt *, PetscInt *, PetscInt *);
PetscErrorCode DMStagGetCorners(DM, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *, PetscInt *);
PetscErrorCode DMStagGetProductCoordinateArraysRead(DM, void *, void *, void *);
PetscErrorCode DMStagRestoreProductCoordinateArraysRead(DM, void *, void *, void *);
PetscErrorCode DMStagGetProductCoordinateLocationSlot(DM, DMStagStencilLocation, PetscInt *);
PetscErrorCode DMStagGetDOF(DM, PetscInt *, PetscInt *, PetscInt *, PetscInt *);
#endif
