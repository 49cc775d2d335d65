/*
 * fluca_host_impl.h -- the structs behind the opaque Mesh / NS / FlucaViewer handles of fluca_host.h, for code that
 * REGISTERS a type (NSRegister, MeshRegister) or a viewer: what fluca/include/fluca/private/{nsimpl.h, meshimpl.h,
 * meshcartimpl.h, nslinearcnimpl.h} are to the reference's type implementations.  Applications include fluca_host.h only.
 *
 * As in the reference the ops table is the FIRST member of each object (PETSCHEADER(struct _NSOps) puts `ops` in the
 * object header), so a constructor registered with NSRegister fills ns->ops->{...} exactly like NSCreate_CNLinear
 * (cnlinear.c:164-187); a derived type calls NSCreate_CNLinear first and replaces the slots it overrides.
 */
#ifndef FLUCA_HOST_IMPL_H
#define FLUCA_HOST_IMPL_H

#include <stdarg.h>
#include "fluca_host.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Viewer ------------------------------------------------------------------------------------------------------------
 * The reference's VecView / FlucaVecLoad dispatch on the viewer type inside the Vec's own ops (VecView_Cart, cart.c:226-227);
 * here the viewer carries the table.  Any slot may be NULL (= PETSC_ERR_SUP for that operation). */
struct _FlucaViewerOps {
  FlErrorCode (*vprintf)(FlucaViewer, const char *fmt, va_list ap); /* PetscViewerASCIIPrintf (honours the tab level) */
  FlErrorCode (*viewmesh)(FlucaViewer, Mesh);                       /* MeshView_Cart_CGNS, cartcgns.c:8-118 */
  /* MeshLoad_Cart_CGNS, cartcgns.c:120-158: global sizes and malloc'ed face coordinates (N[d] + 1 each; the caller frees) */
  FlErrorCode (*loadmesh)(FlucaViewer, int64_t N[3], double *xf[3]);
  /* one NSViewSolution / NSLoadSolution: begin, one call per field in the reference's VecView order, end */
  FlErrorCode (*solutionbegin)(FlucaViewer, NS, int write);
  FlErrorCode (*cellfield)(FlucaViewer, NS, const char *name, int ncomp, double *dev);  /* ncomp 3: "<name>X|Y|Z", cartcgns.c:383-386 */
  FlErrorCode (*facefield)(FlucaViewer, NS, const char *name, double *const dev[3]);
  FlErrorCode (*solutionend)(FlucaViewer, NS);
  FlErrorCode (*destroy)(FlucaViewer);
};
struct _p_FlucaViewer {
  struct _FlucaViewerOps ops[1];
  FlucaViewerType        type;
  char                   mode; /* 'w' or 'r' */
  int                    tab;  /* PetscViewerASCIIPushTab level */
  /* the output sequence of the solution last read (Mesh{Set,Get}OutputSequenceNumber, nssol.c:186,197-201) */
  int64_t seqnum;
  double  seqval;
  void   *data;
};
FlErrorCode FlucaViewerCreate(FlucaViewerType type, char mode, FlucaViewer *viewer); /* an empty table; the caller fills ops and data */
FlErrorCode FlucaViewerASCIIPrintf(FlucaViewer viewer, const char *fmt, ...);
FlErrorCode FlucaViewerASCIIPushTab(FlucaViewer viewer);
FlErrorCode FlucaViewerASCIIPopTab(FlucaViewer viewer);

/* ---- Mesh ---- */
typedef struct {
  int64_t              N[3];
  int                  nRanks[3];
  int64_t             *l[3]; /* ownership ranges */
  MeshCartBoundaryType bndTypes[3];
  int64_t              refineFactor[3]; /* -cart_refine_{x,y,z}, default 2 (cart.c:276) */
  double              *xf[3], *xc[3];   /* global coordinates, set by MeshSetUp / SetUniformCoordinates */
  double              *coordLoaded[3];  /* face coordinates MeshLoad read (cartcgns.c:146-150); MeshSetUp installs them (cart.c:131-140) */
} Mesh_Cart; /* = fluca/include/fluca/private/meshcartimpl.h:8-17 */

struct _p_Mesh {
  struct _MeshOps ops[1];
  char            type_name[32];
  int             dim, rank, size, setupcalled;
  fl_decomp       decomp;
  void           *data;
};
FlErrorCode MeshCreate_Cart(Mesh mesh); /* cart.c:262-288 */

/* ---- NS ---- */
#define MAXNSMONITORS 10
struct _p_NS {
  struct _NSOps        ops[1];
  char                 type_name[32];
  double               rho, mu, dt, t;
  int64_t              step, max_steps;
  double               max_time; /* nsbasic.c:30: PETSC_MAX_REAL = not set */
  int                  errorifstepfailed; /* nsbasic.c:46: PETSC_TRUE */
  int                  mom_guess_previous; /* -ns_abf_momentum_guess_previous (1) / -ns_abf_momentum_guess_extrapolate (2); mirror only, default 0: see NSSetFromOptions */
  int                  bc_keep;           /* -ns_keep_boundary_values (mirror only, default 1): a step's callback values at t + dt serve the next step's t */
  Mesh                 mesh;
  NSBoundaryCondition *bcs;
  int                  nb, device, setupcalled;
  fl_poisson          *poisson;  /* plays PC_ABF's kspS + S */
  fl_momentum         *momentum; /* ns->J: plays the MATNEST Jacobian and PC_ABF's kspA + A (created by NSSetUp, wired by formjacobian(NS_INIT_JACOBIAN)) */
  fl_ibm              *ibm;      /* immersed boundary (build-defined direct forcing, NSSetImmersedBoundary) */
  int64_t              ibm_L;
  const double        *ibm_dV, *ibm_Ut;
  double              *ibm_U;
  fl_ksp_opts          schur;    /* -ns_abf_schur_* */
  fl_ksp_opts          mom;      /* -ns_abf_momentum_* */
  int                  schur_ainv, upper_ainv; /* -ns_pc_abf_schur_ainv_type / -ns_pc_abf_upper_ainv_type (PCABFAinvType), default ID */
  int                  ksp_type;           /* -ns_ksp_type: 0 richardson, 1 preonly, 2 gmres (the reference's default, nssol.c:21-29) */
  int                  gmres_restart;      /* -ns_ksp_gmres_restart (PETSc default 30) */
  double               ksp_rtol, ksp_atol; /* -ns_ksp_rtol 1e-5 (nssol.c:24), unpreconditioned norm (nssol.c:25) */
  int                  ksp_max_it;
  int                  ksp_its, reason;    /* of the last step */
  int                  mom_its, schur_its; /* inner Krylov iterations summed over the last step's outer iterations */
  double               ksp_rnorm;
  double               ksp_rnorm0; /* norm of the right-hand side the outer solve started from (the reference norm of its rtol test) */
  /* NSMonitorSet list (nsimpl.h: monitor[], monitorctx[], monitordestroy[], MAXNSMONITORS) */
  int                  nmon;
  FlErrorCode (*mon[MAXNSMONITORS])(NS, void *);
  void *monctx[MAXNSMONITORS];
  FlErrorCode (*mondestroy[MAXNSMONITORS])(void **);
  void                *data;
};

/* NSCNLINEAR's data (nslinearcnimpl.h: v0interp, phalf, B) plus, in this mirror, the solution and solver vectors the
 * reference keeps in NS itself (ns->sol, sol0, x, r): they are device arrays allocated by the type's setup */
typedef struct {
  int64_t sz[4];                               /* cells, x-, y-, z-faces of this rank */
  double *sol_v, *sol_V[3], *sol_p;            /* ns->sol  */
  double *sol0_v, *sol0_V[3], *sol0_p;         /* ns->sol0 */
  double *phalf;                               /* cnl->phalf */
  double *x_v, *x_V[3], *x_p;                  /* ns->x: v, V, dp */
  double *f_v, *f_V[3], *f_p;                  /* ns->r as SNESPicard uses it: the right-hand side (momrhs, interprhs, contrhs) formfunction writes */
  int     have_sol0;                           /* "if (ns->sol0)" of NSFormJacobian (:2930): set by the first NSStep */
  double *W[9];                                /* cnl->v0interp */
  double *r_v, *r_V[3], *r_p, *d_v, *d_V[3], *d_p;
  double *plane_dev, *plane_host[7];           /* page-locked scratch planes, handed out in turn (cnl_scratch) */
  int64_t plane_cap;
  int     plane_next;                          /* planes handed out since the last fence */
  /* values of the velocity callbacks on the boundary planes, kept for the two times a step looks at (t and t + dt: the second is the next step's
   * first): bc_plane[b][slot][component], page-locked; valid for callback bc_fn / context bc_ctx at time bc_time */
  double     *bc_plane[6][2][3];
  double      bc_time[6][2];
  int         bc_have[6][2], bc_old[6];
  void       *bc_ctx[6][2];
  NSBoundaryConditionFunction bc_fn[6][2];
  /* GMRES work vectors, kept from step to step (allocating and freeing ~70 GB of device memory per step at 512^3 left the
   * GPU idle for a third of the step): w, t and the Krylov basis, grown lazily up to restart + 1 */
  NSVec gm_w, gm_t, *gm_V;
  int   gm_nalloc, gm_cap;
} NS_CNLinear;
FlErrorCode NSCreate_CNLinear(NS ns); /* cnlinear.c:164-187 */

#ifdef __cplusplus
}
#endif
#endif
