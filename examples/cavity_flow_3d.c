/*
 * cavity_flow_3d.c -- the reference's 3-D lid-driven cavity (fluca/tests/cavity_flow/cavity_flow_3d.c) against the C host
 * mirror (include/fluca_host.h): the same call sequence -- mesh, NS, six boundary conditions, NSSetFromOptions, NSSetUp,
 * zero initial solution, NSSolve -- with every operator of the time step on the GPU and no PETSc, no Python.
 *
 *   gcc -O2 examples/cavity_flow_3d.c -Iinclude -Lfluca_amd/lib -lfluca_host -lflucahip -lm -Wl,-rpath,$PWD/fluca_amd/lib
 *   ./a.out -cart_grid_x 64 -cart_grid_y 64 -cart_grid_z 32 -ns_time_step_size 5e-3 -ns_max_steps 20 -Re 100
 *
 * Difference from the reference run: kspA is BiCGStab + Jacobi instead of PETSc's default GMRES + ILU;
 * the outer solver is GMRES with PC_ABF as in the reference (-ns_ksp_type richardson|preonly are the alternatives).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fluca_host.h"
#ifdef FLUCA_HAVE_CGNS
#include "fluca_cgns.h"
#endif

#define CHK(call)                                                                 \
  do {                                                                            \
    FlErrorCode e_ = (call);                                                      \
    if (e_) {                                                                     \
      fprintf(stderr, "%s:%d: %s -> error %d\n", __FILE__, __LINE__, #call, e_); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)
#define ABI(call) CHK(-(call))

static FlErrorCode wall_velocity(int dim, double t, const double x[], double val[], void *ctx)
{
  (void)dim; (void)t; (void)x; (void)ctx;
  val[0] = val[1] = val[2] = 0.;
  return 0;
}
static FlErrorCode moving_wall_velocity(int dim, double t, const double x[], double val[], void *ctx)
{
  (void)dim; (void)t; (void)x; (void)ctx;
  val[0] = 1.;
  val[1] = val[2] = 0.;
  return 0;
}

int main(int argc, char **argv)
{
  Mesh   mesh;
  NS     ns;
  double Re = 100., rho = 1., mu;
  for (int a = 1; a + 1 < argc; ++a)
    if (!strcmp(argv[a], "-Re")) Re = atof(argv[a + 1]);
  mu = 1. / Re;

  CHK(MeshCartCreate3d(MESHCART_BOUNDARY_NONE, MESHCART_BOUNDARY_NONE, MESHCART_BOUNDARY_NONE, 64, 64, 32, FL_DECIDE, FL_DECIDE, FL_DECIDE, NULL, NULL, NULL, &mesh));
  CHK(MeshSetFromOptions(mesh, argc, argv));
  CHK(MeshSetUp(mesh));
  CHK(MeshCartSetUniformCoordinates(mesh, 0., 1., 0., 1., 0., 0.5));

  CHK(NSCreate(&ns));
  CHK(NSSetType(ns, NSCNLINEAR));
  CHK(NSSetMesh(ns, mesh));
  CHK(NSSetDensity(ns, rho));
  CHK(NSSetViscosity(ns, mu));
  {
    NSBoundaryCondition wallbc = {.type = NS_BC_VELOCITY, .velocity = wall_velocity}, movingwallbc = {.type = NS_BC_VELOCITY, .velocity = moving_wall_velocity},
                        symbc = {.type = NS_BC_SYMMETRY};
    int il, ir, id, iu, ib, ifr;
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_LEFT, &il));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_RIGHT, &ir));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_DOWN, &id));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_UP, &iu));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_BACK, &ib));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_FRONT, &ifr));
    CHK(NSSetBoundaryCondition(ns, il, wallbc));
    CHK(NSSetBoundaryCondition(ns, ir, wallbc));
    CHK(NSSetBoundaryCondition(ns, id, wallbc));
    CHK(NSSetBoundaryCondition(ns, iu, movingwallbc));
    CHK(NSSetBoundaryCondition(ns, ib, symbc));
    CHK(NSSetBoundaryCondition(ns, ifr, wallbc));
  }
  CHK(NSSetTimeStepSize(ns, 5e-3));
  CHK(NSSetMaxSteps(ns, 10));
  CHK(NSSetFromOptions(ns, argc, argv));
  CHK(NSSetUp(ns));
  /* NSGetSolution + VecSet(sol, 0): the mirror's solution arrays start zeroed */

  int64_t sz[4], M, N, P, maxsteps = 0, step = 0;
  CHK(NSGetLocalSizes(ns, sz));
  CHK(MeshCartGetGlobalSizes(mesh, &M, &N, &P));
  for (int a = 1; a + 1 < argc; ++a)
    if (!strcmp(argv[a], "-ns_max_steps")) maxsteps = atoll(argv[a + 1]);
  if (maxsteps <= 0) maxsteps = 10;
  double *v_dev, *V_dev[3], *p_dev;
  CHK(NSGetSolutionArrays(ns, &v_dev, V_dev, &p_dev));
  double *v = (double *)malloc(sizeof(double) * 3 * (size_t)sz[0]);
  if (!v) return 1;
#ifdef FLUCA_HAVE_CGNS
  /* -ns_monitor_solution cgns:<template with %d> [-ns_monitor_solution_interval n] [-viewer_cgns_batch_size n] (nsmon.c:47-100)
   * and -ns_view_solution cgns:<file> after the last step (nsbasic.c:349) */
  FlucaViewerCGNS  monviewer = NULL, endviewer = NULL;
  FlucaCGNSMonitor mon = {NULL, 1};
  int              batch = 1;
  for (int a = 1; a + 1 < argc; ++a) {
    if (!strcmp(argv[a], "-ns_monitor_solution") && !strncmp(argv[a + 1], "cgns:", 5)) CHK(FlucaViewerCGNSOpen(argv[a + 1] + 5, 'w', &monviewer));
    if (!strcmp(argv[a], "-ns_view_solution") && !strncmp(argv[a + 1], "cgns:", 5)) CHK(FlucaViewerCGNSOpen(argv[a + 1] + 5, 'w', &endviewer));
    if (!strcmp(argv[a], "-ns_monitor_solution_interval")) mon.view_interval = atoi(argv[a + 1]);
    if (!strcmp(argv[a], "-viewer_cgns_batch_size")) batch = atoi(argv[a + 1]);
  }
  if (monviewer) {
    CHK(FlucaViewerCGNSSetBatchSize(monviewer, batch));
    mon.viewer = monviewer;
    CHK(NSMonitorSet(ns, NSMonitorSolutionCGNS, &mon, NULL));
  }
#endif
  /* NSSolve, one step at a time so that a monitor line can be printed (-ns_monitor of the reference) */
  while (step < maxsteps) {
    CHK(NSMonitor(ns));
    int    its, reason;
    double rnorm, t;
    CHK(NSStep(ns));
    CHK(NSGetTimeStep(ns, &step));
    CHK(NSGetTime(ns, &t));
    CHK(NSGetLinearSolveInfo(ns, &its, &rnorm, &reason));
    if (reason < 0) {
      fprintf(stderr, "step %lld failed\n", (long long)step);
      return 2;
    }
    ABI(fl_memcpy_d2h(0, v, v_dev, sizeof(double) * 3 * (size_t)sz[0]));
    double ke = 0., umax = 0.;
    for (int64_t q = 0; q < 3 * sz[0]; ++q) {
      ke += 0.5 * v[q] * v[q];
      if (fabs(v[q]) > umax) umax = fabs(v[q]);
    }
    ke *= (1. / M) * (1. / N) * (0.5 / P);
    printf("%lld NS time %g  outer its %d  residual %.3e  kinetic energy %.8e  max |v| %.6f\n", (long long)step, t, its, rnorm, ke, umax);
  }
  /* u on the vertical centre line of the symmetry plane (k = 0): the classic cavity profile */
  printf("u(x=0.5, y, z~0):");
  for (int64_t j = 0; j < N; j += (N >= 16 ? N / 16 : 1)) printf(" %.4f", 0.5 * (v[(0 * N + j) * M + M / 2 - 1] + v[(0 * N + j) * M + M / 2]));
  printf("\n");
  free(v);
  CHK(NSMonitor(ns));
#ifdef FLUCA_HAVE_CGNS
  if (endviewer) CHK(NSViewSolution(ns, endviewer));
  CHK(FlucaViewerCGNSDestroy(&endviewer));
  CHK(FlucaViewerCGNSDestroy(&monviewer));
#endif
  CHK(MeshDestroy(&mesh));
  CHK(NSDestroy(&ns));
  return 0;
}
