/*
 * cavity_pressure_step.c -- the reference's 3-D lid-driven-cavity driver (fluca/tests/cavity_flow/cavity_flow_3d.c)
 * re-written against the C host mirror (include/fluca_host.h): same call sequence, no PETSc, no Python.  This one drives
 * only the pressure half of PCApply_ABF (examples/cavity_flow_3d.c runs whole time steps): it feeds a synthetic
 * intermediate velocity V* (the staggered gradient
 * of a smooth field, so that the exact pressure correction is known) through the pressure half of PCApply_ABF and prints
 * what -ns_abf_schur_ksp_monitor-style output would show.
 *
 *   cc -O2 examples/cavity_pressure_step.c -Iinclude -Lfluca_amd/lib -lfluca_host -lflucahip -lm -Wl,-rpath,$PWD/fluca_amd/lib
 *   ./a.out -cart_grid_x 128 -cart_grid_y 128 -cart_grid_z 64 -ns_time_step_size 1e-3 -ns_abf_schur_ksp_rtol 1e-8
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fluca_host.h"

#define CHK(call)                                                          \
  do {                                                                     \
    FlErrorCode e_ = (call);                                               \
    if (e_) {                                                              \
      fprintf(stderr, "%s:%d: %s -> error %d\n", __FILE__, __LINE__, #call, e_); \
      return 1;                                                            \
    }                                                                      \
  } while (0)
#define ABI(call) CHK(-(call))

int main(int argc, char **argv)
{
  Mesh mesh;
  NS   ns;
  const double Re = 100., rho = 1., mu = 1. / Re;
  const double PI = 3.14159265358979323846;

  CHK(MeshCartCreate3d(MESHCART_BOUNDARY_NONE, MESHCART_BOUNDARY_NONE, MESHCART_BOUNDARY_NONE, 64, 64, 32, FL_DECIDE, FL_DECIDE, FL_DECIDE, NULL, NULL, NULL, &mesh));
  CHK(MeshSetFromOptions(mesh, argc, argv));
  CHK(MeshSetUp(mesh));
  CHK(MeshCartSetUniformCoordinates(mesh, 0., 1., 0., 1., 0., 0.5));

  CHK(NSCreate(&ns));
  CHK(NSSetType(ns, NSCNLINEAR));
  CHK(NSSetMesh(ns, mesh));
  CHK(NSSetDensity(ns, rho));
  CHK(NSSetViscosity(ns, mu));
  CHK(NSSetTimeStepSize(ns, 1e-3));
  {
    NSBoundaryCondition wallbc = {.type = NS_BC_VELOCITY}, symbc = {.type = NS_BC_SYMMETRY};
    int                 il, ir, id, iu, ib, ifr;
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_LEFT, &il));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_RIGHT, &ir));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_DOWN, &id));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_UP, &iu));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_BACK, &ib));
    CHK(MeshCartGetBoundaryIndex(mesh, MESHCART_FRONT, &ifr));
    CHK(NSSetBoundaryCondition(ns, il, wallbc));
    CHK(NSSetBoundaryCondition(ns, ir, wallbc));
    CHK(NSSetBoundaryCondition(ns, id, wallbc));
    CHK(NSSetBoundaryCondition(ns, iu, wallbc)); /* the lid only enters through the momentum right-hand side */
    CHK(NSSetBoundaryCondition(ns, ib, symbc));
    CHK(NSSetBoundaryCondition(ns, ifr, wallbc));
  }
  CHK(NSSetFromOptions(ns, argc, argv));
  CHK(NSSetUp(ns));

  int64_t M, N, P, sz[4];
  CHK(MeshCartGetGlobalSizes(mesh, &M, &N, &P));
  CHK(NSGetLocalSizes(ns, sz));
  /* q = cos(pi x) cos(pi y) cos(2 pi z) (mean-free on the cell centres); V* := kappa Gst q computed by the library itself:
     start from V = 0 and project with p = -q  ->  V = 0 - kappa Gst (-q) */
  double *q = (double *)malloc(sizeof(double) * (size_t)sz[0]);
  double  mean = 0.;
  for (int64_t k = 0; k < P; ++k)
    for (int64_t j = 0; j < N; ++j)
      for (int64_t i = 0; i < M; ++i) {
        const double x = (i + 0.5) / M, y = (j + 0.5) / N, z = 0.5 * (k + 0.5) / P;
        q[(k * N + j) * M + i] = cos(PI * x) * cos(PI * y) * cos(2. * PI * z);
        mean += q[(k * N + j) * M + i];
      }
  mean /= (double)sz[0];
  for (int64_t c = 0; c < sz[0]; ++c) q[c] = -(q[c] - mean);
  void       *d_q, *d_dp, *d_V[3], *d_p0, *d_phalf, *d_p;
  fl_poisson *h;
  CHK(NSGetPoisson(ns, &h));
  ABI(fl_malloc(0, sizeof(double) * sz[0], &d_q));
  ABI(fl_malloc(0, sizeof(double) * sz[0], &d_dp));
  ABI(fl_malloc(0, sizeof(double) * sz[0], &d_p0));
  ABI(fl_malloc(0, sizeof(double) * sz[0], &d_phalf));
  ABI(fl_malloc(0, sizeof(double) * sz[0], &d_p));
  for (int a = 0; a < 3; ++a) ABI(fl_malloc(0, sizeof(double) * sz[1 + a], &d_V[a]));
  ABI(fl_memcpy_h2d(0, d_q, q, sizeof(double) * sz[0]));
  ABI(fl_poisson_project(h, (const double *)d_q, NULL, NULL, NULL, (double *)d_V[0], (double *)d_V[1], (double *)d_V[2]));

  fl_ksp_stats st;
  double      *Vs[3] = {(double *)d_V[0], (double *)d_V[1], (double *)d_V[2]};
  CHK(NSPressureCorrection(ns, NULL, Vs, NULL, (double *)d_dp, &st));
  CHK(NSUpdatePressure(ns, (const double *)d_dp, (const double *)d_p0, (double *)d_phalf, (double *)d_p));

  double *dp = (double *)malloc(sizeof(double) * (size_t)sz[0]);
  ABI(fl_memcpy_d2h(0, dp, d_dp, sizeof(double) * sz[0]));
  double dmean = 0., err = 0.;
  for (int64_t c = 0; c < sz[0]; ++c) dmean += dp[c];
  dmean /= (double)sz[0];
  for (int64_t c = 0; c < sz[0]; ++c) err = fmax(err, fabs((dp[c] - dmean) - (-q[c])));
  int64_t step;
  double  t;
  CHK(NSGetTimeStep(ns, &step));
  CHK(NSGetTime(ns, &t));
  printf("%lld NS dt %g time %g\n", (long long)step, 1e-3, t); /* NSMonitorDefault format, nsmon.c:71-88 */
  printf("grid %lldx%lldx%lld  KSP(schur) its %d reason %d  |r|/|r0| %.3e  max|dp - dp_exact| %.3e  %.3f ms\n", (long long)M, (long long)N, (long long)P, st.iters, st.reason, st.rnorm / st.rnorm0, err,
         st.seconds * 1e3);
  const int ok = st.reason > 0 && err < 1e-3;
  CHK(NSDestroy(&ns));
  CHK(MeshDestroy(&mesh));
  free(q);
  free(dp);
  return ok ? 0 : 2;
}
