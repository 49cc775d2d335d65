/*
 * flow_configs.c -- the flow set-ups BASELINE.json lists as whole time steps on one GPU, from C:
 *   -config cavity   N x N x N/2 lid-driven cavity on [0,1]^2 x [0,0.5] (fluca/tests/cavity_flow/cavity_flow_3d.c)
 *   -config channel  N^3 channel on the unit cube: parabolic VELOCITY inlet, PRESSURE_OUTLET p = 0, no-slip walls, periodic span
 *   -config cylinder the channel with an immersed cylinder of diameter 64 h along the periodic span (BASELINE config 5's body on
 *                    one GPU: rings of markers one h apart, N rings -- 102 912 markers at N = 512)
 *   -config sphere   the channel with an immersed sphere of diameter 64 h at the centre (markers on a Fibonacci lattice,
 *                    spacing ~ h: L = 12 868), direct-forcing IBM active every step
 * Prints wall time and solver work per step.  Options of the mirror apply (-ns_time_step_size, -ns_max_steps,
 * -ns_abf_schur_pc_type mg, -ns_ksp_type preonly, ...).
 *
 *   gcc -O2 examples/flow_configs.c -Iinclude -Lfluca_amd/lib -lfluca_host -lflucahip -lm -Wl,-rpath,$PWD/fluca_amd/lib
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fluca_host.h"

#define CHK(call)                                                                 \
  do {                                                                            \
    FlErrorCode e_ = (call);                                                      \
    if (e_) {                                                                     \
      fprintf(stderr, "%s:%d: %s -> error %d\n", __FILE__, __LINE__, #call, e_); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)
#define ABI(call) CHK(-(call))

static FlErrorCode zero_velocity(int dim, double t, const double x[], double val[], void *ctx)
{
  (void)dim; (void)t; (void)x; (void)ctx;
  val[0] = val[1] = val[2] = 0.;
  return 0;
}
static FlErrorCode lid_velocity(int dim, double t, const double x[], double val[], void *ctx)
{
  (void)dim; (void)t; (void)x; (void)ctx;
  val[0] = 1.;
  val[1] = val[2] = 0.;
  return 0;
}
static FlErrorCode inlet_velocity(int dim, double t, const double x[], double val[], void *ctx)
{
  (void)dim; (void)t; (void)ctx;
  val[0] = 4. * x[1] * (1. - x[1]);
  val[1] = val[2] = 0.;
  return 0;
}
static FlErrorCode outlet_pressure(int dim, double t, const double x[], double val[], void *ctx)
{
  (void)dim; (void)t; (void)x; (void)ctx;
  val[0] = 0.;
  return 0;
}

static const char *opt(int argc, char **argv, const char *name, const char *dflt)
{
  for (int a = 1; a + 1 < argc; ++a)
    if (!strcmp(argv[a], name)) return argv[a + 1];
  return dflt;
}
static double now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
  const char *config = opt(argc, argv, "-config", "cavity");
  const int   cavity = !strcmp(config, "cavity"), cylinder = !strcmp(config, "cylinder"), sphere = !strcmp(config, "sphere") || cylinder;
  if (!cavity && !sphere && strcmp(config, "channel")) {
    fprintf(stderr, "unknown -config %s\n", config);
    return 1;
  }
  const int64_t N = atoll(opt(argc, argv, "-n", "128")), steps = atoll(opt(argc, argv, "-ns_max_steps", "3"));
  const double  Re = atof(opt(argc, argv, "-Re", "100")), rho = 1., mu = 1. / Re;
  Mesh          mesh;
  NS            ns;
  const int64_t P3 = cavity ? N / 2 : N;
  CHK(MeshCartCreate3d(MESHCART_BOUNDARY_NONE, MESHCART_BOUNDARY_NONE, cavity ? MESHCART_BOUNDARY_NONE : MESHCART_BOUNDARY_PERIODIC, N, N, P3, FL_DECIDE, FL_DECIDE,
                       FL_DECIDE, NULL, NULL, NULL, &mesh));
  CHK(MeshSetUp(mesh));
  CHK(MeshCartSetUniformCoordinates(mesh, 0., 1., 0., 1., 0., cavity ? 0.5 : 1.));
  CHK(NSCreate(&ns));
  CHK(NSSetType(ns, NSCNLINEAR));
  CHK(NSSetMesh(ns, mesh));
  CHK(NSSetDensity(ns, rho));
  CHK(NSSetViscosity(ns, mu));
  {
    NSBoundaryCondition wall = {.type = NS_BC_VELOCITY, .velocity = zero_velocity}, lid = {.type = NS_BC_VELOCITY, .velocity = lid_velocity},
                        sym = {.type = NS_BC_SYMMETRY}, in = {.type = NS_BC_VELOCITY, .velocity = inlet_velocity},
                        out = {.type = NS_BC_PRESSURE_OUTLET, .pressure = outlet_pressure}, per = {.type = NS_BC_PERIODIC};
    if (cavity) { /* cavity_flow_3d.c:72-77 */
      CHK(NSSetBoundaryCondition(ns, 0, wall)); CHK(NSSetBoundaryCondition(ns, 1, wall)); CHK(NSSetBoundaryCondition(ns, 2, wall));
      CHK(NSSetBoundaryCondition(ns, 3, lid)); CHK(NSSetBoundaryCondition(ns, 4, sym)); CHK(NSSetBoundaryCondition(ns, 5, wall));
    } else {
      CHK(NSSetBoundaryCondition(ns, 0, in)); CHK(NSSetBoundaryCondition(ns, 1, out)); CHK(NSSetBoundaryCondition(ns, 2, wall));
      CHK(NSSetBoundaryCondition(ns, 3, wall)); CHK(NSSetBoundaryCondition(ns, 4, per)); CHK(NSSetBoundaryCondition(ns, 5, per));
    }
  }
  CHK(NSSetTimeStepSize(ns, 0.5 / (double)N)); /* CFL ~ 0.5 on the unit inflow / lid velocity */
  CHK(NSSetMaxSteps(ns, steps));
  CHK(NSSetFromOptions(ns, argc, argv));
  CHK(NSSetUp(ns));
  int64_t sz[4];
  CHK(NSGetLocalSizes(ns, sz));
  double *v_dev, *V_dev[3], *p_dev;
  CHK(NSGetSolutionArrays(ns, &v_dev, V_dev, &p_dev));
  int64_t L = 0;
  if (sphere) {
    /* sphere of diameter 64 h: markers on a Fibonacci lattice, one per h^2 of surface; marker volume h^3 */
    const double h = 1. / (double)N, R = 32. * h, PI = 3.14159265358979323846, ga = PI * (3. - sqrt(5.));
    const int64_t ring = (int64_t)llround(2. * PI * R / h);
    L = cylinder ? ring * P3 : (int64_t)llround(4. * PI * R * R / (h * h));
    double *X = (double *)malloc(sizeof(double) * 4 * (size_t)L);
    if (!X) return 1;
    for (int64_t l = 0; cylinder && l < L; ++l) { /* axis along z through (0.5, 0.5): one ring per cell layer, markers one h apart */
      const double th = 2. * PI * ((double)(l % ring) + 0.5 * (double)((l / ring) & 1)) / (double)ring;
      X[l]         = 0.5 + R * cos(th);
      X[L + l]     = 0.5 + R * sin(th);
      X[2 * L + l] = ((double)(l / ring) + 0.5) * h;
      X[3 * L + l] = h * h * h;
    }
    for (int64_t l = 0; !cylinder && l < L; ++l) {
      const double z = 1. - 2. * ((double)l + 0.5) / (double)L, r = sqrt(1. - z * z), th = ga * (double)l;
      X[l]         = 0.5 + R * r * cos(th);
      X[L + l]     = 0.5 + R * r * sin(th);
      X[2 * L + l] = 0.5 + R * z;
      X[3 * L + l] = h * h * h;
    }
    void *Xd = NULL;
    ABI(fl_malloc(0, sizeof(double) * 4 * (size_t)L, &Xd));
    ABI(fl_memcpy_h2d(0, Xd, X, sizeof(double) * 4 * (size_t)L));
    const double *D = (const double *)Xd;
    CHK(NSSetImmersedBoundary(ns, FL_DELTA_PESKIN4, L, D, D + L, D + 2 * L, D + 3 * L, NULL));
    free(X);
  }
  printf("config %s  cells %lld x %lld x %lld  dt %g  Re %g  markers %lld\n", config, (long long)N, (long long)N, (long long)P3, 0.5 / (double)N, Re, (long long)L);
  double *v = (double *)malloc(sizeof(double) * 3 * (size_t)sz[0]);
  if (!v) return 1;
  for (int64_t s = 0; s < steps; ++s) {
    int    its, reason, mi, si;
    double rnorm, t0 = now();
    CHK(NSStep(ns));
    const double dtw = now() - t0;
    double rnorm0;
    CHK(NSGetLinearSolveInfo(ns, &its, &rnorm, &reason));
    CHK(NSGetLinearSolveResidualNorms(ns, &rnorm0, NULL));
    CHK(NSGetInnerIterations(ns, &mi, &si));
    if (reason < 0) {
      fprintf(stderr, "step %lld failed\n", (long long)(s + 1));
      return 2;
    }
    /* |r| / |f|: the quantity -ns_ksp_rtol bounds (with -ns_ksp_type preonly no residual is formed: one PCApply_ABF per step) */
    printf("step %lld  wall %.3f s  outer its %d  kspA its %d  kspS its %d  |r|/|f| %.2e  (|f| %.2e)\n", (long long)(s + 1), dtw, its, mi, si,
           rnorm0 > 0. ? rnorm / rnorm0 : 0., rnorm0);
    fflush(stdout);
  }
  ABI(fl_memcpy_d2h(0, v, v_dev, sizeof(double) * 3 * (size_t)sz[0]));
  double umax = 0., ke = 0.;
  for (int64_t q = 0; q < 3 * sz[0]; ++q) {
    if (!(fabs(v[q]) <= 1e30)) {
      fprintf(stderr, "non-finite velocity\n");
      return 3;
    }
    if (fabs(v[q]) > umax) umax = fabs(v[q]);
    ke += 0.5 * v[q] * v[q];
  }
  printf("max |v| %.5f  mean kinetic energy %.6e\n", umax, ke / (double)sz[0]);
  if (sphere) { /* what the forcing acts on: the fluid velocity interpolated to the markers (target 0) */
    fl_ibm *ibm;
    void   *Ud = NULL;
    CHK(NSGetImmersedBoundary(ns, &ibm));
    ABI(fl_malloc(0, sizeof(double) * 3 * (size_t)L, &Ud));
    ABI(fl_ibm_interp(ibm, 3, v_dev, (double *)Ud));
    double *U = (double *)malloc(sizeof(double) * 3 * (size_t)L), s2 = 0.;
    if (!U) return 1;
    ABI(fl_memcpy_d2h(0, U, Ud, sizeof(double) * 3 * (size_t)L));
    for (int64_t l = 0; l < L; ++l) s2 += U[l] * U[l] + U[L + l] * U[L + l] + U[2 * L + l] * U[2 * L + l];
    printf("rms fluid speed at the markers %.5f (inflow peak 1)\n", sqrt(s2 / (double)L));
    free(U);
    ABI(fl_free(0, Ud));
  }
  free(v);
  CHK(MeshDestroy(&mesh));
  CHK(NSDestroy(&ns));
  return 0;
}
