/* petsc_cg_driver.c -- the "true reference" timing of SURVEY 8(d): the SAME Schur complement S = -kappa D Gst of the
 * lid-driven cavity (uniform grid on [0,1] x [0,1] x [0,0.5], VELOCITY / SYMMETRY walls = homogeneous Neumann everywhere,
 * abfpc.c:150-177) assembled as AIJ through a DMDA, solved by KSPCG + PCJACOBI with the constant null space attached, for a
 * fixed number of iterations -- what KSPSolve(kspS) of the reference spends its time on with
 * -ns_abf_schur_ksp_type cg -ns_abf_schur_pc_type jacobi.
 *
 * TEST INFRASTRUCTURE, like everything under oracle/.  bench.py's cpu_baseline leg compiles and runs it ONLY where a PETSc
 * installation is found (PETSC_DIR or pkg-config); this image has none, so the file has NOT been compiled here: it is
 * written against the documented PETSc (>= 3.19, PetscCall) API, and bench.py reports a build or run failure as
 * "unavailable" with the reason instead of a number.
 *
 *   mpiexec -n <cores> ./petsc_cg_driver -n 256 -its 500 -kappa 1e-3
 *   prints:  FLUCA_PETSC n=<n> ranks=<r> its=<k> seconds=<t> rnorm=<||z||>
 */
static char help[] = "Jacobi-PCG on the cavity-flow pressure Schur complement (timing driver).\n";
#include <petscdmda.h>
#include <petscksp.h>
#include <petsctime.h>

int main(int argc, char **argv)
{
  DM           da;
  Mat          S;
  Vec          b, x, p;
  KSP          ksp;
  PC           pc;
  MatNullSpace nsp;
  PetscInt     n = 256, its = 500, xs, ys, zs, xm, ym, zm, done;
  PetscReal    kappa = 1e-3, rnorm;
  PetscMPIInt  size;
  PetscLogDouble t0, t1;
  PetscRandom  rnd;

  PetscFunctionBeginUser;
  PetscCall(PetscInitialize(&argc, &argv, NULL, help));
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-n", &n, NULL));
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-its", &its, NULL));
  PetscCall(PetscOptionsGetReal(NULL, NULL, "-kappa", &kappa, NULL));
  PetscCallMPI(MPI_Comm_size(PETSC_COMM_WORLD, &size));
  PetscCall(DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_STAR, n, n, n, PETSC_DECIDE, PETSC_DECIDE, PETSC_DECIDE, 1, 1, NULL,
                         NULL, NULL, &da));
  PetscCall(DMSetFromOptions(da));
  PetscCall(DMSetUp(da));
  PetscCall(DMCreateMatrix(da, &S));
  PetscCall(DMCreateGlobalVector(da, &b));
  PetscCall(VecDuplicate(b, &x));
  PetscCall(VecDuplicate(b, &p));
  {
    /* row of cell (i,j,k): off-diagonals -kappa / h_d^2 towards every neighbour that exists, diagonal = minus their sum
     * (a VELOCITY / SYMMETRY wall removes the term from both: empty Gst row, cnlinearcart3d.c:2449-2452) */
    const PetscReal h[3] = {1.0 / n, 1.0 / n, 0.5 / n};
    PetscCall(DMDAGetCorners(da, &xs, &ys, &zs, &xm, &ym, &zm));
    for (PetscInt k = zs; k < zs + zm; ++k)
      for (PetscInt j = ys; j < ys + ym; ++j)
        for (PetscInt i = xs; i < xs + xm; ++i) {
          MatStencil  row = {.i = i, .j = j, .k = k, .c = 0}, col[7];
          PetscScalar v[7], diag = 0.;
          PetscInt    cnt = 0;
          const PetscInt idx[3] = {i, j, k};
          for (PetscInt d = 0; d < 3; ++d)
            for (PetscInt s = -1; s <= 1; s += 2) {
              const PetscInt q = idx[d] + s;
              if (q < 0 || q >= n) continue;
              col[cnt]   = row;
              if (d == 0) col[cnt].i = q;
              if (d == 1) col[cnt].j = q;
              if (d == 2) col[cnt].k = q;
              v[cnt] = -kappa / (h[d] * h[d]);
              diag -= v[cnt];
              ++cnt;
            }
          col[cnt] = row;
          v[cnt++] = diag;
          PetscCall(MatSetValuesStencil(S, 1, &row, cnt, col, v, INSERT_VALUES));
        }
    PetscCall(MatAssemblyBegin(S, MAT_FINAL_ASSEMBLY));
    PetscCall(MatAssemblyEnd(S, MAT_FINAL_ASSEMBLY));
  }
  PetscCall(MatNullSpaceCreate(PETSC_COMM_WORLD, PETSC_TRUE, 0, NULL, &nsp)); /* abfpc.c:173-177 */
  PetscCall(MatSetNullSpace(S, nsp));
  /* b = S p*, p* uniform(-1,1) made mean-free (SURVEY 8d micro-benchmark right-hand side) */
  PetscCall(PetscRandomCreate(PETSC_COMM_WORLD, &rnd));
  PetscCall(PetscRandomSetInterval(rnd, -1.0, 1.0));
  PetscCall(PetscRandomSetSeed(rnd, 20260313));
  PetscCall(PetscRandomSeed(rnd));
  PetscCall(VecSetRandom(p, rnd));
  PetscCall(MatNullSpaceRemove(nsp, p));
  PetscCall(MatMult(S, p, b));
  PetscCall(KSPCreate(PETSC_COMM_WORLD, &ksp));
  PetscCall(KSPSetOperators(ksp, S, S));
  PetscCall(KSPSetType(ksp, KSPCG));
  PetscCall(KSPGetPC(ksp, &pc));
  PetscCall(PCSetType(pc, PCJACOBI));
  PetscCall(KSPSetTolerances(ksp, PETSC_DEFAULT, PETSC_DEFAULT, PETSC_DEFAULT, its));
  PetscCall(KSPSetConvergenceTest(ksp, KSPConvergedSkip, NULL, NULL)); /* exactly `its` iterations */
  PetscCall(KSPSetFromOptions(ksp));
  PetscCall(KSPSetUp(ksp));
  PetscCall(PetscTime(&t0));
  PetscCall(KSPSolve(ksp, b, x));
  PetscCall(PetscTime(&t1));
  PetscCall(KSPGetIterationNumber(ksp, &done));
  PetscCall(KSPGetResidualNorm(ksp, &rnorm));
  PetscCall(PetscPrintf(PETSC_COMM_WORLD, "FLUCA_PETSC n=%" PetscInt_FMT " ranks=%d its=%" PetscInt_FMT " seconds=%.6f rnorm=%.6e\n", n, (int)size, done, (double)(t1 - t0), (double)rnorm));
  PetscCall(PetscRandomDestroy(&rnd));
  PetscCall(MatNullSpaceDestroy(&nsp));
  PetscCall(KSPDestroy(&ksp));
  PetscCall(VecDestroy(&p));
  PetscCall(VecDestroy(&x));
  PetscCall(VecDestroy(&b));
  PetscCall(MatDestroy(&S));
  PetscCall(DMDestroy(&da));
  PetscCall(PetscFinalize());
  return 0;
}
