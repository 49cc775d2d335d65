// fl_internal.h -- internal structures of libflucahip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/fluca_hip.h"

#define FL_HIP(call)                                                                                          \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      std::fprintf(stderr, "[flucahip] %s:%d %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return FL_ERR_GPU;                                                                                      \
    }                                                                                                         \
  } while (0)

#define FL_CHK(call)          \
  do {                        \
    int rc_ = (call);         \
    if (rc_ != 0) return rc_; \
  } while (0)

namespace fl {

constexpr int PADX = 16;  // doubles in front of cell i=0 of every padded row (ghost i=-1 is the last of them): 128-B aligned rows

// ---- host description of one axis of the GLOBAL grid -------------------------------------------------------------
struct Axis {
  int64_t             n = 0;
  bool                periodic = false;
  int                 bc_lo = 0, bc_hi = 0;
  std::vector<double> xf;  // n+1
  std::vector<double> xc;  // n+2, xc[1+i], i = -1..n
  // staggered gradient rows, one per face f = 0..n:  g_f = ga0 * p[gc0] + ga1 * p[gc0+1]   (unscaled; wall rows are 0)
  std::vector<double> ga0, ga1;
  std::vector<int64_t> gc0;
  // divergence: 1/dx_i
  std::vector<double> idx;
  // Schur rows (already times kappa):  (S p)_i = sl p[i-1] + sc p[i] + sh p[i+1]   summed over the three axes
  std::vector<double> sl, sc, sh;
  // cell-centred gradient rows (unscaled): (G p)_i = sum_{c<3} Gv[c] p[Gs + c]
  std::vector<int64_t> Gs;
  std::vector<double>  Gv0, Gv1, Gv2;
  double               bcc_lo = 0., bcc_hi = 0.;  // coefficient of the outlet pressure in the Gst boundary vector
  double xcc(int64_t i) const { return xc[(size_t)(i + 1)]; }
};

int build_axis(Axis &a, int64_t n, const double *xf, const double *xc, int bc_lo, int bc_hi, double kappa);

constexpr int MOM_NTAB = 20;  // 1-D numbers per cell and axis of the momentum operator, see build_axis_momentum
int build_axis_momentum(const Axis &a, std::vector<double> &tab);
int build_axis_faceinterp(const Axis &a, int kind, std::vector<double> &w0, std::vector<double> &w1, std::vector<int> &c0);

// ---- device view shared by every kernel ---------------------------------------------------------------------------
// 1-D coefficient arrays are LOCAL (this rank's block) and pre-shifted: valid for index -1..len.
struct GridP {
  int     nx, ny, nz;  // owned cells of this rank
  int     fx, fy, fz;  // owned faces along each axis
  int     sx;          // padded row stride (doubles), multiple of 16
  int64_t sxy;         // padded plane stride
  int64_t off0;        // offset of cell (0,0,0) in a padded array
  const double *sl[3], *sc[3], *sh[3];  // Schur rows; sc of a wall ghost is +inf (so 1/diag = 0 there)
  const double *idx[3];                 // 1/dx, index 0..len-1
  const double *ga0[3], *ga1[3];        // Gst rows per local face 0..f-1
  const int    *gc0[3];                 // local cell index of the first column (may be -1 = ghost)
  const int    *Gs[3];                  // local start column of the G row
  const double *Gv0[3], *Gv1[3], *Gv2[3];
  double        kappa;
};

// device-side scalar state of a Krylov solve (one per handle)
struct KspScal {
  double rz, rz_old, pq, alpha, beta, zshift, dp, rnorm0, ttol;
  double alpha_old;  // CG: the step length of the iteration before (k_cg_Bq applies two x-updates every second iteration)
  double rtol, atol, dtol;
  double ncell_global;
  // BiCGStab
  double rho, rho_old, omega, omega_old, d1, d2, vshift, tshift, rshift;
  // Chebyshev (lazy constant shifts of x and d)
  double ck, ckm1, mu, omegaprod, scale, xshift, dshift, cheb_rho, cheb_c;
  int    it, maxit, reason, norm_type, nullspace, pending_x, cur;
  int    x_valid;  // CG, q-free pair: 0 until k_cg_Bq has written x for the first time (the padded x is not zeroed: the first pair of
                   // updates writes it without reading it)
  int    dcur;  // Chebyshev: which of the two d buffers holds the current d (the fused two-step kernel flips it)
};

constexpr int MAX_PARTIAL_BLOCKS = 4096;
constexpr int NSLOT              = 8;  // partial-sum slots per kernel

}  // namespace fl
