// fl_device.h -- small device-side helpers shared by the Krylov kernels (fl_ksp.hip, fl_momentum.hip).
#pragma once
#include "fl_internal.h"

namespace fl {

__device__ __forceinline__ int64_t pidx(const GridP &g, int i, int j, int k) { return g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i; }
__device__ __forceinline__ double  wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < NV; ++a) {
    v[a] = wave_sum(v[a]);
    if (lane == 0) red[a * 4 + w] = v[a];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int a = 0; a < NV; ++a) v[a] = (red[a * 4 + 0] + red[a * 4 + 1]) + (red[a * 4 + 2] + red[a * 4 + 3]);
  }
}
__device__ __forceinline__ void reduce_partials(const double *__restrict__ partial, int nblocks, int stride, int nslot, double *out, double *red)
{
  double v[NSLOT];
#pragma unroll
  for (int a = 0; a < NSLOT; ++a) {
    v[a] = 0.;
    if (a < nslot)
      for (int b = threadIdx.x; b < nblocks; b += 256) v[a] += partial[(int64_t)a * stride + b];
  }
  block_sum<NSLOT>(v, red);
  if (threadIdx.x == 0)
    for (int a = 0; a < NSLOT; ++a) out[a] = v[a];
  __syncthreads();
}
__device__ __forceinline__ int converged_default(const KspScal *s, double dp)
{
  if (isnan(dp) || isinf(dp)) return FL_DIVERGED_NANORINF;
  if (dp <= s->ttol) return dp < s->atol ? FL_CONVERGED_ATOL : FL_CONVERGED_RTOL;
  if (dp >= s->dtol * s->rnorm0) return FL_DIVERGED_DTOL;
  return 0;
}

// Chebyshev coefficient recurrence (KSPCHEBYSHEV, oracle/fluca_oracle.c):  c_{k+1} = 2 mu c_k - c_{k-1};
// omega = omegaprod c_k / c_{k+1};  d' = (omega - 1) d + omega scale z.  From the state (ck, ckm1) left behind by the
// previous step to the state and the (rho, c) of the next one.
__device__ __forceinline__ void cheb_advance(const KspScal *s, double ck, double ckm1, double &ck_out, double &ckm1_out, double &rho, double &c)
{
  const double ckp1  = 2. * s->mu * ck - ckm1;
  const double omega = s->omegaprod * ck / ckp1;
  ckm1_out           = ck;
  ck_out             = ckp1;
  rho                = omega - 1.;
  c                  = omega * s->scale;
}

}  // namespace fl
