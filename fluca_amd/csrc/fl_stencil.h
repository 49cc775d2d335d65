// fl_stencil.h -- device helpers shared by the LDS-staged 7-point kernels (k_cg_A / k_cg_Bq in fl_kernels.hip, the BiCGStab
// kernels in fl_ksp.hip): 16-byte accesses, the fixed-order stencil row, the tile shape and the XCD-aware block order.
#pragma once
#include "fl_internal.h"

namespace fl {

typedef double v2d __attribute__((ext_vector_type(2)));
// 16-B accesses with an optional non-temporal hint (streams that are not re-read before they would be evicted anyway)
template <int NT>
__device__ __forceinline__ double2 ld2(const double *p)
{
  if (NT) {
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
    return make_double2(v.x, v.y);
  }
  return *reinterpret_cast<const double2 *>(p);
}
template <int NT>
__device__ __forceinline__ void st2(double *p, double2 v)
{
  if (NT) {
    v2d t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<v2d *>(p));
  } else *reinterpret_cast<double2 *>(p) = v;
}

// One row of S in a fixed rounding order (explicit fma chain): k_cg_A (for p.q), k_cg_Bq (for r - alpha q) and the boundary-layer
// pack of the overlapped halo exchange all form q = S p' with it, so the three agree bit for bit.  dc = xc + (yc + zc).
__device__ __forceinline__ double st7(double dc, double c, double xl, double w, double xh, double e, double yl, double s, double yh, double n, double zl, double b, double zh, double a)
{
  return fma(zh, a, fma(zl, b, fma(yh, n, fma(yl, s, fma(xh, e, fma(xl, w, dc * c))))));
}

template <int RY, int NW>
struct TileA {
  static constexpr int TX = 128, TY = NW * RY, LX = TX + 4, LY = TY + 2;
};

// blockIdx -> logical block.  Blocks are dealt round-robin over the 8 XCDs, so physical blocks b, b+8, ... share an L2.
// Give each XCD a contiguous range of logical blocks (= neighbouring tiles of one z-chunk): the halo rows / columns a
// tile re-reads were just fetched into the same L2 by its neighbour.  Speed only; any mapping is correct.
__device__ __forceinline__ int xcd_remap(int b, int nblocks) { return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3); }

}  // namespace fl
