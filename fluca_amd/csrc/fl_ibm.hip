// fl_ibm.hip -- immersed-boundary delta-function interpolation / spreading.
//
// There is NO reference implementation (thecasterian/fluca THEORY_GUIDE.md:130-132 is an empty TODO); the specification
// is DESIGN.md "IBM":  U_l = sum_x u(x) delta_h(x - X_l) h^3,  f(x) += sum_l F_l delta_h(x - X_l) dV_l,
// delta_h = prod_d phi(r_d / h_d) / h_d, phi = Peskin 4-point or Roma 3-point, collocated fields (all velocity components live
// at cell centres in Fluca, fluca/src/ns/interface/nsbasic.c:180) so one weight set serves every component.
// Stretched axes (round 2): the delta function lives in INDEX space -- the marker is mapped to the continuous cell-centre index s
// (piecewise linear through the centres; beyond the first / last one through its mirror image in the wall or the periodic image),
// the weights are phi(s - i), interpolation is U = sum u w and spreading divides by the volume of the TARGET cell:
// f_i += F dV w / (dx_i dy_j dz_k).  sum w = 1, <interp u, F dV> = sum_i u_i f_i V_i and sum_i f_i V_i = sum F dV hold on any
// grid; on a uniform one this is exactly the formula above.
//
// Kernels (gfx950, 64-lane wavefronts):
//   k_ibm_weights   one lane per marker: first support cell and the S 1-D weights per axis
//   k_ibm_interp    ONE WAVEFRONT PER MARKER, one lane per support cell (4^3 = 64 lanes exactly), fixed-order wave
//                   reduction -> deterministic
//   k_ibm_spread    gather form, no atomics: one 256-thread block per 8x8x8 tile of cells; the markers whose support
//                   touches the tile (per-tile bins built at create/update) are staged in LDS, rank-sorted by marker id
//                   in LDS so that every cell accumulates in a run-independent order -> bitwise reproducible
#include "fl_handle.h"

namespace fl {

constexpr int TB       = 8;    // tile edge (cells)
constexpr int BIN_CHUNK = 256; // markers staged in LDS per pass

__device__ __forceinline__ double phi_peskin4(double r)
{
  r = fabs(r);
  if (r <= 1.) return (3. - 2. * r + sqrt(1. + 4. * r - 4. * r * r)) / 8.;
  if (r <= 2.) return (5. - 2. * r - sqrt(fmax(-7. + 12. * r - 4. * r * r, 0.))) / 8.;
  return 0.;
}
__device__ __forceinline__ double phi_roma3(double r)
{
  r = fabs(r);
  if (r <= 0.5) return (1. + sqrt(1. - 3. * r * r)) / 3.;
  if (r <= 1.5) return (5. - 3. * r - sqrt(fmax(1. - 3. * (1. - r) * (1. - r), 0.))) / 6.;
  return 0.;
}

struct IbmP {
  int     kind, S;
  int64_t L;
  int     n[3];        // local cells
  int     lo[3];       // global index of local cell 0
  int     ng[3];       // global cells
  int     periodic[3]; // periodic axis of the GLOBAL grid (support indices wrap modulo ng, then ownership is tested)
  double  x0[3], h[3]; // global origin, spacing
  int     nt[3];       // tiles
  int     uniform[3];  // axis with equal spacing: s = (X - x0) / h - 1/2; else the search through xcg
  const double *xcg[3];  // GLOBAL cell centres of a stretched axis, index -1..ng (ghost centres: mirror image / periodic image)
  const double *idx[3];  // LOCAL 1/dx (GridP::idx): the target cell's volume
};

// i0[d*L + l] = first support cell (GLOBAL index, may be out of range); w[(d*4 + a)*L + l] = phi weights
__global__ void k_ibm_weights(IbmP P, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z, int *__restrict__ i0, double *__restrict__ w)
{
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= P.L) return;
  const double pos[3] = {X[l], Y[l], Z[l]};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double s;  // position in units of the cell-centre index
    if (P.uniform[d]) s = (pos[d] - P.x0[d]) / P.h[d] - 0.5;
    else {
      // the interval [centre(c), centre(c+1)) that holds the marker, c = -1 .. ng-1 (linear extension beyond the ghost centres)
      const double *xc = P.xcg[d];
      int           lo = -1, hi = P.ng[d] - 1;
      while (lo < hi) {
        const int mid = lo + (hi - lo + 1) / 2;
        if (xc[mid] <= pos[d]) lo = mid;
        else hi = mid - 1;
      }
      s = (double)lo + (pos[d] - xc[lo]) / (xc[lo + 1] - xc[lo]);
    }
    const int    i = (P.kind == FL_DELTA_PESKIN4) ? (int)floor(s) - 1 : (int)floor(s + 0.5) - 1;
    for (int a = 0; a < 4; ++a) {
      const double r          = s - (double)(i + a);
      w[(d * 4 + a) * P.L + l] = a < P.S ? (P.kind == FL_DELTA_PESKIN4 ? phi_peskin4(r) : phi_roma3(r)) : 0.;
    }
    i0[d * P.L + l] = i;
  }
}

// LOCAL cell index of support entry a, or -1 when that cell is not owned by this rank.  Every rank sees every marker
// (markers are replicated); a support that straddles a block face is simply shared out between the two owners, and a
// support that crosses a periodic boundary wraps in GLOBAL index space first.
__device__ __forceinline__ int support_cell(const IbmP &P, int d, int i0, int a)
{
  int c = i0 + a;
  if (P.periodic[d]) {
    if (c < 0) c += P.ng[d];
    else if (c >= P.ng[d]) c -= P.ng[d];
  }
  c -= P.lo[d];
  return (c < 0 || c >= P.n[d]) ? -1 : c;
}

// the distinct tiles the support touches along each axis: at most 3 (S=4 over 8-cell tiles gives <= 2, a wrap adds one)
__device__ __forceinline__ int support_tiles(const IbmP &P, int d, int i0, int t[4])
{
  int nt = 0;
  for (int a = 0; a < P.S; ++a) {
    const int c = support_cell(P, d, i0, a);
    if (c < 0) continue;
    const int tt = c / TB;
    bool      dup = false;
    for (int b = 0; b < nt; ++b) dup |= (t[b] == tt);
    if (!dup) t[nt++] = tt;
  }
  return nt;
}

// pass 0: cnt[tile]++ ; pass 1: list[off[tile] + cursor[tile]++] = marker
__global__ void k_ibm_bin(IbmP P, const int *__restrict__ i0, int *__restrict__ cnt, const int *__restrict__ off, int *__restrict__ list, int pass)
{
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= P.L) return;
  int tx[4], ty[4], tz[4];
  const int nx = support_tiles(P, 0, i0[l], tx), ny = support_tiles(P, 1, i0[P.L + l], ty), nz = support_tiles(P, 2, i0[2 * P.L + l], tz);
  for (int c = 0; c < nz; ++c)
    for (int b = 0; b < ny; ++b)
      for (int a = 0; a < nx; ++a) {
        const int tile = (tz[c] * P.nt[1] + ty[b]) * P.nt[0] + tx[a];
        const int pos  = atomicAdd(&cnt[tile], 1);
        if (pass) list[off[tile] + pos] = (int)l;
      }
}

// exclusive scan of cnt[0..n) -> off[0..n], single block
__global__ void __launch_bounds__(256) k_ibm_scan(const int *__restrict__ cnt, int *__restrict__ off, int n)
{
  __shared__ int part[256];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256 * 16) {
    int loc[16], sum = 0;
    for (int a = 0; a < 16; ++a) {
      const int idx = base + threadIdx.x * 16 + a;
      loc[a]        = idx < n ? cnt[idx] : 0;
      sum += loc[a];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 per-thread sums
    for (int o = 1; o < 256; o <<= 1) {
      const int v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    int run = carry + part[threadIdx.x] - sum;
    for (int a = 0; a < 16; ++a) {
      const int idx = base + threadIdx.x * 16 + a;
      if (idx < n) off[idx] = run;
      run += loc[a];
    }
    __syncthreads();
    if (threadIdx.x == 255) carry += part[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) off[n] = carry;
}

// U[c*L + l] = sum over the support.  One wavefront per marker, lane = (a,b,c3) of the S^3 support.
__global__ void __launch_bounds__(256) k_ibm_interp(IbmP P, const int *__restrict__ i0, const double *__restrict__ w, int ncomp, int64_t ncell, const double *__restrict__ u, double *__restrict__ U)
{
  const int     lane = threadIdx.x & 63;
  const int64_t l    = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (l >= P.L) return;
  const int S = P.S;
  const int a = lane % S, b = (lane / S) % S, c3 = lane / (S * S);
  double    wt   = 0.;
  int64_t   cell = -1;
  if (c3 < S) {
    const int ci = support_cell(P, 0, i0[l], a), cj = support_cell(P, 1, i0[P.L + l], b), ck = support_cell(P, 2, i0[2 * P.L + l], c3);
    if (ci >= 0 && cj >= 0 && ck >= 0) {
      cell = ((int64_t)ck * P.n[1] + cj) * P.n[0] + ci;
      wt   = w[(0 * 4 + a) * P.L + l] * w[(1 * 4 + b) * P.L + l] * w[(2 * 4 + c3) * P.L + l];
    }
  }
  for (int c = 0; c < ncomp; ++c) {
    double v = cell >= 0 ? wt * u[(int64_t)c * ncell + cell] : 0.;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) U[(int64_t)c * P.L + l] = v;
  }
}

// f[c*ncell + x] += sum_l w_l(x) F[c*L + l] dV_l / (hx hy hz), gather over the tile's bin, markers in ascending id order
#ifdef FL_KBENCH_VARIANTS  // round 1's spreading loop (A/B runs)
__global__ void __launch_bounds__(256) k_ibm_spread_v1(IbmP P, const int *__restrict__ i0, const double *__restrict__ w, const int *__restrict__ off, const int *__restrict__ list, const int *__restrict__ active, int ncomp, int64_t ncell, const double *__restrict__ F,
                                                      const double *__restrict__ dV, double *__restrict__ f)
{
  __shared__ int    sraw[BIN_CHUNK];
  __shared__ int    si0[3][BIN_CHUNK];
  __shared__ double sw[3][4][BIN_CHUNK];
  __shared__ double sF[3][BIN_CHUNK];
  const int tile = active[blockIdx.x];  // only tiles with a non-empty bin are launched
  const int beg = off[tile], end = off[tile + 1];
  if (beg == end) return;
  const int tx = tile % P.nt[0], ty = (tile / P.nt[0]) % P.nt[1], tz = tile / (P.nt[0] * P.nt[1]);
  const bool   uni = P.uniform[0] && P.uniform[1] && P.uniform[2];
  const double ih  = uni ? 1. / (P.h[0] * P.h[1] * P.h[2]) : 1.;  // stretched grids: 1 / (volume of the target cell), applied per cell below
  // this thread's two cells: (ci, cj, ck) and (ci, cj, ck + 4)
  const int li = threadIdx.x & 7, lj = (threadIdx.x >> 3) & 7, lk = threadIdx.x >> 6;
  const int ci = tx * TB + li, cj = ty * TB + lj;
  double    acc[2][3] = {{0., 0., 0.}, {0., 0., 0.}};
  // the bin is staged in id-sorted chunks; since every chunk is rank-sorted and chunks are visited in list order, the
  // whole list must already be chunk-monotone -> sort the FULL bin when it fits (the common case), else fall back to
  // a two-level order (chunk order = list order) which is still deterministic because k_ibm_sort_bins sorted the list.
  for (int c0 = beg; c0 < end; c0 += BIN_CHUNK) {
    const int n = min(BIN_CHUNK, end - c0);
    __syncthreads();
    if ((int)threadIdx.x < n) sraw[threadIdx.x] = list[c0 + threadIdx.x];
    __syncthreads();
    if ((int)threadIdx.x < n) {
      const int m = sraw[threadIdx.x];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        si0[d][threadIdx.x] = i0[d * P.L + m];
#pragma unroll
        for (int a = 0; a < 4; ++a) sw[d][a][threadIdx.x] = w[(d * 4 + a) * P.L + m];
      }
      const double dv = dV[m] * ih;
      for (int c = 0; c < 3; ++c) sF[c][threadIdx.x] = c < ncomp ? F[(int64_t)c * P.L + m] * dv : 0.;
    }
    __syncthreads();
    if (ci < P.n[0] && cj < P.n[1]) {
      for (int e = 0; e < n; ++e) {
        int a = ci + P.lo[0] - si0[0][e], b = cj + P.lo[1] - si0[1][e];
        if (P.periodic[0]) { if (a < 0) a += P.ng[0]; else if (a >= P.ng[0]) a -= P.ng[0]; }
        if (P.periodic[1]) { if (b < 0) b += P.ng[1]; else if (b >= P.ng[1]) b -= P.ng[1]; }
        if (a < 0 || a >= P.S || b < 0 || b >= P.S) continue;
        const double wxy = sw[0][a][e] * sw[1][b][e];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int ck = tz * TB + lk + 4 * half;
          int       c3 = ck + P.lo[2] - si0[2][e];
          if (P.periodic[2]) { if (c3 < 0) c3 += P.ng[2]; else if (c3 >= P.ng[2]) c3 -= P.ng[2]; }
          if (ck >= P.n[2] || c3 < 0 || c3 >= P.S) continue;
          const double wt = wxy * sw[2][c3][e];
          acc[half][0] += wt * sF[0][e];
          acc[half][1] += wt * sF[1][e];
          acc[half][2] += wt * sF[2][e];
        }
      }
    }
  }
  if (ci < P.n[0] && cj < P.n[1])
    for (int half = 0; half < 2; ++half) {
      const int ck = tz * TB + lk + 4 * half;
      if (ck >= P.n[2]) continue;
      const int64_t cell = ((int64_t)ck * P.n[1] + cj) * P.n[0] + ci;
      const double  vinv = uni ? 1. : P.idx[0][ci] * P.idx[1][cj] * P.idx[2][ck];
      for (int c = 0; c < ncomp && c < 3; ++c)
        if (acc[half][c] != 0.) f[(int64_t)c * ncell + cell] += acc[half][c] * vinv;
    }
}
#endif  // FL_KBENCH_VARIANTS

// Round 4.  The loop above is a chain of dependent LDS reads per marker (its first cell -> the index into its weights -> the weight) behind two
// divergent tests, one wave per SIMD and block: about 300 cycles per marker and 150 markers per bin of config 4's sphere.  Here the staging step expands
// every marker's 1-D weights onto the tile's eight cells per axis (zero outside the support, periodic wrap and all), so the loop reads four weights and
// three forces at addresses that depend on nothing but the thread and the loop counter -- no test, no dependent read, unrolled.  A cell outside a marker's
// support adds an exact zero, so the sums (ascending marker id per cell, as before) are the same bits.
__global__ void __launch_bounds__(256) k_ibm_spread(IbmP P, const int *__restrict__ i0, const double *__restrict__ w, const int *__restrict__ off, const int *__restrict__ list, const int *__restrict__ active, int ncomp, int64_t ncell, const double *__restrict__ F,
                                                    const double *__restrict__ dV, double *__restrict__ f)
{
  __shared__ double swl[3][BIN_CHUNK][TB];  // weight of marker e on the tile's cell c of axis d
  __shared__ double sF[3][BIN_CHUNK];
  const int tile = active[blockIdx.x];  // only tiles with a non-empty bin are launched
  const int beg = off[tile], end = off[tile + 1];
  if (beg == end) return;
  const int tt[3] = {tile % P.nt[0], (tile / P.nt[0]) % P.nt[1], tile / (P.nt[0] * P.nt[1])};
  const bool   uni = P.uniform[0] && P.uniform[1] && P.uniform[2];
  const double ih  = uni ? 1. / (P.h[0] * P.h[1] * P.h[2]) : 1.;  // stretched grids: 1 / (volume of the target cell), applied per cell below
  // this thread's two cells: (ci, cj, ck) and (ci, cj, ck + 4)
  const int li = threadIdx.x & 7, lj = (threadIdx.x >> 3) & 7, lk = threadIdx.x >> 6;
  const int ci = tt[0] * TB + li, cj = tt[1] * TB + lj;
  double    acc[2][3] = {{0., 0., 0.}, {0., 0., 0.}};
  for (int c0 = beg; c0 < end; c0 += BIN_CHUNK) {  // bins are sorted by marker id (k_ibm_sort_bins): chunks in list order keep the order
    const int n = min(BIN_CHUNK, end - c0);
    __syncthreads();
    if ((int)threadIdx.x < n) {
      const int m = list[c0 + threadIdx.x];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int first = i0[d * P.L + m];
        double    wd[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) wd[a] = w[(d * 4 + a) * P.L + m];
#pragma unroll
        for (int c = 0; c < TB; ++c) {
          int a = tt[d] * TB + c + P.lo[d] - first;
          if (P.periodic[d]) { if (a < 0) a += P.ng[d]; else if (a >= P.ng[d]) a -= P.ng[d]; }
          swl[d][threadIdx.x][c] = (a < 0 || a >= P.S) ? 0. : (a == 0 ? wd[0] : (a == 1 ? wd[1] : (a == 2 ? wd[2] : wd[3])));
        }
      }
      const double dv = dV[m] * ih;
      for (int c = 0; c < 3; ++c) sF[c][threadIdx.x] = c < ncomp ? F[(int64_t)c * P.L + m] * dv : 0.;
    }
    __syncthreads();
#pragma unroll 4
    for (int e = 0; e < n; ++e) {
      const double wxy = swl[0][e][li] * swl[1][e][lj];
      const double wt0 = wxy * swl[2][e][lk], wt1 = wxy * swl[2][e][lk + 4];
      const double f0 = sF[0][e], f1 = sF[1][e], f2 = sF[2][e];
      acc[0][0] += wt0 * f0;
      acc[0][1] += wt0 * f1;
      acc[0][2] += wt0 * f2;
      acc[1][0] += wt1 * f0;
      acc[1][1] += wt1 * f1;
      acc[1][2] += wt1 * f2;
    }
  }
  if (ci < P.n[0] && cj < P.n[1])
    for (int half = 0; half < 2; ++half) {
      const int ck = tt[2] * TB + lk + 4 * half;
      if (ck >= P.n[2]) continue;
      const int64_t cell = ((int64_t)ck * P.n[1] + cj) * P.n[0] + ci;
      const double  vinv = uni ? 1. : P.idx[0][ci] * P.idx[1][cj] * P.idx[2][ck];
      for (int c = 0; c < ncomp && c < 3; ++c)
        if (acc[half][c] != 0.) f[(int64_t)c * ncell + cell] += acc[half][c] * vinv;
    }
}

// compact list of the tiles whose bin is not empty (order irrelevant: every tile owns its cells)
__global__ void k_ibm_active(const int *__restrict__ off, int ntiles, int *__restrict__ active, int *__restrict__ nactive)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ntiles && off[t + 1] > off[t]) active[atomicAdd(nactive, 1)] = t;
}

// in-place ascending sort of every bin (marker ids are unique within a bin): rank sort, one block per tile
__global__ void __launch_bounds__(256) k_ibm_sort_bins(const int *__restrict__ off, int *__restrict__ list, int *__restrict__ scratch, const int *__restrict__ active)
{
  const int tile = active[blockIdx.x];
  const int beg = off[tile], end = off[tile + 1], n = end - beg;
  if (n <= 1) return;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int v = list[beg + e];
    int       rank = 0;
    for (int o = 0; o < n; ++o) rank += list[beg + o] < v;
    scratch[beg + rank] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += 256) list[beg + e] = scratch[beg + e];
}

}  // namespace fl

using namespace fl;

struct fl_ibm {
  fl_poisson *gp = nullptr;
  IbmP        P;
  double     *X = nullptr, *Y = nullptr, *Z = nullptr, *w = nullptr;
  int        *i0 = nullptr, *cnt = nullptr, *off = nullptr, *list = nullptr, *scratch = nullptr, *active = nullptr, *nact_dev = nullptr;
  int         ntiles = 0, listcap = 0, nactive = 0;
  double     *xcg[3] = {nullptr, nullptr, nullptr};  // device copies of the extended global centre arrays (stretched axes only)
};

static int ibm_rebin(fl_ibm *m)
{
  fl_poisson *h = m->gp;
  hipStream_t s = h->stream;
  const IbmP &P = m->P;
  const int   nb = (int)((P.L + 255) / 256);
  hipLaunchKernelGGL(k_ibm_weights, dim3(nb), dim3(256), 0, s, P, m->X, m->Y, m->Z, m->i0, m->w);
  FL_HIP(hipMemsetAsync(m->cnt, 0, sizeof(int) * (m->ntiles + 1), s));
  hipLaunchKernelGGL(k_ibm_bin, dim3(nb), dim3(256), 0, s, P, m->i0, m->cnt, m->off, m->list, 0);
  hipLaunchKernelGGL(k_ibm_scan, dim3(1), dim3(256), 0, s, m->cnt, m->off, m->ntiles);
  FL_HIP(hipMemsetAsync(m->cnt, 0, sizeof(int) * (m->ntiles + 1), s));
  hipLaunchKernelGGL(k_ibm_bin, dim3(nb), dim3(256), 0, s, P, m->i0, m->cnt, m->off, m->list, 1);
  FL_HIP(hipMemsetAsync(m->nact_dev, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_ibm_active, dim3((m->ntiles + 255) / 256), dim3(256), 0, s, m->off, m->ntiles, m->active, m->nact_dev);
  FL_HIP(hipMemcpyAsync(&m->nactive, m->nact_dev, sizeof(int), hipMemcpyDeviceToHost, s));
  FL_HIP(hipStreamSynchronize(s));  // set-up time only (create / marker update), never inside interp / spread
  // sort only the non-empty bins (ascending marker id -> run-independent accumulation order)
  if (m->nactive > 0) hipLaunchKernelGGL(k_ibm_sort_bins, dim3(m->nactive), dim3(256), 0, s, m->off, m->list, m->scratch, m->active);
  FL_HIP(hipGetLastError());
  return 0;
}

extern "C" int fl_ibm_create(fl_poisson *h, int kind, int64_t L, const double *X, const double *Y, const double *Z, fl_ibm **out)
{
  if (!h || !X || !Y || !Z || !out) return FL_ERR_ARG_NULL;
  if (kind != FL_DELTA_PESKIN4 && kind != FL_DELTA_ROMA3) return FL_ERR_ARG_OUTOFRANGE;
  if (L < 1 || L > (int64_t)1 << 27) return FL_ERR_ARG_OUTOFRANGE;
  *out = nullptr;
  FL_HIP(hipSetDevice(h->device));
  fl_ibm *m = new fl_ibm();
  m->gp     = h;
  IbmP &P   = m->P;
  P.kind    = kind;
  P.S       = kind == FL_DELTA_PESKIN4 ? 4 : 3;
  P.L       = L;
  for (int d = 0; d < 3; ++d) {
    const Axis &A = h->ax[d];
    P.n[d]        = (int)h->dec.len[d];
    P.lo[d]       = (int)h->dec.lo[d];
    P.ng[d]       = (int)A.n;
    P.periodic[d] = A.periodic ? 1 : 0;
    P.x0[d]       = A.xf[0];
    P.h[d]        = (A.xf[A.n] - A.xf[0]) / (double)A.n;
    P.uniform[d]  = 1;
    for (int64_t i = 0; i < A.n; ++i)
      if (std::fabs((A.xf[i + 1] - A.xf[i]) - P.h[d]) > 1e-10 * P.h[d]) P.uniform[d] = 0;
    P.xcg[d] = nullptr;
    P.idx[d] = h->g.idx[d];
    if (!P.uniform[d]) {
      // extended centres, index -1..n: the periodic images (stored by build_axis) or the mirror images in the walls
      std::vector<double> xc((size_t)A.n + 2);
      for (int64_t i = -1; i <= A.n; ++i) xc[(size_t)(i + 1)] = A.xcc(i);
      if (!A.periodic) {
        xc[0]                  = 2. * A.xf[0] - A.xcc(0);
        xc[(size_t)A.n + 1]    = 2. * A.xf[A.n] - A.xcc(A.n - 1);
      }
      if (hipMalloc((void **)&m->xcg[d], sizeof(double) * xc.size()) != hipSuccess || hipMemcpy(m->xcg[d], xc.data(), sizeof(double) * xc.size(), hipMemcpyHostToDevice) != hipSuccess) {
        fl_ibm_destroy(m);
        return FL_ERR_GPU;
      }
      P.xcg[d] = m->xcg[d] + 1;
    }
    if (P.periodic[d] && P.ng[d] < 2 * P.S) {
      fl_ibm_destroy(m);
      return FL_ERR_ARG_OUTOFRANGE;
    }
    P.nt[d] = (P.n[d] + TB - 1) / TB;
  }
  m->ntiles  = P.nt[0] * P.nt[1] * P.nt[2];
  m->listcap = (int)std::min<int64_t>(L * 27, (int64_t)1 << 30);
  int rc     = 0;
  rc |= fl_dev_alloc(h, (void **)&m->X, sizeof(double) * L, false);
  rc |= fl_dev_alloc(h, (void **)&m->Y, sizeof(double) * L, false);
  rc |= fl_dev_alloc(h, (void **)&m->Z, sizeof(double) * L, false);
  rc |= fl_dev_alloc(h, (void **)&m->w, sizeof(double) * 12 * L, false);
  rc |= fl_dev_alloc(h, (void **)&m->i0, sizeof(int) * 3 * L, false);
  rc |= fl_dev_alloc(h, (void **)&m->cnt, sizeof(int) * (m->ntiles + 1), true);
  rc |= fl_dev_alloc(h, (void **)&m->off, sizeof(int) * (m->ntiles + 1), true);
  rc |= fl_dev_alloc(h, (void **)&m->list, sizeof(int) * m->listcap, true);
  rc |= fl_dev_alloc(h, (void **)&m->scratch, sizeof(int) * m->listcap, true);
  rc |= fl_dev_alloc(h, (void **)&m->active, sizeof(int) * m->ntiles, true);
  rc |= fl_dev_alloc(h, (void **)&m->nact_dev, sizeof(int), true);
  if (rc) {
    fl_ibm_destroy(m);
    return FL_ERR_MEM;
  }
  *out = m;
  return fl_ibm_update(m, X, Y, Z);
}

extern "C" int fl_ibm_update(fl_ibm *m, const double *X, const double *Y, const double *Z)
{
  if (!m || !X || !Y || !Z) return FL_ERR_ARG_NULL;
  fl_poisson *h = m->gp;
  FL_HIP(hipSetDevice(h->device));
  FL_HIP(hipMemcpyAsync(m->X, X, sizeof(double) * m->P.L, hipMemcpyDeviceToDevice, h->stream));
  FL_HIP(hipMemcpyAsync(m->Y, Y, sizeof(double) * m->P.L, hipMemcpyDeviceToDevice, h->stream));
  FL_HIP(hipMemcpyAsync(m->Z, Z, sizeof(double) * m->P.L, hipMemcpyDeviceToDevice, h->stream));
  return ibm_rebin(m);
}

extern "C" int fl_ibm_interp(fl_ibm *m, int ncomp, const double *u, double *U)
{
  if (!m || !u || !U) return FL_ERR_ARG_NULL;
  if (ncomp < 1) return FL_ERR_ARG_OUTOFRANGE;
  fl_poisson *h = m->gp;
  FL_HIP(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_ibm_interp, dim3((unsigned)((m->P.L + 3) / 4)), dim3(256), 0, h->stream, m->P, m->i0, m->w, ncomp, h->ncell, u, U);
  FL_HIP(hipGetLastError());
  // multi-rank: every rank summed over the support cells it owns; the marker value is the sum over ranks
  if (h->multi) {
    if (h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
    FL_CHK(h->comm.allreduce(h->stream, U, (int)(m->P.L * ncomp)));
  }
  return FL_SUCCESS;
}

// bin statistics of the current marker positions (experiments): tiles with a non-empty bin, entries of all bins, the largest bin
#ifdef FL_KBENCH_VARIANTS  // bin statistics for tools/ibm_bench.py
extern "C" int fldbg_ibm_stats(fl_ibm *m, int *nactive, int *entries, int *maxbin)
{
  if (!m) return FL_ERR_ARG_NULL;
  std::vector<int> off((size_t)m->ntiles + 1);
  FL_HIP(hipStreamSynchronize(m->gp->stream));
  FL_HIP(hipMemcpy(off.data(), m->off, sizeof(int) * off.size(), hipMemcpyDeviceToHost));
  int mx = 0;
  for (int t = 0; t < m->ntiles; ++t) mx = std::max(mx, off[(size_t)t + 1] - off[(size_t)t]);
  if (nactive) *nactive = m->nactive;
  if (entries) *entries = off[(size_t)m->ntiles];
  if (maxbin) *maxbin = mx;
  return FL_SUCCESS;
}
#endif  // FL_KBENCH_VARIANTS

extern "C" int fl_ibm_spread(fl_ibm *m, int ncomp, const double *F, const double *dV, double *f)
{
  if (!m || !F || !dV || !f) return FL_ERR_ARG_NULL;
  if (ncomp < 1 || ncomp > 3) return FL_ERR_ARG_OUTOFRANGE;
  fl_poisson *h = m->gp;
  FL_HIP(hipSetDevice(h->device));
  if (m->nactive > 0) {
#ifdef FL_KBENCH_VARIANTS
    if (FL_VARIANT(ibm_spread, 0) == 1)  // round 1's loop over the bin (A/B runs)
      hipLaunchKernelGGL(k_ibm_spread_v1, dim3(m->nactive), dim3(256), 0, h->stream, m->P, m->i0, m->w, m->off, m->list, m->active, ncomp, h->ncell, F, dV, f);
    else
#endif
      hipLaunchKernelGGL(k_ibm_spread, dim3(m->nactive), dim3(256), 0, h->stream, m->P, m->i0, m->w, m->off, m->list, m->active, ncomp, h->ncell, F, dV, f);
  }
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_ibm_destroy(fl_ibm *m)
{
  if (!m) return FL_SUCCESS;
  if (m->gp) (void)hipStreamSynchronize(m->gp->stream);
  for (void *p : {(void *)m->X, (void *)m->Y, (void *)m->Z, (void *)m->w, (void *)m->i0, (void *)m->cnt, (void *)m->off, (void *)m->list, (void *)m->scratch, (void *)m->active, (void *)m->nact_dev, (void *)m->xcg[0],
                  (void *)m->xcg[1], (void *)m->xcg[2]})
    if (p) (void)hipFree(p);
  delete m;
  return FL_SUCCESS;
}
