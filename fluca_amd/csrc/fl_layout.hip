// fl_layout.hip -- DMStag vectors <-> the arrays of this library (the de-/interleaving a PETSc-side caller of the C-ABI needs).
//
// The reference keeps its fields in DMStag vectors (fluca/src/mesh/impl/cart/cart.c:88-116): sdm (1 dof per element: p),
// vdm (3 dof per element: the velocity components of a cell next to each other), Sdm (1 dof per face: face-normal velocity;
// the BACK, DOWN and LEFT face of an element are adjacent) and Vdm (3 dof per face: v0interp, cnlinearcart3d.c:896-905).
// The library takes one array per field, x fastest (fluca_hip.h).  Two DMStag representations are converted, on the device:
//   LOCAL  (DMStagVecGetArray on a local vector): arr[k][j][i][slot] over the ghosted box DMStagGetGhostCorners reports,
//          every element with all DMStagGetEntriesPerElement slots, slot from DMStagGetLocationSlot;
//   GLOBAL (the array of a global vector, what PCApply_ABF is handed): the rank's elements x fastest with all their slots,
//          and behind the last element of a non-periodic axis a partial element that holds only the dofs on its low face
//          (PETSc, DMSetUp_Stag_3d: entriesPerElementRow = n0 * entriesPerElement + entriesPerFace on the last rank, ...).
//          Only strata 2 (faces) and 3 (elements) may carry dofs -- all the reference uses.
// PETSc is not available in this build environment: both orderings follow PETSc's documented DMStag layout and are checked here
// against an independent numpy enumeration of it (tests/test_gpu_layout.py), not against PETSc itself.
#include "fl_handle.h"

namespace fl {

struct LayoutK {
  int     n[3];      // extents of the library array (cells, or faces with the extra face)
  int64_t s[3];      // strides of the DMStag array in doubles per element step along x, y, z
  int64_t base;      // offset of item (0,0,0)
  // GLOBAL ordering only: where rows / layers change shape
  int     global;
  int     nel[3];    // owned elements
  int     epe;       // entries per full element
  int     d2;        // dofs per face
  int     axis;      // 0 cells, 1..3 faces of x, y, z
  int     comp;
  int     slot;      // slot of (location, comp) inside a full element
  int64_t row, layer;  // entries per element row / layer
};

// offset of item (i,j,k) in a GLOBAL DMStag array (faces: item index = element index of the face's owner)
__device__ __forceinline__ int64_t global_off(const LayoutK &L, int i, int j, int k)
{
  if (k < L.nel[2]) {
    const int64_t lo = (int64_t)k * L.layer;
    if (j < L.nel[1]) {
      const int64_t ro = lo + (int64_t)j * L.row;
      if (i < L.nel[0]) return ro + (int64_t)i * L.epe + L.slot;
      return ro + (int64_t)L.nel[0] * L.epe + L.comp;  // partial element behind the row: LEFT faces only
    }
    return lo + (int64_t)L.nel[1] * L.row + (int64_t)i * L.d2 + L.comp;  // partial row behind the layer: DOWN faces only
  }
  return (int64_t)L.nel[2] * L.layer + ((int64_t)j * L.nel[0] + i) * L.d2 + L.comp;  // partial layer at the end: BACK faces only
}

template <bool TO>
__global__ void __launch_bounds__(256) k_layout(LayoutK L, const double *__restrict__ src, double *__restrict__ dst)
{
  const int64_t total = (int64_t)L.n[0] * L.n[1] * L.n[2];
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int     i = (int)(q % L.n[0]);
    const int64_t r = q / L.n[0];
    const int     j = (int)(r % L.n[1]), k = (int)(r / L.n[1]);
    const int64_t o = L.global ? global_off(L, i, j, k) : L.base + (int64_t)k * L.s[2] + (int64_t)j * L.s[1] + (int64_t)i * L.s[0];
    if (TO) dst[o] = src[q];
    else dst[q] = src[o];
  }
}

}  // namespace fl

using namespace fl;

namespace {

int item_extents(const fl_poisson *h, int what, int n[3])
{
  const GridP &g = h->g;
  n[0] = g.nx;
  n[1] = g.ny;
  n[2] = g.nz;
  if (what < 0 || what > 3) return FL_ERR_ARG_OUTOFRANGE;
  if (what == 1) n[0] = g.fx;
  if (what == 2) n[1] = g.fy;
  if (what == 3) n[2] = g.fz;
  return 0;
}

int run(fl_poisson *h, const LayoutK &L, bool to, const double *src, double *dst)
{
  const int64_t total = (int64_t)L.n[0] * L.n[1] * L.n[2];
  const int     nb    = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
  if (to) hipLaunchKernelGGL(k_layout<true>, dim3(nb), dim3(256), 0, h->stream, L, src, dst);
  else hipLaunchKernelGGL(k_layout<false>, dim3(nb), dim3(256), 0, h->stream, L, src, dst);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

int local_layout(fl_poisson *h, const fl_dmstag_local *D, int what, int slot, LayoutK &L)
{
  if (!h || !D) return FL_ERR_ARG_NULL;
  FL_CHK(item_extents(h, what, L.n));
  if (D->entries < 1 || slot < 0 || slot >= D->entries) return FL_ERR_ARG_OUTOFRANGE;
  for (int d = 0; d < 3; ++d) {
    // the owned items (and the extra face) must lie inside the ghosted box
    const int64_t lo = D->start[d] - D->gstart[d];
    if (D->start[d] != h->dec.lo[d] || lo < 0 || lo + L.n[d] > D->gsize[d]) return FL_ERR_ARG_WRONG;
  }
  L.global = 0;
  L.s[0]   = D->entries;
  L.s[1]   = D->entries * D->gsize[0];
  L.s[2]   = D->entries * D->gsize[0] * D->gsize[1];
  L.base   = (D->start[2] - D->gstart[2]) * L.s[2] + (D->start[1] - D->gstart[1]) * L.s[1] + (D->start[0] - D->gstart[0]) * L.s[0] + slot;
  return 0;
}

int global_layout(fl_poisson *h, const int dof[4], int what, int comp, LayoutK &L)
{
  if (!h || !dof) return FL_ERR_ARG_NULL;
  FL_CHK(item_extents(h, what, L.n));
  if (dof[0] != 0 || dof[1] != 0) return FL_ERR_SUP;  // vertex / edge dofs: not a layout of the reference
  const int nd = what == 0 ? dof[3] : dof[2];
  if (dof[2] < 0 || dof[3] < 0 || comp < 0 || comp >= nd) return FL_ERR_ARG_OUTOFRANGE;
  const GridP &g = h->g;
  L.global = 1;
  L.nel[0] = g.nx;
  L.nel[1] = g.ny;
  L.nel[2] = g.nz;
  L.d2     = dof[2];
  L.epe    = 3 * dof[2] + dof[3];
  L.axis   = what;
  L.comp   = comp;
  // slots inside a full element: BACK faces, DOWN faces, LEFT faces, ELEMENT (DMStag's location order in 3-D)
  L.slot   = what == 0 ? 3 * dof[2] + comp : (what == 3 ? comp : (what == 2 ? dof[2] + comp : 2 * dof[2] + comp));
  const bool ex = g.fx > g.nx, ey = g.fy > g.ny;  // partial elements exist behind the last element of a non-periodic axis
  L.row   = (int64_t)g.nx * L.epe + (ex ? dof[2] : 0);
  L.layer = (int64_t)g.ny * L.row + (ey ? (int64_t)g.nx * dof[2] : 0);
  for (int d = 0; d < 3; ++d) L.s[d] = 0;
  L.base = 0;
  return 0;
}

}  // namespace

extern "C" int fl_layout_from_dmstag_local(fl_poisson *h, const fl_dmstag_local *D, int what, int slot, const double *local_dev, double *out_dev)
{
  if (!local_dev || !out_dev) return FL_ERR_ARG_NULL;
  LayoutK L{};
  FL_CHK(local_layout(h, D, what, slot, L));
  FL_HIP(hipSetDevice(h->device));
  return run(h, L, false, local_dev, out_dev);
}

extern "C" int fl_layout_to_dmstag_local(fl_poisson *h, const fl_dmstag_local *D, int what, int slot, const double *in_dev, double *local_dev)
{
  if (!local_dev || !in_dev) return FL_ERR_ARG_NULL;
  LayoutK L{};
  FL_CHK(local_layout(h, D, what, slot, L));
  FL_HIP(hipSetDevice(h->device));
  return run(h, L, true, in_dev, local_dev);
}

extern "C" int fl_layout_from_dmstag_global(fl_poisson *h, const int dof[4], int what, int comp, const double *global_dev, double *out_dev)
{
  if (!global_dev || !out_dev) return FL_ERR_ARG_NULL;
  LayoutK L{};
  FL_CHK(global_layout(h, dof, what, comp, L));
  FL_HIP(hipSetDevice(h->device));
  return run(h, L, false, global_dev, out_dev);
}

extern "C" int fl_layout_to_dmstag_global(fl_poisson *h, const int dof[4], int what, int comp, const double *in_dev, double *global_dev)
{
  if (!global_dev || !in_dev) return FL_ERR_ARG_NULL;
  LayoutK L{};
  FL_CHK(global_layout(h, dof, what, comp, L));
  FL_HIP(hipSetDevice(h->device));
  return run(h, L, true, in_dev, global_dev);
}

extern "C" int fl_dmstag_global_entries(const fl_poisson *h, const int dof[4], int64_t *entries)
{
  if (!h || !dof || !entries) return FL_ERR_ARG_NULL;
  if (dof[0] != 0 || dof[1] != 0) return FL_ERR_SUP;
  const GridP  &g   = h->g;
  const bool    ex = g.fx > g.nx, ey = g.fy > g.ny, ez = g.fz > g.nz;
  const int64_t epe = 3 * (int64_t)dof[2] + dof[3];
  const int64_t row = (int64_t)g.nx * epe + (ex ? dof[2] : 0), layer = (int64_t)g.ny * row + (ey ? (int64_t)g.nx * dof[2] : 0);
  *entries = (int64_t)g.nz * layer + (ez ? (int64_t)g.nx * g.ny * dof[2] : 0);
  return FL_SUCCESS;
}
