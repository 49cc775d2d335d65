// fl_api.hip -- C-ABI entry points of libflucahip.so (see include/fluca_hip.h), handle management, the Krylov drivers
// and the two halo transports (RCCL Send/Recv; host-staged callbacks).
#include <new>

#include <cctype>
#include <mutex>

#include "fl_handle.h"

int fl_dev_alloc(fl_poisson *h, void **p, size_t bytes, bool zero)
{
  FL_HIP(hipMalloc(p, bytes ? bytes : 8));
  if (zero) FL_HIP(hipMemsetAsync(*p, 0, bytes ? bytes : 8, h->stream));
  return 0;
}

template <class T>
static int upload_table(fl_poisson *h, const std::vector<T> &host, const T **dev, int shift)
{
  void *p = nullptr;
  FL_HIP(hipMalloc(&p, sizeof(T) * std::max<size_t>(host.size(), 1)));
  FL_HIP(hipMemcpy(p, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
  h->tables.push_back(p);
  *dev = (const T *)p + shift;
  return 0;
}

static int face_count(const fl_poisson *h, int d) { return d == 0 ? h->g.fx : (d == 1 ? h->g.fy : h->g.fz); }
static int plane_size(const fl_poisson *h, int d)
{
  const GridP &g = h->g;
  return d == 0 ? g.ny * g.nz : (d == 1 ? g.nx * g.nz : g.nx * g.ny);
}

// "abi N": bumped whenever a struct of include/fluca_hip.h grows or an entry point changes its meaning (FL_ABI_VERSION there): a caller built
// against another header must not be handed this library.  5: fl_ksp_opts carries cg_single_reduction (round 3's trailing field), the
// momentum solve accepts FL_KSP_CHEBYSHEV, fl_momentum_gershgorin / fl_momentum_chebyshev_interval exist.
extern "C" const char *fl_build_id(void);  // lib/fl_build_id.cpp, written by fluca_amd/build.py: a hash over every source, header and compiler flag
extern "C" const char *fl_version(void)
{
  static const std::string v = std::string("fluca_amd 0.3 (gfx950, abi 6, sources ") + fl_build_id() + ")";
  return v.c_str();
}
extern "C" int fl_abi_version(void) { return FL_ABI_VERSION; }

extern "C" void fl_ksp_opts_default(fl_ksp_opts *o)
{
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->type             = FL_KSP_CG;
  o->pc               = FL_PC_JACOBI;
  o->norm_type        = FL_NORM_PRECONDITIONED;
  o->remove_nullspace = 1;
  o->maxit            = 10000;
  o->rtol             = 1e-5;
  o->atol             = 1e-50;
  o->dtol             = 1e5;
  o->check_every      = 16;
}

// everything that can fail after the handle exists; the caller destroys the handle on any error
static int poisson_init(fl_poisson *h, const fl_grid *grid, const int bc[6], double kappa, const fl_decomp *decomp, int device)
{
  h->device     = device;
  h->kappa      = kappa;
  std::memcpy(h->bc, bc, sizeof(int) * 6);
  for (int d = 0; d < 3; ++d) {
    int rc = build_axis(h->ax[d], grid->n[d], grid->xf[d], grid->xc[d], bc[2 * d], bc[2 * d + 1], kappa);
    if (rc) return rc;
  }
  if (decomp) h->dec = *decomp;
  else
    for (int d = 0; d < 3; ++d) {
      h->dec.ranks[d] = 1;
      h->dec.coord[d] = 0;
      h->dec.lo[d]    = 0;
      h->dec.len[d]   = grid->n[d];
    }
  int periodic[3];
  for (int d = 0; d < 3; ++d) {
    const fl_decomp &D = h->dec;
    periodic[d]        = h->ax[d].periodic;
    if (D.ranks[d] < 1 || D.coord[d] < 0 || D.coord[d] >= D.ranks[d] || D.len[d] < 1 || D.lo[d] < 0 || D.lo[d] + D.len[d] > grid->n[d] || D.len[d] > 100000) {
      return FL_ERR_ARG_OUTOFRANGE;
    }
    if ((D.coord[d] == 0) != (D.lo[d] == 0) || (D.coord[d] == D.ranks[d] - 1) != (D.lo[d] + D.len[d] == grid->n[d])) {
      return FL_ERR_ARG_WRONG;
    }
    h->wrap_local[d] = periodic[d] && D.ranks[d] == 1;
  }
  h->multi = h->dec.ranks[0] * h->dec.ranks[1] * h->dec.ranks[2] > 1;
  {
    if (knob(K_comm_loopback) != 0 && !h->multi && (periodic[0] || periodic[1] || periodic[2])) {
      h->loopback = h->comm.loopback = true;
      h->multi    = true;
      for (int d = 0; d < 3; ++d) h->wrap_local[d] = false;
    }
  }
  for (int b = 0; b < 6; ++b) h->nbr[b] = h->multi ? fl_decomp_neighbor(&h->dec, periodic, b) : -1;

  FL_HIP(hipSetDevice(device));
  FL_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  FL_HIP(hipEventCreate(&h->ev0));
  FL_HIP(hipEventCreate(&h->ev1));

  // ---- local tables ---------------------------------------------------------------------------------------------
  GridP &g = h->g;
  std::memset(&g, 0, sizeof(g));
  g.nx = (int)h->dec.len[0];
  g.ny = (int)h->dec.len[1];
  g.nz = (int)h->dec.len[2];
  int fl_[3];
  for (int d = 0; d < 3; ++d) {
    const bool last = h->dec.coord[d] == h->dec.ranks[d] - 1;
    fl_[d]          = (int)h->dec.len[d] + ((last && !h->ax[d].periodic) ? 1 : 0);
  }
  g.fx    = fl_[0];
  g.fy    = fl_[1];
  g.fz    = fl_[2];
  // Row-interleaved vector layout (FLUCA_INTERLEAVE=NV > 1): row (j,k) of vector v lives at ((k*sy + j)*NV + v)*sx0, i.e. the
  // same row of all NV solver vectors is contiguous in memory and a kernel that streams several vectors sweeps ONE
  // address range instead of NV ranges a gigabyte apart.  Kernels only ever see (pointer, row stride, plane stride).
  {
    h->nv_il = FL_VARIANT(interleave, FL_DEFAULT_INTERLEAVE);
    if (h->nv_il < 2) h->nv_il = 1;
    if (h->nv_il > 8) h->nv_il = 8;
  }
  // Ghost width of the padded layout: 1 = the reference's star stencil (cart.c:66,91), what every operator needs; 2 where a neighbouring RANK
  // sits behind a boundary, so that the two-deep exchange of fl_fill_ghosts_deep has somewhere to put its second layer and the fused
  // two-step smoother (k_cheb2) can run on several ranks.  Kernels only ever see (off0, sx, sxy): the width is a property of the handle,
  // not of the kernels.  FLUCA_GHOST_WIDTH=2 forces the wide layout on a single rank (tests: the whole suite must not care).
  {
    const int forced = knob(K_ghost_width);
    h->gw            = forced > 0 ? forced : (h->multi && !h->loopback ? 2 : 1);
    if (h->gw < 1 || h->gw > 2 || h->nv_il > 1) h->gw = 1;
  }
  const int gw = h->gw;
  h->sx0  = ((PADX + g.nx + gw + 15) / 16) * 16;
  g.sx    = h->sx0 * h->nv_il;
  g.sxy   = (int64_t)g.sx * (g.ny + 2 * gw);
  g.off0  = (int64_t)gw * g.sxy + (int64_t)gw * g.sx + PADX;
  g.kappa = kappa;
  h->padlen = (size_t)g.sxy * (g.nz + 2 * gw) + 256;  // doubles spanned by one vector (interleaved: by the whole slab)
  h->ncell  = (int64_t)g.nx * g.ny * g.nz;
  h->nface[0] = (int64_t)g.fx * g.ny * g.nz;
  h->nface[1] = (int64_t)g.nx * g.fy * g.nz;
  h->nface[2] = (int64_t)g.nx * g.ny * g.fz;
  const double inf = std::numeric_limits<double>::infinity();
  for (int d = 0; d < 3; ++d) {
    const Axis   &A  = h->ax[d];
    const int64_t lo = h->dec.lo[d], len = h->dec.len[d], n = A.n;
    std::vector<double> sl(len + 2), sc(len + 2), sh(len + 2), idx(len);
    for (int64_t i = -1; i <= len; ++i) {
      int64_t gi = lo + i;
      bool    in = true;
      if (gi < 0 || gi >= n) {
        if (A.periodic) gi = (gi + n) % n;
        else in = false;
      }
      sl[i + 1] = in ? A.sl[gi] : 0.;
      sc[i + 1] = in ? A.sc[gi] : inf;  // wall ghost: 1/diag = 0
      sh[i + 1] = in ? A.sh[gi] : 0.;
    }
    for (int64_t i = 0; i < len; ++i) idx[i] = A.idx[lo + i];
    std::vector<double> ga0(fl_[d]), ga1(fl_[d]);
    std::vector<int>    gc0(fl_[d]);
    for (int f = 0; f < fl_[d]; ++f) {
      ga0[f] = A.ga0[lo + f];
      ga1[f] = A.ga1[lo + f];
      gc0[f] = (int)(A.gc0[lo + f] - lo);
    }
    std::vector<int>    Gs(len);
    std::vector<double> Gv0(len), Gv1(len), Gv2(len);
    for (int64_t i = 0; i < len; ++i) {
      Gs[i]  = (int)(A.Gs[lo + i] - lo);
      Gv0[i] = A.Gv0[lo + i];
      Gv1[i] = A.Gv1[lo + i];
      Gv2[i] = A.Gv2[lo + i];
      if (Gs[i] < -1 || Gs[i] + (Gv2[i] != 0. ? 2 : 1) > len) {
        // a one-sided wall row that leaves the block + its single ghost layer
        return FL_ERR_SUP;
      }
    }
    int rc = 0;
    rc |= upload_table(h, sl, &g.sl[d], 1);
    rc |= upload_table(h, sc, &g.sc[d], 1);
    rc |= upload_table(h, sh, &g.sh[d], 1);
    rc |= upload_table(h, idx, &g.idx[d], 0);
    rc |= upload_table(h, ga0, &g.ga0[d], 0);
    rc |= upload_table(h, ga1, &g.ga1[d], 0);
    rc |= upload_table(h, gc0, &g.gc0[d], 0);
    rc |= upload_table(h, Gs, &g.Gs[d], 0);
    rc |= upload_table(h, Gv0, &g.Gv0[d], 0);
    rc |= upload_table(h, Gv1, &g.Gv1[d], 0);
    rc |= upload_table(h, Gv2, &g.Gv2[d], 0);
    if (rc) return FL_ERR_GPU;
  }
  FL_HIP(hipMalloc((void **)&h->scal, sizeof(KspScal)));
  FL_HIP(hipHostMalloc((void **)&h->scal_host, sizeof(KspScal)));
  FL_HIP(hipMalloc((void **)&h->sums, sizeof(double) * NSLOT));
  FL_HIP(hipMalloc((void **)&h->tickets, sizeof(unsigned) * 2));
  FL_HIP(hipMemset(h->tickets, 0, sizeof(unsigned) * 2));
  return FL_SUCCESS;
}

extern "C" int fl_poisson_create(const fl_grid *grid, const int bc[6], double kappa, const fl_decomp *decomp, int device, fl_poisson **out)
{
  if (!grid || !bc || !out) return FL_ERR_ARG_NULL;
  *out = nullptr;
  for (int d = 0; d < 3; ++d)
    if (grid->n[d] < 1 || grid->n[d] > (int64_t)1 << 30 || !grid->xf[d]) return FL_ERR_ARG_OUTOFRANGE;
  if (!(kappa > 0.) || !std::isfinite(kappa)) return FL_ERR_ARG_OUTOFRANGE;
  fl_poisson *h  = new fl_poisson();
  const int   rc = poisson_init(h, grid, bc, kappa, decomp, device);
  if (rc) {
    fl_poisson_destroy(h);  // releases whatever was created before the failure
    return rc;
  }
  *out = h;
  return FL_SUCCESS;
}

extern "C" int fl_poisson_destroy(fl_poisson *h)
{
  if (!h) return FL_SUCCESS;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  fl_mg_destroy(h);
  h->comm.destroy();
  for (SmoothSeq &q : h->smooth_seq)
    if (q.dev) (void)hipFree(q.dev);
  for (void *p : h->tables) (void)hipFree(p);
  for (void *p : h->vec_bases) (void)hipFree(p);
  fl_vmm_destroy(h);
  for (double *p : {h->partial, h->sums, h->hist})
    if (p) (void)hipFree(p);
  for (int b = 0; b < 6; ++b) {
    if (h->fsend[b]) (void)hipFree(h->fsend[b]);
    if (h->frecv[b]) (void)hipFree(h->frecv[b]);
    if (h->xsend[b]) (void)hipFree(h->xsend[b]);
    if (h->xrecv[b]) (void)hipFree(h->xrecv[b]);
  }
  for (int d = 0; d < 3; ++d) {
    if (h->hiface[d]) (void)hipFree(h->hiface[d]);
    if (h->loface_send[d]) (void)hipFree(h->loface_send[d]);
  }
  if (h->scal) (void)hipFree(h->scal);
  if (h->tickets) (void)hipFree(h->tickets);
  if (h->scal_host) (void)hipHostFree(h->scal_host);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->ev_packed) (void)hipEventDestroy(h->ev_packed);
  if (h->ev_ghosts) (void)hipEventDestroy(h->ev_ghosts);
  if (h->ev_upload) (void)hipEventDestroy(h->ev_upload);
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return FL_SUCCESS;
}

extern "C" int fl_poisson_set_stream(fl_poisson *h, void *hip_stream)
{
  if (!h) return FL_ERR_ARG_NULL;
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  fl_mg_set_stream(h);
  return FL_SUCCESS;
}

extern "C" int fl_poisson_synchronize(fl_poisson *h)
{
  if (!h) return FL_ERR_ARG_NULL;
  FL_HIP(hipStreamSynchronize(h->stream));
  return FL_SUCCESS;
}

// All ranks of the handle's communicator have reached this call (and the handle's stream is idle) when it returns: a
// one-double sum over the ranks.  The host mirror sequences file output of the ranks with it (MPI_Barrier in the reference).
extern "C" int fl_poisson_barrier(fl_poisson *h)
{
  if (!h) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  if (h->multi) {
    FL_HIP(hipMemsetAsync(h->sums, 0, sizeof(double) * NSLOT, h->stream));
    FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  }
  FL_HIP(hipStreamSynchronize(h->stream));
  return FL_SUCCESS;
}

// max of one host number over the ranks of the handle's communicator, through the sum all-reduce it has: every rank writes its value into its
// own slot of a zeroed array (at most NSLOT ranks).  A host wait; for set-up quantities only (bounds, estimates).
int fl_allreduce_max(fl_poisson *h, double *v)
{
  if (!h->multi) return 0;
  const int nr = h->comm.nranks;
  if (nr > NSLOT) return FL_ERR_SUP;
  double host[NSLOT] = {0., 0., 0., 0., 0., 0., 0., 0.};
  host[h->comm.rank] = *v;
  FL_HIP(hipMemcpyAsync(h->sums, host, sizeof(double) * NSLOT, hipMemcpyHostToDevice, h->stream));
  FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  FL_HIP(hipMemcpyAsync(host, h->sums, sizeof(double) * NSLOT, hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  double mx = host[0];
  for (int a = 1; a < nr; ++a) mx = std::max(mx, host[a]);
  *v = mx;
  return 0;
}

int fl_allreduce_sum(fl_poisson *h, double *v)
{
  if (!h->multi) return 0;
  double host[NSLOT] = {*v, 0., 0., 0., 0., 0., 0., 0.};
  FL_HIP(hipMemcpyAsync(h->sums, host, sizeof(double) * NSLOT, hipMemcpyHostToDevice, h->stream));
  FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  FL_HIP(hipMemcpyAsync(host, h->sums, sizeof(double) * NSLOT, hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  *v = host[0];
  return 0;
}

extern "C" int fl_poisson_allreduce_sum(fl_poisson *h, double *host_vals, int n)
{
  if (!h || !host_vals) return FL_ERR_ARG_NULL;
  if (n < 0 || n > NSLOT) return FL_ERR_ARG_OUTOFRANGE;
  if (!h->multi || n == 0) return FL_SUCCESS;
  FL_HIP(hipSetDevice(h->device));
  double host[NSLOT] = {0., 0., 0., 0., 0., 0., 0., 0.};
  std::memcpy(host, host_vals, sizeof(double) * (size_t)n);
  FL_HIP(hipMemcpyAsync(h->sums, host, sizeof(double) * NSLOT, hipMemcpyHostToDevice, h->stream));
  FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  FL_HIP(hipMemcpyAsync(host, h->sums, sizeof(double) * NSLOT, hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  std::memcpy(host_vals, host, sizeof(double) * (size_t)n);
  return FL_SUCCESS;
}

extern "C" int fl_poisson_sizes(const fl_poisson *h, int64_t out[4])
{
  if (!h || !out) return FL_ERR_ARG_NULL;
  out[0] = h->ncell;
  out[1] = h->nface[0];
  out[2] = h->nface[1];
  out[3] = h->nface[2];
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ placement
// Kernels that stream five or six gigabyte-sized vectors in lock step (k_cg_A: reads r, p, x, writes p', q, x) run in one
// of two modes on MI355X, 1.10 ms or 1.27 ms per launch at 512^3.  Measured cause (profiles/r02_placement.md): the mode is
// a property of WHERE IN PHYSICAL MEMORY the vectors live relative to each other.  Vectors that sit in one physically
// contiguous block of HBM -- what back-to-back hipMallocs, and any layout inside the first 16 GiB of one large allocation,
// produce -- are slow at every spacing and alignment; as soon as two or three of the five come from a different block the
// kernel runs 13 % faster (a sliding window of five packed vectors inside one 96 GiB allocation is slow everywhere except
// where it straddles the seams between the driver's blocks, at 16 GiB and 64 GiB into the allocation).
// So placement is no lottery: ONE arena large enough to contain a seam is allocated, a window of five packed vectors slides
// through it (k_cg_A itself is the probe, ~20 positions of a few ms), and the solver vectors are carved out where the window
// was fastest; the vectors outside the window come alternately from the arena's two sides.  Done once per handle, by the
// first fl_ensure_vec of a large handle (tuning knob "placement", default 1) or explicitly by fl_poisson_tune_placement.

// ------------------------------------------------------------------------------------------------ knobs (fl_knobs.h)
namespace fl {
namespace {
struct KnobEntry {
  const char      *name;
  int              dflt;
  std::atomic<int> v;
};
KnobEntry g_knobs[K_COUNT + 1] = {
#define X(n, d) {#n, (d), {(d)}},
    FL_PUBLIC_KNOBS(X) FL_VARIANT_KNOBS(X)
#undef X
        {nullptr, 0, {0}}};
// the one place where the library reads its environment: FLUCA_<NAME> gives a knob its initial value
void knob_table_init()
{
  static std::once_flag once;
  std::call_once(once, []() {
    for (int k = 0; k < K_COUNT; ++k) {
      std::string env = "FLUCA_";
      for (const char *c = g_knobs[k].name; *c; ++c) env.push_back((char)std::toupper((unsigned char)*c));
      if (const char *e = std::getenv(env.c_str())) g_knobs[k].v.store(std::atoi(e), std::memory_order_relaxed);
    }
  });
}
}  // namespace
int knob(Knob k)
{
  knob_table_init();
  return g_knobs[k].v.load(std::memory_order_relaxed);
}
void knob_set(Knob k, int v)
{
  knob_table_init();
  g_knobs[k].v.store(v, std::memory_order_relaxed);
}
int knob_find(const char *name)
{
  for (int k = 0; k < K_COUNT; ++k)
    if (std::strcmp(g_knobs[k].name, name) == 0) return k;
  return -1;
}
const char *knob_name(int k) { return k >= 0 && k < K_COUNT ? g_knobs[k].name : nullptr; }
#ifdef FL_KBENCH_VARIANTS
const char *variant_env(const char *name) { return std::getenv(name); }
#endif
}  // namespace fl

extern "C" int fl_tuning_set(const char *name, int value)
{
  if (!name) return FL_ERR_ARG_NULL;
  const int k = knob_find(name);
  if (k < 0) return FL_ERR_ARG_WRONG;
  knob_set((Knob)k, value);
  return FL_SUCCESS;
}
extern "C" int fl_tuning_get(const char *name, int *value)
{
  if (!name || !value) return FL_ERR_ARG_NULL;
  const int k = knob_find(name);
  if (k < 0) return FL_ERR_ARG_WRONG;
  *value = knob((Knob)k);
  return FL_SUCCESS;
}
// An arena whose physical memory is a row of separately created chunks mapped into one reserved address range (HIP virtual memory
// management).  The search below slides its window through it like through a plain allocation; afterwards the chunks the chosen window
// does not touch are unmapped and released, so the handle keeps the window's own physical memory -- the place the probe measured --
// and nothing else.
struct VmmArena {
  char                                      *va = nullptr;
  size_t                                     size = 0, chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
  std::vector<char>                          live;
  size_t bytes_live() const
  {
    size_t n = 0;
    for (char c : live) n += c ? chunk : 0;
    return n;
  }
  void release_outside(size_t lo, size_t hi)  // keeps every chunk that overlaps [lo, hi)
  {
    for (size_t c = 0; c < handles.size(); ++c) {
      const size_t b = c * chunk, e = b + chunk;
      if (live[c] && (e <= lo || b >= hi)) {
        (void)hipMemUnmap(va + b, chunk);
        (void)hipMemRelease(handles[c]);
        live[c] = 0;
      }
    }
  }
  ~VmmArena()
  {
    if (!va) return;
    release_outside(0, 0);
    (void)hipMemAddressFree(va, size);
  }
};
static VmmArena *vmm_arena_create(int device, size_t want, size_t chunk_hint)
{
  hipMemAllocationProp prop = {};
  prop.type                 = hipMemAllocationTypePinned;
  prop.location.type        = hipMemLocationTypeDevice;
  prop.location.id          = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
    (void)hipGetLastError();
    return nullptr;
  }
  VmmArena *A = new (std::nothrow) VmmArena;
  if (!A) return nullptr;
  A->chunk = ((chunk_hint + gran - 1) / gran) * gran;
  const size_t n = (want + A->chunk - 1) / A->chunk;
  A->size = n * A->chunk;
  void *va = nullptr;
  if (hipMemAddressReserve(&va, A->size, 0, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    delete A;
    return nullptr;
  }
  A->va = (char *)va;
  hipMemAccessDesc acc = {};
  acc.location         = prop.location;
  acc.flags            = hipMemAccessFlagsProtReadWrite;
  size_t accessible = 0;
  for (size_t c = 0; c < n; ++c) {
    hipMemGenericAllocationHandle_t hd;
    if (hipMemCreate(&hd, A->chunk, &prop, 0) != hipSuccess) break;
    if (hipMemMap(A->va + c * A->chunk, A->chunk, 0, hd, 0) != hipSuccess) {
      (void)hipMemRelease(hd);
      break;
    }
    A->handles.push_back(hd);
    A->live.push_back(1);
    if (hipMemSetAccess(A->va + c * A->chunk, A->chunk, &acc, 1) != hipSuccess) break;
    ++accessible;
  }
  if (accessible != n) {  // a chunk that could not be created, mapped or made accessible: the arena's destructor unmaps and releases what exists
    (void)hipGetLastError();
    delete A;
    return nullptr;
  }
  return A;
}
void fl_vmm_destroy(fl_poisson *h)
{
  if (h->vmm) delete h->vmm;
  h->vmm = nullptr;
}

namespace {
constexpr size_t PL_MIN_VEC   = (size_t)256 << 20;  // smaller vectors: nothing to gain, plain allocations
constexpr size_t PL_SEAM      = (size_t)16 << 30;   // where the first seam of a fresh allocation has been found on every box
constexpr int    PL_WIN       = 5;                   // r, P0, P1, q, xp
constexpr int    PL_SIDE      = 3;                   // pool slots on either side of the window

int place_vectors(fl_poisson *h)
{
  if (h->placed || h->nv_il > 1) return 0;
  h->placed = true;  // whatever happens below is final for this handle
  hipStream_t  s    = h->stream;
  const size_t vecb = ((sizeof(double) * h->padlen + ((size_t)2 << 20) - 1) / ((size_t)2 << 20)) * ((size_t)2 << 20);
  const int    nslot = PL_WIN + 2 * PL_SIDE;
  if (vecb < PL_MIN_VEC) return 0;  // small vectors: the kernels are not bandwidth-bound enough to notice; plain allocations
  PlanA plan = plan_cg_A(h->g, 0, 0);
  plan.probe = 1;  // launches k_cg_A_probe / k_cg_Bq_probe: identical code, separate names in profiles
  FL_CHK(fl_ensure_partials(h, plan.nblocks));
  struct Held {  // two scalar blocks (direction buffer parity 0 and 1) and the arenas: whatever is not handed to the handle is released
    KspScal                *p = nullptr;
    std::vector<void *>     arenas;
    std::vector<VmmArena *> vmm;  // parallel to arenas: non-null where the arena is chunk-mapped virtual memory
    ~Held()
    {
      if (p) (void)hipFree(p);
      for (size_t a = 0; a < arenas.size(); ++a) {
        if (vmm[a]) delete vmm[a];
        else if (arenas[a]) (void)hipFree(arenas[a]);
      }
    }
  } sc;
  FL_HIP(hipMalloc((void **)&sc.p, 2 * sizeof(KspScal)));
  {
    KspScal S2[2];
    std::memset(S2, 0, sizeof(S2));
    for (int a = 0; a < 2; ++a) {
      S2[a].beta = 0.5; S2[a].alpha = 1e-3; S2[a].alpha_old = 1e-3; S2[a].zshift = 1e-4; S2[a].ncell_global = (double)h->ncell; S2[a].maxit = 1 << 30; S2[a].cur = a;
    }
    FL_HIP(hipMemcpy(sc.p, S2, sizeof(S2), hipMemcpyHostToDevice));
  }
  const int verbose = knob(K_placement_verbose);
  // probe = the pair the solver runs: k_cg_A (r, p -> p') and the odd-iteration k_cg_Bq (p', p_old, r, x -> r, x: every window vector but q)
  auto probe = [&](void *arena, size_t b, double *ms_out) -> int {
    auto vec = [&](int k) { return (double *)((char *)arena + b + (size_t)k * vecb); };
    auto run = [&](int reps) {
      for (int r = 0; r < reps; ++r)
        for (int par = 0; par < 2; ++par) {
          launch_cg_A(s, h->g, true, plan, vec(0), vec(1), vec(2), vec(3), vec(4), sc.p + par, h->partial, nullptr, nullptr, 0);
          launch_cg_Bq(s, h->g, true, plan, 2, vec(1), vec(2), vec(0), vec(4), sc.p + par, h->partial, h->partial_stride, nullptr, nullptr, 0);
        }
    };
    // one untimed pair (TLB / L2 warm-up of the new position), then one timed repetition = two pairs (both direction-buffer parities)
    launch_cg_A(s, h->g, true, plan, vec(0), vec(1), vec(2), vec(3), vec(4), sc.p, h->partial, nullptr, nullptr, 0);
    launch_cg_Bq(s, h->g, true, plan, 2, vec(1), vec(2), vec(0), vec(4), sc.p, h->partial, h->partial_stride, nullptr, nullptr, 0);
    FL_HIP(hipEventRecord(h->ev0, s));
    run(1);
    FL_HIP(hipEventRecord(h->ev1, s));
    FL_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_out = ms / 2.;
    return 0;
  };
  // Where the seams of an allocation lie depends on what the device's memory manager handed out before (on most fresh boxes one is
  // found 16 GiB in; on some the whole arena answers flat).  A flat arena is kept allocated -- so that the next one comes from other
  // physical memory -- and the search repeated, at most PL_ARENAS times; the losers are freed at the end.
  constexpr int PL_ARENAS = 3;
  const int use_vmm = knob(K_placement_vmm);  // 1 (default): chunk-mapped arenas, everything but the chosen window is released
  void  *arena = nullptr;
  size_t want = 0, best = 0;
  double first_ms = 0., best_ms = 0.;
  for (int attempt = 0; attempt < PL_ARENAS; ++attempt) {
    size_t freeb = 0, total = 0;
    if (hipMemGetInfo(&freeb, &total) != hipSuccess) break;
    size_t w = ((PL_SEAM + (size_t)(PL_WIN + PL_SIDE) * vecb + ((size_t)1 << 30) - 1) >> 30) << 30;
    const size_t reserve = (size_t)16 << 30;
    if (freeb < w + reserve) {
      if (attempt > 0) break;  // further arenas only while memory is plentiful
      w = freeb > reserve + (size_t)nslot * vecb ? ((freeb - reserve) >> 30) << 30 : 0;
    }
    if (w < (size_t)nslot * vecb) break;  // not enough memory for an arena
    void     *a  = nullptr;
    VmmArena *va = use_vmm ? vmm_arena_create(h->device, w, (size_t)256 << 20) : nullptr;
    if (va) a = va->va;
    else if (hipMalloc(&a, w) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    sc.arenas.push_back(a);
    sc.vmm.push_back(va);
    FL_HIP(hipMemsetAsync(a, 0, w, s));
    const size_t lo = (size_t)PL_SIDE * vecb, hi = w - (size_t)(PL_WIN + PL_SIDE) * vecb;
    // coarse pass in steps of one vector (the fast stretch before a seam is four vectors long), then the two half steps next to the best
    const size_t step = std::max(vecb, (((hi - lo) / 32) >> 21) << 21);
    size_t       abest = lo;
    double       afirst = 0., abest_ms = 0.;
    int          nprobe = 0;
    auto         try_at = [&](size_t b) -> int {
      double ms = 0.;
      FL_CHK(probe(a, b, &ms));
      if (verbose) std::fprintf(stderr, "[fluca placement] arena %d, window at %.2f GiB: %.4f ms\n", attempt, (double)b / (double)((size_t)1 << 30), ms);
      if (nprobe++ == 0) afirst = abest_ms = ms;
      if (ms < abest_ms) {
        abest_ms = ms;
        abest    = b;
      }
      return 0;
    };
    for (size_t b = lo; b <= hi; b += step) FL_CHK(try_at(b));
    if (abest_ms <= 0.985 * afirst) {
      const size_t c = abest, half = ((step / 2) >> 21) << 21;
      if (c >= lo + half) FL_CHK(try_at(c - half));
      if (c + half <= hi) FL_CHK(try_at(c + half));
    }
    if (attempt == 0) first_ms = afirst;
    if (!arena || abest_ms < best_ms) {
      arena   = a;
      want    = w;
      best    = abest;
      best_ms = abest_ms;
    }
    const char  *te = variant_env("FLUCA_PLACEMENT_THRESH");  // experiments: 0 walks through all PL_ARENAS arenas
    const double thresh = te ? std::atof(te) : 0.97;
    if (best_ms <= thresh * first_ms) break;  // a seam was found
  }
  if (!arena) return 0;  // no memory for an arena: plain allocations
  // A chunk-mapped arena gives back everything but the chunks under the chosen window: the handle keeps five vectors (plus at most two
  // chunks of 256 MiB of slack), on the very physical memory the probe measured.  (Round 2 kept the whole arena, 16 GiB + 8 vectors;
  // giving it back and allocating "the same place" again -- a filler of the window's offset, then the window -- was tried and does not
  // land on the same physical memory: probe 1.729 ms where the search had found 1.636, profiles/r03_placement.txt.)
  {
    VmmArena *chosen = nullptr;
    for (size_t a = 0; a < sc.arenas.size(); ++a)
      if (sc.arenas[a] == arena) chosen = sc.vmm[a];
    if (chosen) {
      const size_t winb = (size_t)PL_WIN * vecb;
      for (size_t a = 0; a < sc.arenas.size(); ++a)
        if (sc.vmm[a] == chosen) {
          sc.vmm[a]    = nullptr;
          sc.arenas[a] = nullptr;
        }
      // `chosen` left the search's guard above: until the handle owns it, every early return below must give it back
      struct Owner {
        VmmArena *a;
        ~Owner() { delete a; }
      } own{chosen};
      chosen->release_outside(best, best + winb);
      FL_HIP(hipMemsetAsync((char *)arena + best, 0, winb, s));
      double again = 0.;
      const int prc = probe(arena, best, &again);
      if (prc != 0) return prc;
      if (verbose) std::fprintf(stderr, "[fluca placement] window at %.2f GiB kept (%.2f GiB live of %.2f), probe again %.4f ms (search %.4f, first %.4f)\n", (double)best / (double)((size_t)1 << 30), (double)chosen->bytes_live() / (double)((size_t)1 << 30), (double)chosen->size / (double)((size_t)1 << 30), again, best_ms, first_ms);
      FL_HIP(hipMemsetAsync((char *)arena + best, 0, winb, s));
      FL_HIP(hipStreamSynchronize(s));
      own.a          = nullptr;
      h->vmm         = chosen;
      h->arena       = nullptr;  // no side pools: every other vector is a plain allocation
      h->arena_bytes = chosen->bytes_live();
      h->vec_bytes += chosen->bytes_live();
      double **wv[PL_WIN] = {&h->r, &h->P0, &h->P1, &h->q, &h->xp};
      for (int k = 0; k < PL_WIN; ++k) *wv[k] = (double *)((char *)arena + best + (size_t)k * vecb);
      h->nvec += PL_WIN;
      h->placed_ms[0] = first_ms;
      h->placed_ms[1] = again;
      h->placed_at    = (double)best / (double)((size_t)1 << 30);
      return 0;
    }
  }
  for (void *&a : sc.arenas)
    if (a == arena) a = nullptr;  // this one goes to the handle
  // the probes wrote into the arena: ghost layers of fresh solver vectors are zero by contract
  FL_HIP(hipMemsetAsync(arena, 0, want, s));
  FL_HIP(hipStreamSynchronize(s));
  h->arena       = arena;
  h->arena_bytes = want;
  h->vec_bases.push_back(arena);
  h->vec_bytes += want;
  double **win[PL_WIN] = {&h->r, &h->P0, &h->P1, &h->q, &h->xp};
  for (int k = 0; k < PL_WIN; ++k) *win[k] = (double *)((char *)arena + best + (size_t)k * vecb);
  h->pool_next[0] = (char *)arena + best - (size_t)PL_SIDE * vecb;
  h->pool_end[0]  = (char *)arena + best;
  h->pool_next[1] = (char *)arena + best + (size_t)PL_WIN * vecb;
  h->pool_end[1]  = h->pool_next[1] + (size_t)PL_SIDE * vecb;
  h->pool_vec     = vecb;
  h->nvec += PL_WIN;
  h->placed_ms[0] = first_ms;
  h->placed_ms[1] = best_ms;
  h->placed_at    = (double)best / (double)((size_t)1 << 30);
  return 0;
}
}  // namespace

// a padded vector from the arena's side pools (alternating sides), or nullptr when there is no arena / no slot left
static double *pool_take(fl_poisson *h)
{
  if (!h->arena) return nullptr;
  for (int t = 0; t < 2; ++t) {
    const int side = (h->pool_flip + t) & 1;
    if (h->pool_next[side] + h->pool_vec <= h->pool_end[side]) {
      double *v = (double *)h->pool_next[side];
      h->pool_next[side] += h->pool_vec;
      h->pool_flip = side ^ 1;
      return v;
    }
  }
  return nullptr;
}

// Explicit form of the placement step (idempotent; max_tries is kept for source compatibility and only has to be >= 1).
// probe_ms_out: {k_cg_A probe time with the window at the start of the arena (all vectors in one physical block: what plain
// back-to-back allocations give), probe time at the chosen position}; {0, 0} when the handle is too small to be placed.
extern "C" int fl_poisson_tune_placement(fl_poisson *h, int max_tries, double probe_ms_out[2])
{
  if (!h) return FL_ERR_ARG_NULL;
  if (max_tries < 1) return FL_ERR_ARG_OUTOFRANGE;
  FL_HIP(hipSetDevice(h->device));
  if (!h->placed && h->nv_il == 1) {
    FL_HIP(hipStreamSynchronize(h->stream));
    // vectors that exist already (a solve ran before this call) are dropped: every solve re-creates what it needs
    fl_mg_destroy(h);
    for (void *p : h->vec_bases) (void)hipFree(p);
    fl_vmm_destroy(h);
    h->vec_bases.clear();
    h->vec_bytes = 0;
    h->nvec = 0;
    h->slab = nullptr;
    for (double **v : {&h->r, &h->P0, &h->P1, &h->q, &h->xp, &h->w0, &h->w1, &h->w2, &h->cd1, &h->rb}) *v = nullptr;
    FL_CHK(place_vectors(h));
  }
  if (probe_ms_out) {
    probe_ms_out[0] = h->placed_ms[0];
    probe_ms_out[1] = h->placed_ms[1];
  }
  return FL_SUCCESS;
}

// bytes of device memory the handle holds for its padded solver vectors (placement window or arena included)
extern "C" int fl_poisson_vector_bytes(fl_poisson *h, int64_t *bytes_out)
{
  if (!h || !bytes_out) return FL_ERR_ARG_NULL;
  *bytes_out = (int64_t)h->vec_bytes;
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ workspace / ghosts

// Padded solver vectors.  Each one starts at a different offset inside its allocation (multiples of FLUCA_SKEW bytes,
// default 0): the hot kernels touch the same logical index of up to six vectors at the same time, and identical
// low address bits put those six streams on the same HBM channel.
int fl_ensure_vec(fl_poisson *h, double **v)
{
  if (*v) return 0;
  if (h->nv_il == 1 && !h->placed && h->nvec == 0 && knob(K_placement) > 0 && sizeof(double) * h->padlen >= PL_MIN_VEC) {
    // carves r, P0, P1, q, xp out of one allocation (see "placement" above).  A failure in there (memory short, a probe launch refused)
    // is no reason to fail the caller's solve: whatever the search held is released and the vectors become plain allocations.
    if (place_vectors(h) != 0) {
      (void)hipGetLastError();
      for (double **w : {&h->r, &h->P0, &h->P1, &h->q, &h->xp}) *w = nullptr;
    }
    if (*v) return 0;
  }
  if (double *p = pool_take(h)) {
    *v = p;
    h->nvec++;
    return 0;
  }
  if (h->nv_il > 1) {
    // interleaved: one slab, vector k starts k*sx0 doubles into it
    if (!h->slab) {
      FL_CHK(fl_dev_alloc(h, &h->slab, sizeof(double) * h->padlen, true));
      h->vec_bases.push_back(h->slab);
    }
    if (h->nvec >= h->nv_il) return FL_ERR_MEM;
    *v = (double *)h->slab + (size_t)h->sx0 * (size_t)h->nvec++;
    return 0;
  }
  const long gap = ((long)FL_VARIANT(gap, FL_DEFAULT_GAP) / 128) * 128;
  constexpr int NSLOTS = 8;
  const size_t  slot   = ((sizeof(double) * h->padlen + 127) / 128) * 128 + (size_t)gap;
  const bool use_slab = FL_VARIANT(slab, 0) != 0;  // default: one hipMalloc per vector (placement)
  if (!use_slab) {
    void *base = nullptr;
    FL_CHK(fl_dev_alloc(h, &base, slot, true));
    h->vec_bases.push_back(base);
    h->vec_bytes += slot;
    h->nvec++;
    *v = (double *)base;
    return 0;
  }
  if (!h->slab) {
    FL_CHK(fl_dev_alloc(h, &h->slab, slot * NSLOTS, true));
    h->vec_bases.push_back(h->slab);
  }
  if (h->nvec >= NSLOTS) return FL_ERR_MEM;
  *v = (double *)((char *)h->slab + slot * (size_t)h->nvec++);
  return 0;
}

// zero a padded vector (ghosts included) on the handle's stream
int fl_zero_vec(fl_poisson *h, double *v)
{
  if (h->nv_il <= 1) {
    FL_HIP(hipMemsetAsync(v, 0, sizeof(double) * h->padlen, h->stream));
    return 0;
  }
  const size_t rows = (size_t)(h->g.ny + 2) * (size_t)(h->g.nz + 2);
  FL_HIP(hipMemset2DAsync(v, sizeof(double) * (size_t)h->g.sx, 0, sizeof(double) * (size_t)h->sx0, rows, h->stream));
  return 0;
}

int fl_ensure_hist(fl_poisson *h, int nhist)
{
  if (h->hist_cap >= nhist) return 0;
  if (h->hist) {
    FL_HIP(hipStreamSynchronize(h->stream));
    FL_HIP(hipFree(h->hist));
    h->hist = nullptr;
  }
  FL_CHK(fl_dev_alloc(h, (void **)&h->hist, sizeof(double) * nhist, true));
  h->hist_cap = nhist;
  return 0;
}

int fl_ensure_partials(fl_poisson *h, int nblocks)
{
  const int want = std::max(nblocks, (int)MAX_PARTIAL_BLOCKS);
  if (h->partial && h->partial_stride >= want) return 0;
  if (h->partial) {
    FL_HIP(hipStreamSynchronize(h->stream));
    FL_HIP(hipFree(h->partial));
    h->partial = nullptr;
  }
  FL_CHK(fl_dev_alloc(h, (void **)&h->partial, sizeof(double) * (size_t)want * NSLOT, true));
  h->partial_stride = want;
  return 0;
}

static int ensure_facebufs(fl_poisson *h)
{
  for (int b = 0; b < 6; ++b) {
    if (h->nbr[b] < 0 || h->wrap_local[b / 2] || h->fsend[b]) continue;
    const size_t n = (size_t)plane_size(h, b / 2);
    FL_CHK(fl_dev_alloc(h, (void **)&h->fsend[b], sizeof(double) * n, true));
    FL_CHK(fl_dev_alloc(h, (void **)&h->frecv[b], sizeof(double) * n, true));
  }
  return 0;
}

// the messages of one ghost exchange (fl_halo_plan + the self-messages of the loopback mode) and the buffers they use
static int halo_messages(fl_poisson *h, std::vector<Msg> &msgs, double *sbuf[6], double *rbuf[6])
{
  FL_CHK(ensure_facebufs(h));
  int periodic[3];
  for (int d = 0; d < 3; ++d) periodic[d] = h->ax[d].periodic;
  fl_halo_msg plan[12];
  int         np = fl_halo_plan(&h->dec, periodic, plan);
  if (h->loopback)  // one rank, periodic axes: both faces go to this very rank, same order as the two-rank periodic case
    for (int ax = 0; ax < 3; ++ax)
      if (periodic[ax]) {
        plan[np++] = {0, 2 * ax + 1, 2 * ax, 2 * ax + 1, 2 * ax + 1};
        plan[np++] = {0, 2 * ax, 2 * ax + 1, 2 * ax, 2 * ax};
      }
  for (int b = 0; b < 6; ++b) sbuf[b] = rbuf[b] = nullptr;
  for (int a = 0; a < np; ++a) {
    const int sb = plan[a].send_boundary, rb = plan[a].recv_boundary;
    sbuf[sb] = h->fsend[sb];
    rbuf[rb] = h->frecv[rb];
    msgs.push_back({plan[a].peer, h->fsend[sb], h->frecv[rb], (int64_t)plane_size(h, sb / 2), plan[a].sendtag, plan[a].recvtag});
  }
  return 0;
}

// ghosts of a padded vector: local periodic images + neighbour ranks' boundary cells (DMGlobalToLocal of the reference)
int fl_fill_ghosts(fl_poisson *h, double *v)
{
  const GridP &g = h->g;
  for (int d = 0; d < 3; ++d)
    if (h->wrap_local[d]) launch_wrap(h->stream, g, v, d);
  if (!h->multi) return 0;
  if (h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  std::vector<Msg> msgs;
  double          *sbuf[6], *rbuf[6];
  FL_CHK(halo_messages(h, msgs, sbuf, rbuf));
  if (!msgs.empty()) launch_pack_faces(h->stream, g, v, sbuf);    // all boundary layers in one launch
  FL_CHK(h->comm.exchange(h->stream, msgs));
  if (!msgs.empty()) launch_unpack_faces(h->stream, g, v, rbuf);  // all ghost layers in one launch
  return 0;
}

// staging buffers of the extended-face exchanges: two layers of the largest face with two cells of extension on every side
static int ensure_xbufs(fl_poisson *h, int sb, int rb)
{
  const GridP &g = h->g;
  const size_t cap = 2 * (size_t)(std::max(g.nx, g.ny) + 4) * (size_t)(std::max(g.ny, g.nz) + 4);
  for (int bnd : {sb, rb}) {
    if (bnd < 0 || h->xsend[bnd]) continue;
    FL_CHK(fl_dev_alloc(h, (void **)&h->xsend[bnd], sizeof(double) * cap, true));
    FL_CHK(fl_dev_alloc(h, (void **)&h->xrecv[bnd], sizeof(double) * cap, true));
  }
  h->xcap = cap;
  return 0;
}

// Ghost layers INCLUDING the edge and corner cells (what a 27-point footprint reads: the tri-linear prolongation of the multigrid cycle):
// the axes are handled one after the other, and the face exchanged / wrapped along axis d spans the ghost layers the axes before it
// have already filled, so that an edge cell arrives in two hops and a corner cell in three -- the reference's DMStag would do the same
// with DMSTAG_STENCIL_BOX.  Three exchanges instead of one; used on coarse correction vectors only.
int fl_fill_ghosts_full(fl_poisson *h, double *v)
{
  const GridP &g = h->g;
  if (h->multi && h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  int periodic[3];
  for (int d = 0; d < 3; ++d) periodic[d] = h->ax[d].periodic;
  fl_halo_msg plan[12];
  int         np = h->multi ? fl_halo_plan(&h->dec, periodic, plan) : 0;
  if (h->multi && h->loopback)
    for (int ax = 0; ax < 3; ++ax)
      if (periodic[ax]) {
        plan[np++] = {0, 2 * ax + 1, 2 * ax, 2 * ax + 1, 2 * ax + 1};
        plan[np++] = {0, 2 * ax, 2 * ax + 1, 2 * ax, 2 * ax};
      }
  for (int d = 0; d < 3; ++d) {
    const int ea = d >= 1 ? 1 : 0, eb = d >= 2 ? 1 : 0;  // in-face directions: (y, z), (x, z), (x, y)
    if (h->wrap_local[d]) {
      launch_face_ext(h->stream, g, v, nullptr, d, 0, ea, eb, 0);
      continue;
    }
    if (!h->multi) continue;
    const int64_t    cnt = (int64_t)((d == 0 ? g.ny : g.nx) + 2 * ea) * ((d == 2 ? g.ny : g.nz) + 2 * eb);
    std::vector<Msg> msgs;
    bool             recv_side[2] = {false, false};
    for (int a = 0; a < np; ++a) {
      const int sb = plan[a].send_boundary, rb = plan[a].recv_boundary;
      if (sb / 2 != d) continue;
      FL_CHK(ensure_xbufs(h, sb, rb));
      launch_face_ext(h->stream, g, v, h->xsend[sb], d, sb % 2, ea, eb, 1);
      msgs.push_back({plan[a].peer, h->xsend[sb], h->xrecv[rb], cnt, plan[a].sendtag + 64, plan[a].recvtag + 64});
      recv_side[rb % 2] = true;
    }
    FL_CHK(h->comm.exchange(h->stream, msgs));
    for (int side = 0; side < 2; ++side)
      if (recv_side[side]) launch_face_ext(h->stream, g, v, h->xrecv[2 * d + side], d, side, ea, eb, 2);
  }
  return 0;
}

// TWO ghost layers across every boundary behind which a neighbouring rank sits, edge and corner cells of that shell included (the reference's
// DMStag has stencil width 1, cart.c:66: this layer is a build-side extension): what two fused stencil steps read (k_cheb2: x at distance two
// along an axis and at the diagonal neighbours in a plane).  Dimension by dimension like fl_fill_ghosts_full: the two layers sent along axis d
// span the ghost layers the axes before it have received.  Axes held by one rank are left alone (a periodic one wraps inside the block, and
// the kernel wraps its indices there; behind a wall there is nothing).  Needs the wide layout (h->gw == 2).
int fl_fill_ghosts_deep(fl_poisson *h, double *v)
{
  const GridP &g = h->g;
  if (!h->multi) return 0;
  if (h->gw < 2) return FL_ERR_ARG_WRONGSTATE;
  if (h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  int periodic[3];
  for (int d = 0; d < 3; ++d) periodic[d] = h->ax[d].periodic;
  fl_halo_msg plan[12];
  const int   np = fl_halo_plan(&h->dec, periodic, plan);
  int         ext[3] = {0, 0, 0};  // ghost layers axis d holds once it has been handled
  for (int d = 0; d < 3; ++d) {
    const int a1 = d == 0 ? 1 : 0, a2 = d == 2 ? 1 : 2;  // in-face directions: (y, z), (x, z), (x, y)
    const int ea = ext[a1], eb = ext[a2];
    const int64_t    cnt = 2 * (int64_t)((d == 0 ? g.ny : g.nx) + 2 * ea) * ((d == 2 ? g.ny : g.nz) + 2 * eb);
    std::vector<Msg> msgs;
    bool             recv_side[2] = {false, false};
    for (int a = 0; a < np; ++a) {
      const int sb = plan[a].send_boundary, rb = plan[a].recv_boundary;
      if (sb / 2 != d) continue;
      FL_CHK(ensure_xbufs(h, sb, rb));
      launch_face_ext_deep(h->stream, g, v, h->xsend[sb], d, sb % 2, ea, eb, 2, 1);
      msgs.push_back({plan[a].peer, h->xsend[sb], h->xrecv[rb], cnt, plan[a].sendtag + 128, plan[a].recvtag + 128});
      recv_side[rb % 2] = true;
    }
    if (msgs.empty()) continue;
    FL_CHK(h->comm.exchange(h->stream, msgs));
    for (int side = 0; side < 2; ++side)
      if (recv_side[side]) launch_face_ext_deep(h->stream, g, v, h->xrecv[2 * d + side], d, side, ea, eb, 2, 2);
    ext[d] = 2;
  }
  return 0;
}

// The CG iteration's ghost exchange of r, hidden behind k_cg_B (the DMGlobalToLocalBegin / ...End pair of the reference,
// fdapply.c:71, cnlinearcart3d.c:893-894).  begin: the boundary layers of r - alpha q are packed on the handle's stream BEFORE
// k_cg_B forms the new r; a second stream waits for the pack, runs the transfers and writes the ghost layers, which k_cg_B neither
// reads nor writes.  end: the handle's stream waits for the ghosts (and fills the locally wrapped axes) before k_cg_A needs them.
int fl_exchange_r_begin(fl_poisson *h, double *r, const double *q)  // q: valid on the boundary layers of the block at least (PlanA::qb)
{
  if (!h->multi) return 0;
  if (h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  if (!h->comm_stream) {
    FL_HIP(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    FL_HIP(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
    FL_HIP(hipEventCreateWithFlags(&h->ev_ghosts, hipEventDisableTiming));
  }
  std::vector<Msg> msgs;
  double          *sbuf[6], *rbuf[6];
  FL_CHK(halo_messages(h, msgs, sbuf, rbuf));
  if (!msgs.empty()) launch_pack_faces_rq(h->stream, h->g, r, q, h->scal, sbuf);
  FL_HIP(hipEventRecord(h->ev_packed, h->stream));
  FL_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_packed, 0));
  FL_CHK(h->comm.exchange(h->comm_stream, msgs));
  if (!msgs.empty()) launch_unpack_faces(h->comm_stream, h->g, r, rbuf);
  FL_HIP(hipEventRecord(h->ev_ghosts, h->comm_stream));
  return 0;
}

// The single-reduction CG's exchange behind its update kernel (MODE 10): the boundary layers of the new residual are packed from r, the kept S and W
// (k_pack_faces_sr) on the handle's stream, a second stream runs the transfers and writes the ghost layers of rn -- the buffer MODE 10 fills with the
// new residual's owned cells meanwhile.  fl_exchange_r_end(h, rn) closes it.
int fl_exchange_sr_begin(fl_poisson *h, const double *r, const double *sb, const double *W, double *rn)
{
  if (!h->multi) return 0;
  if (h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  if (!h->comm_stream) {
    FL_HIP(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    FL_HIP(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
    FL_HIP(hipEventCreateWithFlags(&h->ev_ghosts, hipEventDisableTiming));
  }
  std::vector<Msg> msgs;
  double          *sbuf[6], *rbuf[6];
  FL_CHK(halo_messages(h, msgs, sbuf, rbuf));
  if (!msgs.empty()) launch_pack_faces_sr(h->stream, h->g, r, sb, W, h->scal, sbuf);
  FL_HIP(hipEventRecord(h->ev_packed, h->stream));
  FL_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_packed, 0));
  FL_CHK(h->comm.exchange(h->comm_stream, msgs));
  if (!msgs.empty()) launch_unpack_faces(h->comm_stream, h->g, rn, rbuf);
  FL_HIP(hipEventRecord(h->ev_ghosts, h->comm_stream));
  return 0;
}
int fl_exchange_r_end(fl_poisson *h, double *r)
{
  for (int d = 0; d < 3; ++d)
    if (h->wrap_local[d]) launch_wrap(h->stream, h->g, r, d);
  if (!h->multi) return 0;
  FL_HIP(hipStreamWaitEvent(h->stream, h->ev_ghosts, 0));
  return 0;
}

bool fl_any_ghost_exchange(const fl_poisson *h) { return h->multi || h->wrap_local[0] || h->wrap_local[1] || h->wrap_local[2]; }

// ------------------------------------------------------------------------------------------------ operator entry points

extern "C" int fl_poisson_apply(fl_poisson *h, const double *x_dev, double *y_dev)
{
  if (!h || !x_dev || !y_dev) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(fl_ensure_vec(h, &h->w0));
  launch_pad_copy(h->stream, h->g, x_dev, h->w0);
  FL_CHK(fl_fill_ghosts(h, h->w0));
  FL_CHK(fl_apply_tiled(h, h->w0, y_dev, 1));
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_poisson_diagonal(fl_poisson *h, double *d_dev)
{
  if (!h || !d_dev) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  launch_diagonal(h->stream, h->g, d_dev);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_poisson_rhs(fl_poisson *h, const double *Vx, const double *Vy, const double *Vz, const double *contrhs, double *b)
{
  if (!h || !Vx || !Vy || !Vz || !b) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  const GridP  &g    = h->g;
  const double *V[3] = {Vx, Vy, Vz};
  // the high face of the last owned cell belongs to the next rank (or is the periodic image of face 0)
  std::vector<Msg> msgs;
  for (int d = 0; d < 3; ++d) {
    const int len = d == 0 ? g.nx : (d == 1 ? g.ny : g.nz);
    if (face_count(h, d) > len) continue;  // this rank owns its high boundary face
    const size_t n = (size_t)plane_size(h, d);
    if (!h->hiface[d]) FL_CHK(fl_dev_alloc(h, (void **)&h->hiface[d], sizeof(double) * n, true));
    if (h->wrap_local[d]) launch_face_plane0(h->stream, g, V[d], h->hiface[d], d);
    else if (h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  }
  if (h->multi) {
    // every rank with a low neighbour ships its first face plane there (send only); every rank with a high neighbour
    // receives that plane as the high face of its last cells (receive only)
    for (int d = 0; d < 3; ++d) {
      if (h->wrap_local[d]) continue;
      const int     lo = h->nbr[2 * d], hi = h->nbr[2 * d + 1];
      const int64_t n  = plane_size(h, d);
      if (lo >= 0) {
        if (!h->loface_send[d]) FL_CHK(fl_dev_alloc(h, (void **)&h->loface_send[d], sizeof(double) * n, true));
        launch_face_plane0(h->stream, g, V[d], h->loface_send[d], d);
      }
      if (lo >= 0 && lo == hi) {
        msgs.push_back({lo, h->loface_send[d], h->hiface[d], n, 6 + d, 6 + d});
      } else {
        if (lo >= 0) msgs.push_back({lo, h->loface_send[d], nullptr, n, 6 + d, 6 + d});
        if (hi >= 0) msgs.push_back({hi, nullptr, h->hiface[d], n, 6 + d, 6 + d});
      }
    }
    FL_CHK(h->comm.exchange(h->stream, msgs));
  }
  launch_rhs(h->stream, g, Vx, Vy, Vz, h->hiface[0], h->hiface[1], h->hiface[2], contrhs, b);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_poisson_project(fl_poisson *h, const double *p_dev, double *vx, double *vy, double *vz, double *Vx, double *Vy, double *Vz)
{
  if (!h || !p_dev) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  double *v[3] = {vx, vy, vz}, *V[3] = {Vx, Vy, Vz};
  // A/B runs.  0: one kernel per output array (round 1); 1: k_project_all (round 3); 2: k_project_six on the padded p; 3 (shipped): on the caller's p where
  // one rank holds the grid
  const int fused = FL_VARIANT(project_fused, 3);
  // all six arrays (PCApply_ABF's call), one rank: k_project_six reads the caller's p itself -- no padded copy, no ghost layers
  if (fused >= 3 && !h->multi && project_six_usable(h->g, p_dev, v, V)) {
    int per = 0;
    for (int d = 0; d < 3; ++d) per |= h->wrap_local[d] ? (1 << d) : 0;
    launch_project_six(h->stream, h->g, p_dev, true, per, v, V);
    FL_HIP(hipGetLastError());
    return FL_SUCCESS;
  }
  FL_CHK(fl_ensure_vec(h, &h->w0));
  launch_pad_copy(h->stream, h->g, p_dev, h->w0);
  FL_CHK(fl_fill_ghosts(h, h->w0));
  if (fused >= 2 && project_six_usable(h->g, nullptr, v, V)) {
    launch_project_six(h->stream, h->g, h->w0, false, 0, v, V);
    FL_HIP(hipGetLastError());
    return FL_SUCCESS;
  }
#ifdef FL_KBENCH_VARIANTS
  if (!fused) {
    for (int d = 0; d < 3; ++d) {
      if (v[d]) launch_project_cells(h->stream, h->g, h->w0, v[d], d);
      if (V[d]) launch_project_faces(h->stream, h->g, h->w0, V[d], d);
    }
    FL_HIP(hipGetLastError());
    return FL_SUCCESS;
  }
#endif
  launch_project_all(h->stream, h->g, h->w0, v, V);  // any subset of the six arrays, one pass over p
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_poisson_gst_bc(fl_poisson *h, int boundary, const double *pb_dev, double *V_dev)
{
  if (!h || !pb_dev || !V_dev) return FL_ERR_ARG_NULL;
  if (boundary < 0 || boundary > 5) return FL_ERR_ARG_OUTOFRANGE;
  const int d = boundary / 2, side = boundary % 2;
  if (h->bc[boundary] != FL_BC_PRESSURE_OUTLET) return FL_SUCCESS;
  const bool touches = side ? (h->dec.coord[d] == h->dec.ranks[d] - 1) : (h->dec.coord[d] == 0);
  if (!touches) return FL_SUCCESS;
  FL_HIP(hipSetDevice(h->device));
  launch_gst_bc(h->stream, h->g, pb_dev, V_dev, d, side, side ? h->ax[d].bcc_hi : h->ax[d].bcc_lo);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// The two shapes every boundary-condition vector of the reference has: a value per boundary face, written into the boundary
// faces of a face array (INSERT) or added to the cells next to the boundary (ADD).
static bool touches_boundary(const fl_poisson *h, int boundary)
{
  const int d = boundary / 2, side = boundary % 2;
  if (h->ax[d].periodic) return false;
  return side ? (h->dec.coord[d] == h->dec.ranks[d] - 1) : (h->dec.coord[d] == 0);
}

extern "C" int fl_boundary_set_faces(fl_poisson *h, int boundary, double coeff, const double *plane_dev, double *face_dev)
{
  if (!h || !plane_dev || !face_dev) return FL_ERR_ARG_NULL;
  if (boundary < 0 || boundary > 5) return FL_ERR_ARG_OUTOFRANGE;
  if (!touches_boundary(h, boundary)) return FL_SUCCESS;
  FL_HIP(hipSetDevice(h->device));
  launch_gst_bc(h->stream, h->g, plane_dev, face_dev, boundary / 2, boundary % 2, coeff);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_boundary_add_faces(fl_poisson *h, int boundary, double coeff, const double *plane_dev, double *face_dev)
{
  if (!h || !plane_dev || !face_dev) return FL_ERR_ARG_NULL;
  if (boundary < 0 || boundary > 5) return FL_ERR_ARG_OUTOFRANGE;
  if (!touches_boundary(h, boundary)) return FL_SUCCESS;
  FL_HIP(hipSetDevice(h->device));
  launch_gst_bc(h->stream, h->g, plane_dev, face_dev, boundary / 2, boundary % 2, coeff, 1);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_boundary_add_cells(fl_poisson *h, int boundary, double coeff, const double *plane_dev, double *cell_dev)
{
  if (!h || !plane_dev || !cell_dev) return FL_ERR_ARG_NULL;
  if (boundary < 0 || boundary > 5) return FL_ERR_ARG_OUTOFRANGE;
  if (!touches_boundary(h, boundary)) return FL_SUCCESS;
  FL_HIP(hipSetDevice(h->device));
  launch_bc_add_cells(h->stream, h->g, plane_dev, cell_dev, boundary / 2, boundary % 2, coeff);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_pressure_update(fl_poisson *h, int first, const double *dp, const double *p0, double *phalf, double *p)
{
  if (!h || !dp || !phalf || !p || (first && !p0)) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  launch_pressure_update(h->stream, h->ncell, first, dp, p0, phalf, p);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ KSPSolve

int fl_poll_scal(fl_poisson *h)
{
  FL_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(KspScal), hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// partial sums -> KspScal update.  Single rank: one kernel.  Multi rank: reduce, all-reduce, then the scalar kernel.
static int cg_fin(fl_poisson *h, int mode, int nblocks, int nslot, double *hist, int nhist)
{
  if (!h->multi) {
    launch_cg_fin(h->stream, mode, h->partial, nblocks, h->partial_stride, nullptr, h->scal, hist, nhist);
    return 0;
  }
  launch_reduce(h->stream, h->partial, nblocks, h->partial_stride, nslot, h->sums);
  FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  launch_cg_fin(h->stream, mode, nullptr, 0, 0, h->sums, h->scal, hist, nhist);
  return 0;
}

static int solve_cg(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st)
{
  const GridP &g   = h->g;
  const bool   jac = o->pc == FL_PC_JACOBI;
  FL_CHK(fl_ensure_vec(h, &h->r));
  FL_CHK(fl_ensure_vec(h, &h->P0));
  FL_CHK(fl_ensure_vec(h, &h->P1));
  FL_CHK(fl_ensure_vec(h, &h->q));
  FL_CHK(fl_ensure_vec(h, &h->xp));
  // variant 0 (default): k_cg_A<SQ = false> + k_cg_Bq, q = S p' formed twice and never stored (64 B/cell/iteration);
  // variant 2: k_cg_A stores q, k_cg_B reads it back (72 B/cell; the default until round 2); variant 1: one kernel per step
  const int variant_forced = FL_VARIANT(cg_variant, -1);
  const int variant = (o->variant == 0 && variant_forced >= 0) ? variant_forced : o->variant;
#ifndef FL_KBENCH_VARIANTS
  if (variant != 0) return FL_ERR_SUP;  // variants 1 and 2 (superseded, A/B material) live in the kbench build only (include/fluca_hip.h, fl_ksp_opts.variant)
#endif
  const bool storeq = variant != 0;  // variants 1 and 2 keep q in memory and update r with k_cg_B
  PlanA       plan = plan_cg_A(g, 0, 0);
  plan.sq          = storeq ? 1 : 0;
  const int qb_env = FL_VARIANT(cg_qb, 1);  // experiments (only with "overlap" = 0): 0 = k_cg_A stores no q at all
  plan.qb          = (!storeq && h->multi && qb_env) ? 1 : 0;  // several ranks: q of the boundary layers is kept for the overlapped exchange of r
  const int bq_chunks_env = FL_VARIANT(cgbq_chunks, 0);  // experiments: z chunks of k_cg_Bq (default: those of k_cg_A)
  PlanA planB = storeq ? plan_cg_B(g) : (bq_chunks_env > 0 ? plan_tiles(g, plan.ry, plan.nw, bq_chunks_env, 0) : plan);  // k_cg_Bq walks the tiles of k_cg_A
  {
    struct Force { int ry = 0, nw = 0, nchunk = 0; };
    Force fb;
    if (const char *e = variant_env("FLUCA_CGBQ_PLAN")) std::sscanf(e, "%d,%d,%d", &fb.ry, &fb.nw, &fb.nchunk);  // experiments: a tiling of its own for k_cg_Bq
    if (!storeq && (fb.ry == 1 || fb.ry == 2) && (fb.nw == 4 || (fb.nw == 8 && fb.ry == 2)) && fb.nchunk > 0 && g.ny >= 8) {
      const PlanA keep = planB;
      planB            = plan_tiles(g, fb.ry, fb.nw, fb.nchunk, 0);
      planB.pf = keep.pf; planB.nt = keep.nt; planB.remap = keep.remap; planB.sq = keep.sq; planB.qb = keep.qb;
    }
  }
  const int   nsb  = stream_blocks(g);
  const int   nab  = variant == 1 ? apply_dot_blocks(g) : plan.nblocks;
  FL_CHK(fl_ensure_partials(h, std::max(std::max(nsb, nab), planB.nblocks)));
  const int nhist = o->maxit + 1;
  FL_CHK(fl_ensure_hist(h, nhist));
  hipStream_t s = h->stream;

  const bool xbatch_env = knob(K_cg_xbatch) != 0;
  // q-free pair with batched x-updates: the padded x is not zeroed -- the first pair of updates (iteration 1) writes it without reading
  // it, and until then KspScal::x_valid = 0 tells k_cg_finish that it stands for 0
  const bool xlazy = !storeq && xbatch_env;
  KspScal &S = *h->scal_host;
  std::memset(&S, 0, sizeof(S));
  S.rtol         = o->rtol;
  S.atol         = o->atol;
  S.dtol         = o->dtol;
  S.ncell_global = (double)h->ax[0].n * (double)h->ax[1].n * (double)h->ax[2].n;
  S.maxit        = o->maxit;
  S.norm_type    = o->norm_type;
  S.nullspace    = o->remove_nullspace;
  S.rz_old       = 1.;
  S.x_valid      = xlazy ? 0 : 1;

  FL_HIP(hipEventRecord(h->ev0, s));
  FL_HIP(hipMemcpyAsync(h->scal, h->scal_host, sizeof(KspScal), hipMemcpyHostToDevice, s));
  // The direction buffers need no zeroing: the first iteration multiplies the old direction by beta = 0 and by alpha_prev = 0,
  // so whatever FINITE numbers an earlier solve left there drop out (wall ghosts included: they only ever meet the stencil
  // coefficient 0).  After a solve that produced NaN / Inf they are cleared.  x is zeroed by the kernel that pads b into r.
  if (h->poisoned) {
    FL_CHK(fl_zero_vec(h, h->P0));
    FL_CHK(fl_zero_vec(h, h->P1));
    FL_CHK(fl_zero_vec(h, h->xp));
    h->poisoned = false;
  }
  launch_cg_init(s, g, jac, b, h->r, xlazy ? nullptr : h->xp, h->partial, h->partial_stride, nsb);
  FL_CHK(cg_fin(h, 0, nsb, 5, h->hist, nhist));
  const bool ghosts = fl_any_ghost_exchange(h);
  if (ghosts) FL_CHK(fl_fill_ghosts(h, h->r));
  // single rank: the last block of k_cg_A / k_cg_B performs the scalar update itself (no k_cg_fin launches)
  const bool fusedfin_env = FL_VARIANT(fusedfin, 1) != 0;
  const bool overlap_env  = knob(K_overlap) != 0;  // 0: pack / transfer / unpack after the update kernel, on the handle's stream (A/B measurements, tests)
  const bool fusedfin = !h->multi && variant != 1 && fusedfin_env;
  // several ranks: the last block of k_cg_A / k_cg_B still reduces the rank's partial sums (no k_reduce launch); the
  // all-reduce and the scalar kernel follow
  const bool fusedsum = h->multi && variant != 1 && fusedfin_env;
  if (fusedfin || fusedsum) FL_HIP(hipMemsetAsync(h->tickets, 0, sizeof(unsigned) * 2, s));
  auto fin_sums = [&](int mode) -> int {
    FL_CHK(h->comm.allreduce(s, h->sums, NSLOT));
    launch_cg_fin(s, mode, nullptr, 0, 0, h->sums, h->scal, h->hist, nhist);
    return 0;
  };

  ProfEvents               prof_events;
  std::vector<hipEvent_t> &pev = prof_events.ev;
  if (o->profile) FL_CHK(prof_events.create(4 * (size_t)std::min(o->maxit, 4096)));  // around k_cg_A, around k_cg_Bq / k_cg_B

  const int every = o->check_every > 0 ? o->check_every : 16;
  int       it    = 0;
  int       nprof = 0;    // iterations whose kernels are bracketed by events so far
  int       hostcur = 0;  // host's view of KspScal::cur (exact while the device has not stopped)
  bool      done  = false;
  while (!done) {
    const int stop = std::min(o->maxit, it + every);
    for (; it < stop; ++it) {
      // profile = n: the kernels of every n-th PAIR of iterations are bracketed (k_cg_Bq alternates between two forms)
      const bool prof = o->profile > 0 && (it >> 1) % o->profile == 0 && (size_t)(4 * nprof + 3) < pev.size();
      const int  pi   = 4 * nprof;
      if (prof) ++nprof;
      if (variant == 1) {
        launch_cg_pupdate(s, g, jac, h->r, h->P0, h->P1, h->scal);
        if (ghosts) FL_CHK(fl_fill_ghosts(h, hostcur ? h->P0 : h->P1));
        if (prof) FL_HIP(hipEventRecord(pev[pi], s));
        launch_cg_apply_dot(s, g, h->P0, h->P1, h->q, h->xp, h->scal, h->partial);
        if (prof) FL_HIP(hipEventRecord(pev[pi + 1], s));
      } else {
        if (prof) FL_HIP(hipEventRecord(pev[pi], s));
        launch_cg_A(s, g, jac, plan, h->r, h->P0, h->P1, h->q, h->xp, h->scal, h->partial, (fusedfin || fusedsum) ? h->tickets : nullptr, h->hist, nhist, fusedsum ? h->sums : nullptr);
        if (prof) FL_HIP(hipEventRecord(pev[pi + 1], s));
      }
      const int modeA = storeq ? 1 : 3;  // q-free pair: k_cg_A does not touch x (see cg_fin_apply)
      if (fusedsum) FL_CHK(fin_sums(modeA));
      else if (!fusedfin) FL_CHK(cg_fin(h, modeA, nab, 1, h->hist, nhist));
      hostcur ^= 1;
      // several ranks: the boundary layers of the new r leave now (packed as r - alpha q), the transfers overlap k_cg_B
      // q-free pair: k_cg_Bq owns the x-update -- both updates of an iteration pair on the odd one (x is read and written every second
      // iteration only), or one per iteration with FLUCA_CG_XBATCH=0
      const int  xmode   = xbatch_env ? ((it & 1) ? (it == 1 ? 3 : 2) : 0) : 1;
      const bool overlap = ghosts && variant != 1 && h->multi && overlap_env;
      if (overlap) FL_CHK(fl_exchange_r_begin(h, h->r, h->q));
      if (prof) FL_HIP(hipEventRecord(pev[pi + 2], s));
      if (storeq) launch_cg_B(s, g, jac, planB, h->q, h->r, h->scal, h->partial, h->partial_stride, (fusedfin || fusedsum) ? h->tickets + 1 : nullptr, h->hist, nhist, fusedsum ? h->sums : nullptr);
      else launch_cg_Bq(s, g, jac, planB, xmode, h->P0, h->P1, h->r, h->xp, h->scal, h->partial, h->partial_stride, (fusedfin || fusedsum) ? h->tickets + 1 : nullptr, h->hist, nhist, fusedsum ? h->sums : nullptr);
      if (prof) FL_HIP(hipEventRecord(pev[pi + 3], s));
      // the handle's stream joins the exchange BEFORE the all-reduce is enqueued: the two RCCL operations never run at the same time
      // (one communicator, two streams), only the transfers and k_cg_B do
      if (overlap) FL_CHK(fl_exchange_r_end(h, h->r));
      const int modeB = (!storeq && xmode) ? 4 : 2;
      if (fusedsum) FL_CHK(fin_sums(modeB));
      else if (!fusedfin) FL_CHK(cg_fin(h, modeB, planB.nblocks, 5, h->hist, nhist));
      if (!overlap && ghosts && variant != 1) FL_CHK(fl_fill_ghosts(h, h->r));
    }
    FL_CHK(fl_poll_scal(h));
    if (h->scal_host->reason != 0 || it >= o->maxit) done = true;
  }
  launch_cg_finish(s, g, h->P0, h->P1, h->xp, x, h->scal, nsb);  // x = xp + the x-update still owed
  FL_HIP(hipEventRecord(h->ev1, s));
  FL_CHK(fl_poll_scal(h));
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  const KspScal &R = *h->scal_host;
  st->iters        = R.it;
  st->reason       = R.reason ? R.reason : FL_DIVERGED_ITS;
  if (R.reason == FL_DIVERGED_NANORINF || R.reason == FL_DIVERGED_DTOL || !std::isfinite(R.dp)) h->poisoned = true;
  st->rnorm0       = R.rnorm0;
  st->rnorm        = R.dp;
  st->seconds      = ms * 1e-3;
  st->kernel_ms    = 0.;
  st->kernel_launches = 0;
  st->kernel2_ms   = 0.;
  st->kernel2_launches = 0;
  if (o->profile) {
    // iterations enqueued after the device had stopped are early exits: count only those that ran
    int ran = 0;
    for (int a = 0, q = 0; a < R.it; ++a)
      if ((a >> 1) % o->profile == 0 && q++ < nprof) ++ran;
    prof_events.mean_of(ran, 4, 0, 1, &st->kernel_ms, &st->kernel_launches);
    prof_events.mean_of(ran, 4, 2, 3, &st->kernel2_ms, &st->kernel2_launches);
  }
  if (o->history && o->nhistory > 0) {
    const int n = std::min(o->nhistory, R.it + 1);
    FL_HIP(hipMemcpy(o->history, h->hist, sizeof(double) * n, hipMemcpyDeviceToHost));
  }
  return FL_SUCCESS;
}

extern "C" int fl_poisson_solve(fl_poisson *h, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats)
{
  if (!h || !b_dev || !x_dev || !opts || !stats) return FL_ERR_ARG_NULL;
  if (opts->maxit < 0 || opts->maxit > 10000000) return FL_ERR_ARG_OUTOFRANGE;
  if (opts->pc != FL_PC_NONE && opts->pc != FL_PC_JACOBI && opts->pc != FL_PC_MG) return FL_ERR_SUP;
  if (opts->initial_guess_nonzero) return FL_ERR_SUP;  // the Schur solvers start from zero, as the reference's kspS does (nsbasic.c:250-251)
  FL_HIP(hipSetDevice(h->device));
  std::memset(stats, 0, sizeof(*stats));
  if (opts->pc == FL_PC_MG) {
    if (opts->type != FL_KSP_CG) return FL_ERR_SUP;
    if (opts->cg_single_reduction) return FL_ERR_SUP;  // the cycle's sums ride on its last smoothing sweep: no single-reduction form of that loop is built
    return fl_solve_cg_mg(h, b_dev, x_dev, opts, stats);
  }
  switch (opts->type) {
  case FL_KSP_CG:
    if (opts->norm_type < 0 || opts->norm_type > FL_NORM_NONE) return FL_ERR_ARG_OUTOFRANGE;
    if (opts->cg_single_reduction) return fl_solve_cg_sr(h, b_dev, x_dev, opts, stats);  // -ksp_cg_single_reduction
    return solve_cg(h, b_dev, x_dev, opts, stats);
  case FL_KSP_BCGS:
    // KSPBCGS: left preconditioning, preconditioned residual norm only
    if (opts->norm_type != FL_NORM_PRECONDITIONED) return FL_ERR_SUP;
    return fl_solve_bcgs(h, b_dev, x_dev, opts, stats);
  case FL_KSP_CHEBYSHEV:
    if (opts->norm_type < 0 || opts->norm_type > FL_NORM_NONE) return FL_ERR_ARG_OUTOFRANGE;
    return fl_solve_cheb(h, b_dev, x_dev, opts, stats);
  default:
    return FL_ERR_SUP;
  }
}

extern "C" int fl_poisson_gershgorin(const fl_poisson *h, int pc, double *bound)
{
  if (!h || !bound) return FL_ERR_ARG_NULL;
  if (pc != FL_PC_NONE && pc != FL_PC_JACOBI) return FL_ERR_ARG_OUTOFRANGE;
  *bound = fl_gershgorin_bound(h, pc == FL_PC_JACOBI);
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ plain device memory

extern "C" int fl_current_device(int *device)
{
  if (!device) return FL_ERR_ARG_NULL;
  FL_HIP(hipGetDevice(device));
  return FL_SUCCESS;
}

extern "C" int fl_malloc(int device, size_t bytes, void **dev_out)
{
  if (!dev_out) return FL_ERR_ARG_NULL;
  *dev_out = nullptr;
  FL_HIP(hipSetDevice(device));
  if (hipMalloc(dev_out, bytes ? bytes : 8) != hipSuccess) return FL_ERR_MEM;
  // hipMemset on the null stream is asynchronous and the handles work on non-blocking streams, which do not wait for
  // the null stream: finish it here, or the zeroes can land on top of the first results written into this buffer.
  FL_HIP(hipMemset(*dev_out, 0, bytes ? bytes : 8));
  FL_HIP(hipDeviceSynchronize());
  return FL_SUCCESS;
}
extern "C" int fl_free(int device, void *dev)
{
  if (!dev) return FL_SUCCESS;
  FL_HIP(hipSetDevice(device));
  FL_HIP(hipFree(dev));
  return FL_SUCCESS;
}
extern "C" int fl_memcpy_h2d(int device, void *dev, const void *host, size_t bytes)
{
  if (!dev || !host) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(device));
  FL_HIP(hipDeviceSynchronize());  // whatever still reads or writes `dev` on a handle's stream finishes first
  FL_HIP(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
  FL_HIP(hipDeviceSynchronize());
  return FL_SUCCESS;
}
// Page-locked host memory and a copy out of it that is ordered on the handle's stream like a kernel: no device-wide wait on either side (the C host
// mirror hands over some thirty boundary planes per time step; fl_memcpy_h2d drains the device twice per plane).
extern "C" int fl_malloc_host(size_t bytes, void **host_out)
{
  if (!host_out) return FL_ERR_ARG_NULL;
  *host_out = nullptr;
  if (hipHostMalloc(host_out, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) return FL_ERR_MEM;
  return FL_SUCCESS;
}
extern "C" int fl_free_host(void *host)
{
  if (!host) return FL_SUCCESS;
  FL_HIP(hipHostFree(host));
  return FL_SUCCESS;
}
extern "C" int fl_poisson_upload(fl_poisson *h, void *dev, const void *host, size_t bytes)
{
  if (!h || !dev || !host) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  if (!h->ev_upload) FL_HIP(hipEventCreateWithFlags(&h->ev_upload, hipEventDisableTiming));
  FL_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream));
  FL_HIP(hipEventRecord(h->ev_upload, h->stream));
  return FL_SUCCESS;
}
extern "C" int fl_poisson_upload_fence(fl_poisson *h)
{
  if (!h) return FL_ERR_ARG_NULL;
  if (h->ev_upload) FL_HIP(hipEventSynchronize(h->ev_upload));
  return FL_SUCCESS;
}
extern "C" int fl_memcpy_d2h(int device, void *host, const void *dev, size_t bytes)
{
  if (!dev || !host) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(device));
  FL_HIP(hipDeviceSynchronize());
  FL_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ comm init

extern "C" int fl_comm_unique_id(void *out128)
{
  if (!out128) return FL_ERR_ARG_NULL;
  FL_CHK(g_rccl.load());
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == FL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  FL_NCCL(g_rccl.GetUniqueId(&id));
  std::memcpy(out128, &id, sizeof(id));
  return FL_SUCCESS;
}

extern "C" int fl_poisson_comm_init_rccl(fl_poisson *h, const void *id128, int rank, int nranks)
{
  if (h) fl_mg_destroy(h);  // the levels of a multigrid hierarchy borrow this handle's communicator: rebuilt on the next solve

  if (!h || !id128) return FL_ERR_ARG_NULL;
  if (nranks != h->dec.ranks[0] * h->dec.ranks[1] * h->dec.ranks[2] || rank < 0 || rank >= nranks) return FL_ERR_ARG_WRONG;
  FL_CHK(g_rccl.load());
  FL_HIP(hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  h->comm.destroy();
  FL_NCCL(g_rccl.CommInitRank(&h->comm.nccl, nranks, id, rank));
  h->comm.kind   = Comm::RCCL;
  h->cheb2_agreed[0] = h->cheb2_agreed[1] = -1;  // a new communicator: the ranks vote again (fl_cheb2_agree)
  h->comm.rank   = rank;
  h->comm.nranks = nranks;
  return FL_SUCCESS;
}

extern "C" int fl_poisson_comm_init_host(fl_poisson *h, fl_exchange_fn xchg, fl_allreduce_fn allred, void *ctx, int rank, int nranks)
{
  if (h) fl_mg_destroy(h);  // the levels of a multigrid hierarchy borrow this handle's communicator: rebuilt on the next solve

  if (!h || !xchg || !allred) return FL_ERR_ARG_NULL;
  if (nranks != h->dec.ranks[0] * h->dec.ranks[1] * h->dec.ranks[2] || rank < 0 || rank >= nranks) return FL_ERR_ARG_WRONG;
  h->comm.destroy();
  h->comm.kind   = Comm::HOST;
  h->cheb2_agreed[0] = h->cheb2_agreed[1] = -1;  // a new communicator: the ranks vote again (fl_cheb2_agree)
  h->comm.xchg   = xchg;
  h->comm.allred = allred;
  h->comm.ctx    = ctx;
  h->comm.rank   = rank;
  h->comm.nranks = nranks;
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ one-shot all-reduce (fl_handle.h: OneShotBox)
namespace fl {
// One wave.  Lane l < nranks delivers to peer l and later fetches rank l's slot; every store that a peer waits for is a system-scope release, every
// load of a flag a system-scope acquire (the mailboxes are fine-grained memory, possibly of another device).  A wait gives up after about two
// seconds of wall clock and raises the mailbox's error flag -- a kernel that spins for ever would take the GPU (and its neighbours) down with it.
__global__ void __launch_bounds__(64) k_oneshot_allreduce(OneShotBox *const *boxes, int rank, int nranks, unsigned long long number, double *vals, int n)
{
  const int lane = threadIdx.x, par = (int)(number & 1ull);
  __shared__ double got[NSLOT][NSLOT];
  if (lane < nranks) {
    OneShotBox *peer = boxes[lane];
    for (int a = 0; a < n; ++a) __hip_atomic_store(&peer->slot[par][rank][a], vals[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&peer->seq[par][rank], number, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    OneShotBox     *mine = boxes[rank];
    const long long t0 = wall_clock64();  // 100 MHz
    bool            ok = mine->error == 0;  // sticky: after one timed-out wait every later call gives up at once (NaN sums end the solve)
    while (ok && __hip_atomic_load(&mine->seq[par][lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != number) {
      if (wall_clock64() - t0 > 200000000ll) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) mine->error = 1;
    for (int a = 0; a < n; ++a) got[lane][a] = ok ? __hip_atomic_load(&mine->slot[par][lane][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : nan("");
  }
  __syncthreads();
  if (lane < n) {
    double sum = 0.;
    for (int r = 0; r < nranks; ++r) sum += got[r][lane];  // rank order: the same bits on every rank
    vals[lane] = sum;
  }
}
void launch_oneshot_allreduce(hipStream_t st, OneShotBox *const *boxes, int rank, int nranks, unsigned long long number, double *vals, int n)
{
  hipLaunchKernelGGL(k_oneshot_allreduce, dim3(1), dim3(64), 0, st, boxes, rank, nranks, number, vals, n);
}
}  // namespace fl

// This rank's mailbox (created by the first call) as a 64-byte hipIpcMemHandle_t for the other PROCESSES, and its address for other handles of
// this process.  The host gathers the handles of all ranks (torch.distributed, MPI, ...) and hands every rank the whole list.
extern "C" int fl_poisson_comm_oneshot_handle(fl_poisson *h, void *ipc_handle64, void **address)
{
  if (!h) return FL_ERR_ARG_NULL;
  if (h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  if (h->comm.nranks > NSLOT) return FL_ERR_SUP;
  FL_HIP(hipSetDevice(h->device));
  if (!h->comm.box) {
    FL_HIP(hipExtMallocWithFlags((void **)&h->comm.box, sizeof(OneShotBox), hipDeviceMallocFinegrained));
    FL_HIP(hipMemset(h->comm.box, 0, sizeof(OneShotBox)));
  }
  if (ipc_handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == FL_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    hipIpcMemHandle_t hd;
    FL_HIP(hipIpcGetMemHandle(&hd, h->comm.box));
    std::memcpy(ipc_handle64, &hd, sizeof(hd));
  }
  if (address) *address = h->comm.box;
  return FL_SUCCESS;
}
// handles: nranks x 64 bytes in rank order (NULL entries are not allowed), or -- same process -- addresses: nranks mailbox addresses as
// fl_poisson_comm_oneshot_handle returned them.  Exactly one of the two is given.  From then on "allreduce" = 1 routes this handle's scalar
// reductions through the mailboxes (the multigrid levels keep the communicator's own all-reduce).
extern "C" int fl_poisson_comm_oneshot_attach(fl_poisson *h, const void *handles, void *const *addresses)
{
  if (!h || (!handles == !addresses)) return FL_ERR_ARG_NULL;
  Comm &c = h->comm;
  if (c.kind == Comm::NONE || !c.box) return FL_ERR_ARG_WRONGSTATE;
  FL_HIP(hipSetDevice(h->device));
  std::vector<OneShotBox *> peers((size_t)c.nranks, nullptr);
  for (int r = 0; r < c.nranks; ++r) {
    if (r == c.rank) peers[(size_t)r] = c.box;
    else if (addresses) peers[(size_t)r] = (OneShotBox *)addresses[r];
    else {
      hipIpcMemHandle_t hd;
      std::memcpy(&hd, (const char *)handles + (size_t)r * FL_IPC_HANDLE_BYTES, sizeof(hd));
      void *p = nullptr;
      FL_HIP(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
      c.ipc_opened.push_back(p);
      peers[(size_t)r] = (OneShotBox *)p;
    }
    if (!peers[(size_t)r]) return FL_ERR_ARG_NULL;
  }
  if (!c.peers_dev) FL_HIP(hipMalloc((void **)&c.peers_dev, sizeof(OneShotBox *) * NSLOT));
  FL_HIP(hipMemcpy(c.peers_dev, peers.data(), sizeof(OneShotBox *) * peers.size(), hipMemcpyHostToDevice));
  c.oneshot_calls = 0;
  c.oneshot_ready = true;
  return FL_SUCCESS;
}
// 1 if a wait of a one-shot all-reduce on this handle ever ran into its time limit (the sums of that call are NaN)
extern "C" int fl_poisson_comm_oneshot_error(fl_poisson *h, int *error)
{
  if (!h || !error) return FL_ERR_ARG_NULL;
  *error = 0;
  if (!h->comm.box) return FL_SUCCESS;
  FL_HIP(hipSetDevice(h->device));
  OneShotBox host;
  FL_HIP(hipStreamSynchronize(h->stream));
  FL_HIP(hipMemcpy(&host, h->comm.box, sizeof(host), hipMemcpyDeviceToHost));
  *error = host.error;
  return FL_SUCCESS;
}

extern "C" int fl_poisson_comm_info(fl_poisson *h, fl_comm_info *out)
{
  if (!h || !out) return FL_ERR_ARG_NULL;
  std::memset(out, 0, sizeof(*out));
  out->transport = h->comm.kind == Comm::RCCL ? 1 : h->comm.kind == Comm::HOST ? 2 : 0;
  out->rank      = h->comm.rank;
  out->nranks    = h->comm.nranks;
  out->loopback  = h->loopback ? 1 : 0;
  if (h->comm.kind == Comm::RCCL && h->comm.nccl) {  // what the communicator itself says, not what init was told
    FL_NCCL(g_rccl.CommCount(h->comm.nccl, &out->nranks));
    FL_NCCL(g_rccl.CommUserRank(h->comm.nccl, &out->rank));
  }
  if (h->multi) {
    int periodic[3];
    for (int d = 0; d < 3; ++d) periodic[d] = h->ax[d].periodic;
    fl_halo_msg plan[12];
    const int   np = fl_halo_plan(&h->dec, periodic, plan);
    std::vector<int> peers;
    for (int a = 0; a < np; ++a) {
      if (plan[a].send_boundary >= 0) out->halo_bytes += (int64_t)sizeof(double) * (int64_t)plane_size(h, plan[a].send_boundary / 2);
      if (std::find(peers.begin(), peers.end(), plan[a].peer) == peers.end()) peers.push_back(plan[a].peer);
    }
    out->messages   = np;
    out->neighbours = (int)peers.size();
  }
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ kernel micro-bench (tools/kbench.py)
// Not part of the public C-ABI (not declared in fluca_hip.h): times one hot kernel in isolation with HIP events.
//   kernel 0: k_cg_A (ry, pf, nchunk)   1: k_cg_B (ry, nchunk)   2: streaming reference with ry reads / pf writes
extern "C" int fldbg_bench(fl_poisson *h, int kernel, int ry, int pf, int nchunk, int reps, const double *src_dev, double *ms_out, int *nblocks_out)
{
  if (!h || !ms_out) return FL_ERR_ARG_NULL;
#ifndef FL_KBENCH_VARIANTS
  if (kernel != 0 && kernel != 2 && kernel != 3) return FL_ERR_SUP;  // the product keeps what bench.py measures with: k_cg_A and the streaming probes
#endif
  FL_HIP(hipSetDevice(h->device));
  const GridP &g = h->g;
  if (kernel == 8) {
    // experiment: forget the current padded vectors WITHOUT freeing them (they stay allocated, so the next set must land
    // on different physical memory)
    FL_HIP(hipStreamSynchronize(h->stream));
    h->vec_bases.clear();
    h->vec_bytes = 0;
    h->nvec = 0;
    h->slab = nullptr;
    for (double **v : {&h->r, &h->P0, &h->P1, &h->q, &h->xp, &h->w0, &h->w1, &h->w2}) *v = nullptr;
    *ms_out = 0.;
    return FL_SUCCESS;
  }
  if (kernel == 9) {
    // experiment: drop every padded vector so that the next call gets fresh physical memory (ry extra junk allocations
    // of pf MiB each are made first and kept, to shift the placement)
    FL_HIP(hipStreamSynchronize(h->stream));
    for (void *p : h->vec_bases) (void)hipFree(p);
    h->vec_bases.clear();
    h->vec_bytes = 0;
    h->nvec = 0;
    h->slab = nullptr;
    for (double **v : {&h->r, &h->P0, &h->P1, &h->q, &h->xp, &h->w0, &h->w1, &h->w2}) *v = nullptr;
    for (int a = 0; a < ry; ++a) {
      void *junk = nullptr;
      FL_HIP(hipMalloc(&junk, (size_t)pf << 20));
    }
    *ms_out = 0.;
    return FL_SUCCESS;
  }
  for (double **v : {&h->r, &h->P0, &h->P1, &h->q, &h->xp, &h->w0}) FL_CHK(fl_ensure_vec(h, v));
  hipStream_t s = h->stream;
  if (src_dev) {
    launch_pad_copy(s, g, src_dev, h->r);
    launch_pad_copy(s, g, src_dev, h->P0);
    launch_pad_copy(s, g, src_dev, h->xp);
    launch_pad_copy(s, g, src_dev, h->q);
  }
  // ry = 10*RY + NW(4|8) ; pf = 100*remap + 10*PF + NT
  PlanA plan = (kernel <= 1) ? plan_tiles(g, ry / 10, ry % 10, nchunk, 512) : plan_cg_A(g, 0, 0);
  if (kernel <= 1) {
    plan.remap = pf / 100;
    plan.pf    = (pf / 10) % 10;
    plan.nt    = pf % 10;
  }
  FL_CHK(fl_ensure_partials(h, plan.nblocks));
  KspScal &S = *h->scal_host;
  std::memset(&S, 0, sizeof(S));
  S.beta = 0.5; S.alpha = 1e-3; S.zshift = 1e-4; S.ncell_global = (double)h->ncell; S.maxit = 1 << 30; S.pending_x = 1; S.nullspace = 1; S.rz = 1.;
  FL_HIP(hipMemcpyAsync(h->scal, h->scal_host, sizeof(KspScal), hipMemcpyHostToDevice, s));
  if (FL_VARIANT(print_ptrs, 0) && kernel == 0)
    std::fprintf(stderr, "[ptrs] r=%p P0=%p P1=%p q=%p xp=%p w0=%p\n", (void *)h->r, (void *)h->P0, (void *)h->P1, (void *)h->q, (void *)h->xp, (void *)h->w0);
  auto once = [&]() {
    if (kernel == 0) launch_cg_A(s, g, true, plan, h->r, h->P0, h->P1, h->q, h->xp, h->scal, h->partial, nullptr, nullptr, 0);
    else if (kernel == 1) launch_cg_B(s, g, true, plan, h->q, h->r, h->scal, h->partial, h->partial_stride, nullptr, nullptr, 0);
    else if (kernel == 2) launch_stream_ref(s, ry, pf, (int64_t)(h->padlen - 256) / 2, h->r, h->P0, h->xp, h->P1, h->q, h->w0);
    else {
      // kernel 3: parametric stream.  ry = 10*NR + NW, pf = 10*U + NT, nchunk = blocks
      launch_stream_par(s, ry / 10, ry % 10, pf / 10, pf % 10, nchunk, (int64_t)(h->padlen - 256) / 2, h->r, h->P0, h->xp, h->P1, h->q, h->w0);
    }
  };
  once();
  once();
  FL_HIP(hipEventRecord(h->ev0, s));
  for (int a = 0; a < reps; ++a) once();
  FL_HIP(hipEventRecord(h->ev1, s));
  FL_HIP(hipStreamSynchronize(s));
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_out = ms / reps;
  if (nblocks_out) *nblocks_out = plan.nblocks;
  return FL_SUCCESS;
}

#ifdef FL_KBENCH_VARIANTS  // experiments behind the placement notes of DESIGN.md: not in the product
// Experiment behind fl_poisson_tune_placement (tools/experiments/pool_probe.py): K vectors allocated once, M random
// assignments of five of them to the roles (r, p0, p1, q, x) of k_cg_A, probe time of each.
extern "C" int fldbg_pool_probe(fl_poisson *h, int K, int M, unsigned seed, double *ms_out, int *sel_out)
{
  if (!h || !ms_out || K < 5 || K > 64) return FL_ERR_ARG_WRONG;
  FL_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  PlanA       plan = plan_cg_A(h->g, 0, 0);
  plan.probe       = 1;
  FL_CHK(fl_ensure_partials(h, plan.nblocks));
  std::vector<double *> pool(K, nullptr);
  for (int k = 0; k < K; ++k) {
    FL_HIP(hipMalloc((void **)&pool[k], sizeof(double) * h->padlen));
    FL_HIP(hipMemsetAsync(pool[k], 0x3f, sizeof(double) * h->padlen, s));
  }
  KspScal *scal2 = nullptr;
  FL_HIP(hipMalloc((void **)&scal2, 2 * sizeof(KspScal)));
  KspScal S2[2];
  std::memset(S2, 0, sizeof(S2));
  for (int a = 0; a < 2; ++a) {
    S2[a].beta = 0.5; S2[a].alpha = 1e-3; S2[a].zshift = 1e-4; S2[a].ncell_global = (double)h->ncell; S2[a].maxit = 1 << 30; S2[a].cur = a;
  }
  FL_HIP(hipMemcpy(scal2, S2, sizeof(S2), hipMemcpyHostToDevice));
  unsigned st = seed * 2654435761u + 12345u;
  auto     rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (int m = 0; m < M; ++m) {
    int sel[5];
    for (int a = 0; a < 5; ++a) {
      bool ok;
      do {
        sel[a] = (int)(rnd() % (unsigned)K);
        ok     = true;
        for (int b = 0; b < a; ++b) ok &= sel[b] != sel[a];
      } while (!ok);
    }
    auto probe = [&](int reps) {
      for (int r = 0; r < reps; ++r)
        for (int par = 0; par < 2; ++par) launch_cg_A(s, h->g, true, plan, pool[sel[0]], pool[sel[1]], pool[sel[2]], pool[sel[3]], pool[sel[4]], scal2 + par, h->partial, nullptr, nullptr, 0);
    };
    probe(1);
    FL_HIP(hipEventRecord(h->ev0, s));
    probe(2);
    FL_HIP(hipEventRecord(h->ev1, s));
    FL_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    ms_out[m] = ms / 4.;
    if (sel_out)
      for (int a = 0; a < 5; ++a) sel_out[m * 5 + a] = sel[a];
  }
  (void)hipFree(scal2);
  for (double *p : pool) (void)hipFree(p);
  return 0;
}

// Experiment behind the placement note in DESIGN.md (tools/experiments/arena_probe.py): one arena allocated once, six
// streams (nr reads, nw writes of n doubles each) placed at caller-chosen byte offsets inside it, launch time of the plain
// streaming kernel.  Not part of the public C-ABI.
extern "C" int fldbg_arena_probe(fl_poisson *h, int64_t arena_bytes, const int64_t *off_bytes, int64_t n, int nr, int nw, int reps, double *ms_out, void **arena_out)
{
  if (!h || !off_bytes || !ms_out) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  static void   *arena = nullptr;
  static int64_t cap   = 0;
  if (arena_bytes < 0) {  // release
    if (arena) (void)hipFree(arena);
    arena = nullptr;
    cap   = 0;
    return FL_SUCCESS;
  }
  if (cap < arena_bytes) {
    if (arena) (void)hipFree(arena);
    arena = nullptr;
    FL_HIP(hipMalloc(&arena, (size_t)arena_bytes));
    FL_HIP(hipMemset(arena, 0, (size_t)arena_bytes));
    cap = arena_bytes;
  }
  if (arena_out) *arena_out = arena;
  double *v[6];
  for (int a = 0; a < 6; ++a) {
    if (off_bytes[a] < 0 || off_bytes[a] + n * 8 > cap || (off_bytes[a] & 15)) return FL_ERR_ARG_OUTOFRANGE;
    v[a] = (double *)((char *)arena + off_bytes[a]);
  }
  hipStream_t s = h->stream;
  auto once = [&]() { launch_stream_ref(s, nr, nw, n / 2, v[0], v[1], v[2], v[3], v[4], v[5]); };
  once();
  FL_HIP(hipEventRecord(h->ev0, s));
  for (int a = 0; a < reps; ++a) once();
  FL_HIP(hipEventRecord(h->ev1, s));
  FL_HIP(hipStreamSynchronize(s));
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_out = ms / reps;
  return FL_SUCCESS;
}

// tools/experiments/arena_probe3.py: the plain streaming kernel on six caller-owned device pointers.  Not part of the C-ABI.
extern "C" int fldbg_stream_ptrs(fl_poisson *h, void *const *ptrs, int64_t n, int nr, int nw, int reps, double *ms_out)
{
  if (!h || !ptrs || !ms_out) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  double     *v[6];
  for (int a = 0; a < 6; ++a) v[a] = (double *)ptrs[a];
  auto once = [&]() { launch_stream_ref(s, nr, nw, n / 2, v[0], v[1], v[2], v[3], v[4], v[5]); };
  once();
  once();
  FL_HIP(hipEventRecord(h->ev0, s));
  for (int a = 0; a < reps; ++a) once();
  FL_HIP(hipEventRecord(h->ev1, s));
  FL_HIP(hipStreamSynchronize(s));
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_out = ms / reps;
  return FL_SUCCESS;
}

// tools/experiments/phase_scan.py: k_cg_A (kernel 0: pointers r, P0, P1, q, x) or the fused two-step Chebyshev kernel
// (kernel 1: X0, X1, B, D0, D1) on five caller-owned padded vectors, both parities of the double buffers.  Not part of the C-ABI.
extern "C" int fldbg_kernel_ptrs(fl_poisson *h, int kernel, void *const *ptrs, int nchunk, int reps, double *ms_out)
{
  if (!h || !ptrs || !ms_out) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  double     *v[5];
  for (int a = 0; a < 5; ++a) v[a] = (double *)ptrs[a];
  KspScal *scal2 = nullptr;
  FL_HIP(hipMalloc((void **)&scal2, 2 * sizeof(KspScal)));
  KspScal S2[2];
  std::memset(S2, 0, sizeof(S2));
  for (int a = 0; a < 2; ++a) {
    S2[a].beta = 0.5; S2[a].alpha = 1e-3; S2[a].zshift = 1e-4; S2[a].ncell_global = (double)h->ncell; S2[a].maxit = 1 << 30; S2[a].cur = a; S2[a].dcur = a;
    S2[a].cheb_rho = 0.3; S2[a].cheb_c = 0.2; S2[a].mu = 1.2; S2[a].ck = 1.5; S2[a].ckm1 = 1.2; S2[a].omegaprod = 2.4; S2[a].scale = 0.9;
  }
  FL_HIP(hipMemcpy(scal2, S2, sizeof(S2), hipMemcpyHostToDevice));
  PlanA     plan = plan_cg_A(h->g, 0, nchunk);
  Cheb2Plan cp   = fl_cheb2_plan(h->g);
  if (nchunk > 0 && kernel == 1) {
    cp.nchunk  = nchunk;
    cp.zc      = (h->g.nz + nchunk - 1) / nchunk;
    cp.nchunk  = (h->g.nz + cp.zc - 1) / cp.zc;
    cp.nblocks = cp.tiles * cp.nchunk;
  }
  FL_CHK(fl_ensure_partials(h, std::max(plan.nblocks, cp.nblocks)));
  KspScal *keep = h->scal;
  auto     once = [&]() {
    for (int par = 0; par < 2; ++par) {
      if (kernel == 0) launch_cg_A(s, h->g, true, plan, v[0], v[1], v[2], v[3], v[4], scal2 + par, h->partial, nullptr, nullptr, 0);
      else {
        h->scal = scal2 + par;
        fl_launch_cheb2(h, cp, true, v[0], v[1], v[2], v[3], v[4]);
      }
    }
  };
  once();
  FL_HIP(hipEventRecord(h->ev0, s));
  for (int a = 0; a < reps; ++a) once();
  FL_HIP(hipEventRecord(h->ev1, s));
  FL_HIP(hipStreamSynchronize(s));
  h->scal = keep;
  (void)hipFree(scal2);
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_out = ms / (2 * reps);
  return FL_SUCCESS;
}
#endif  // FL_KBENCH_VARIANTS
