// fl_handle.h -- the fl_poisson handle, the halo transports and the kernel launchers shared by the .hip files.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <list>
#include <mutex>
#include <string>

#include "fl_internal.h"
#include "fl_knobs.h"

namespace fl {

// launchers defined in fl_kernels.hip
void launch_pad_copy(hipStream_t, const GridP &, const double *, double *);
void launch_unpad_copy(hipStream_t, const GridP &, const double *, double *, const double *);
void launch_project_all(hipStream_t, const GridP &, const double *p, double *const v[3], double *const V[3]);
bool project_six_usable(const GridP &, const double *p_unpadded, double *const v[3], double *const V[3]);
void launch_project_six(hipStream_t, const GridP &, const double *p, bool direct, int per, double *const v[3], double *const V[3]);
void launch_wrap(hipStream_t, const GridP &, double *, int axis, int nvec = 1, int64_t vstride = 0);
void launch_pack_faces_sr(hipStream_t, const GridP &, const double *r, const double *sb, const double *W, const KspScal *s, double *const bufs[6]);
void launch_face_ext(hipStream_t, const GridP &, double *v, double *buf, int axis, int side, int ea, int eb, int mode);
void launch_face_ext_deep(hipStream_t, const GridP &, double *v, double *buf, int axis, int side, int ea, int eb, int dp, int mode);
void launch_pack(hipStream_t, const GridP &, const double *, double *, int, int);
void launch_unpack(hipStream_t, const GridP &, double *, const double *, int, int);
void launch_pack_faces(hipStream_t, const GridP &, const double *, double *const bufs[6]);
void launch_unpack_faces(hipStream_t, const GridP &, double *, double *const bufs[6]);
void launch_pack_faces_rq(hipStream_t, const GridP &, const double *, const double *, const KspScal *, double *const bufs[6]);
void launch_apply(hipStream_t, const GridP &, const double *, double *, int);
void launch_diagonal(hipStream_t, const GridP &, double *);
void launch_rhs(hipStream_t, const GridP &, const double *, const double *, const double *, const double *, const double *, const double *, const double *, double *);
void launch_face_plane0(hipStream_t, const GridP &, const double *, double *, int);
void launch_project_faces(hipStream_t, const GridP &, const double *, double *, int);
void launch_project_cells(hipStream_t, const GridP &, const double *, double *, int);
void launch_gst_bc(hipStream_t, const GridP &, const double *, double *, int, int, double, int add = 0);
void launch_bc_add_cells(hipStream_t, const GridP &, const double *, double *, int, int, double);
void launch_pressure_update(hipStream_t, int64_t, int, const double *, const double *, double *, double *);
void launch_reduce(hipStream_t, const double *, int, int, int, double *);
void launch_cg_fin(hipStream_t, int, const double *, int, int, const double *, KspScal *, double *, int);
int  stream_blocks(const GridP &);
void launch_cg_init(hipStream_t, const GridP &, bool, const double *, double *, double *, double *, int, int);
void launch_cg_finish(hipStream_t, const GridP &, const double *, const double *, const double *, double *, const KspScal *, int);
struct PlanA {
  int ry, nw, tiles_x, tiles_y, nchunk, zc, nblocks, pf, nt, remap, probe;
  int sq, qb;  // keep in step with the definition in fl_kernels.hip
};
PlanA plan_tiles(const GridP &, int ry, int nw, int nchunk_force, int target_blocks, int min_zc = 8);
PlanA plan_cg_A(const GridP &, int, int);
PlanA plan_cg_B(const GridP &);
void  launch_cg_A(hipStream_t, const GridP &, bool, const PlanA &, const double *, double *, double *, double *, double *, KspScal *, double *, unsigned *, double *, int, double *sums = nullptr);
void  launch_cg_Bq(hipStream_t, const GridP &, bool, const PlanA &, int xmode, const double *P0, const double *P1, double *r, double *x, KspScal *, double *partial, int stride, unsigned *counter, double *hist, int nhist,
                   double *sums = nullptr);
void  launch_cg_B(hipStream_t, const GridP &, bool, const PlanA &, const double *, double *, KspScal *, double *, int, unsigned *, double *, int, double *sums = nullptr);
void  launch_stream_ref(hipStream_t, int, int, int64_t, const double *, const double *, const double *, double *, double *, double *);
void  launch_stream_par(hipStream_t, int, int, int, int, int, int64_t, const double *, const double *, const double *, double *, double *, double *);
void  launch_cg_pupdate(hipStream_t, const GridP &, bool, const double *, double *, double *, const KspScal *);
int   apply_dot_blocks(const GridP &);
void  launch_cg_apply_dot(hipStream_t, const GridP &, const double *, const double *, double *, double *, const KspScal *, double *);

// ------------------------------------------------------------------------------------------------ transports

struct Msg {
  int     peer;
  double *send, *recv;  // device
  int64_t count;
  int     sendtag, recvtag;
};

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId)    GetUniqueId    = nullptr;
  decltype(&ncclCommInitRank)   CommInitRank   = nullptr;
  decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
  decltype(&ncclSend)           Send           = nullptr;
  decltype(&ncclRecv)           Recv           = nullptr;
  decltype(&ncclAllReduce)      AllReduce      = nullptr;
  decltype(&ncclGroupStart)     GroupStart     = nullptr;
  decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount)      CommCount      = nullptr;
  decltype(&ncclCommUserRank)   CommUserRank   = nullptr;
  std::mutex mu;  // handles of several host threads may reach their first RCCL call together
  int        load()
  {
    std::lock_guard<std::mutex> lock(mu);
    if (lib) return 0;
    // the soname: inside a process that already loaded torch this resolves to the very RCCL torch.distributed uses
    void *l = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!l) {
      std::fprintf(stderr, "[flucahip] cannot dlopen librccl: %s\n", dlerror());
      return FL_ERR_LIB;
    }
#define FL_SYM(n)                                                \
  n = (decltype(n))dlsym(l, "nccl" #n);                          \
  if (!n) {                                                      \
    std::fprintf(stderr, "[flucahip] librccl lacks nccl" #n "\n"); \
    return FL_ERR_LIB;                                           \
  }
    FL_SYM(GetUniqueId) FL_SYM(CommInitRank) FL_SYM(CommDestroy) FL_SYM(Send) FL_SYM(Recv) FL_SYM(AllReduce) FL_SYM(GroupStart) FL_SYM(GroupEnd) FL_SYM(GetErrorString) FL_SYM(CommCount) FL_SYM(CommUserRank)
#undef FL_SYM
    lib = l;  // last: a reader that sees lib != nullptr sees every symbol
    return 0;
  }
};
inline Rccl g_rccl;

#define FL_NCCL(call)                                                                                              \
  do {                                                                                                             \
    ncclResult_t r_ = (call);                                                                                      \
    if (r_ != ncclSuccess) {                                                                                       \
      std::fprintf(stderr, "[flucahip] %s:%d %s -> %s\n", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
      return FL_ERR_LIB;                                                                                           \
    }                                                                                                              \
  } while (0)

// One-shot all-reduce of up to NSLOT doubles: every rank owns a MAILBOX in fine-grained device memory --
// two parities of nranks slots of NSLOT doubles and nranks sequence numbers each -- that its peers have mapped (hipIpc handles exchanged once
// through the control plane, or plain pointers between handles of one process).  An all-reduce is ONE single-wave kernel per rank: write my
// numbers into my slot of every peer's mailbox, release the sequence number, wait until my own mailbox holds the current sequence number of
// every rank, add the slots in rank order (the same bits on every rank).  No rendezvous through the host, no ring: at 256^3 per GPU the two
// 64-byte ncclAllReduce of a CG iteration cost as much as its kernels.  Opt-in (tuning knob "allreduce" = 1); RCCL stays the default.
struct OneShotBox {
  double             slot[2][NSLOT][NSLOT];  // [parity][rank][value]  (nranks <= NSLOT)
  unsigned long long seq[2][NSLOT];          // [parity][rank]: the all-reduce number the slot belongs to
  int                error;                  // a wait ran into its time limit
  int                pad_;
};
void launch_oneshot_allreduce(hipStream_t st, OneShotBox *const *boxes, int rank, int nranks, unsigned long long number, double *vals, int n);

struct Comm {
  enum Kind { NONE, RCCL, HOST } kind = NONE;
  // one-shot all-reduce (fl_poisson_comm_oneshot_*): my mailbox, the peers' mailboxes as this process sees them (device array), the call counter
  OneShotBox          *box = nullptr;
  OneShotBox         **peers_dev = nullptr;
  std::vector<void *>  ipc_opened;
  unsigned long long   oneshot_calls = 0;
  bool                 oneshot_ready = false;
  int             rank = 0, nranks = 1;
  ncclComm_t      nccl = nullptr;
  fl_exchange_fn  xchg = nullptr;
  fl_allreduce_fn allred = nullptr;
  void           *ctx = nullptr;
  // pinned staging for the host transport
  std::vector<double *> hsend, hrecv;
  std::vector<int64_t>  hcap;
  double               *hred = nullptr;
  int                   hredcap = 0;
  bool                  owns = true;  // false: the connection belongs to another handle (multigrid levels), never torn down here

  // same wire as `o`, own staging buffers
  void borrow(const Comm &o)
  {
    destroy();
    kind = o.kind; rank = o.rank; nranks = o.nranks; nccl = o.nccl; xchg = o.xchg; allred = o.allred; ctx = o.ctx;
    owns = false;
    // NOT the one-shot mailboxes: the call counter lives in the Comm, and two Comms counting on one mailbox would disagree about the
    // sequence numbers -- the coarse levels' reductions go through the borrowed RCCL / host wire
  }

  int exchange(hipStream_t st, const std::vector<Msg> &m)
  {
    if (m.empty()) return 0;
    if (kind == RCCL) {
      FL_NCCL(g_rccl.GroupStart());
      // a failing Send / Recv must not leave the communicator inside an open group: remember the error, close the group, report
      ncclResult_t bad = ncclSuccess;
      for (const Msg &x : m) {
        if (bad == ncclSuccess && x.send) bad = g_rccl.Send(x.send, (size_t)x.count, ncclDouble, x.peer, nccl, st);
        if (bad == ncclSuccess && x.recv) bad = g_rccl.Recv(x.recv, (size_t)x.count, ncclDouble, x.peer, nccl, st);
      }
      const ncclResult_t end = g_rccl.GroupEnd();
      if (bad != ncclSuccess || end != ncclSuccess) {
        std::fprintf(stderr, "[flucahip] halo exchange over RCCL failed: %s\n", g_rccl.GetErrorString(bad != ncclSuccess ? bad : end));
        return FL_ERR_LIB;
      }
      return 0;
    }
    if (kind == HOST) {
      const int n = (int)m.size();
      if (knob(K_comm_trace) != 0) {
        static std::atomic<long> seq{0};
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        std::fprintf(stderr, "[%.3f comm r%d #%ld] exchange %d msgs:", ts.tv_sec % 1000 + 1e-9 * ts.tv_nsec, rank, seq.fetch_add(1), n);
        for (const Msg &x : m) std::fprintf(stderr, " (peer %d stag %d rtag %d n %lld%s%s)", x.peer, x.sendtag, x.recvtag, (long long)x.count, x.send ? " S" : "", x.recv ? " R" : "");
        std::fprintf(stderr, "\n");
        std::fflush(stderr);
      }
      if ((int)hsend.size() < n) {
        hsend.resize(n, nullptr);
        hrecv.resize(n, nullptr);
        hcap.resize(n, 0);
      }
      std::vector<int>     peer(n), stag(n), rtag(n);
      std::vector<void *>  sp(n), rp(n);
      std::vector<int64_t> nb(n);
      for (int a = 0; a < n; ++a) {
        if (hcap[a] < m[a].count) {
          if (hsend[a]) (void)hipHostFree(hsend[a]);
          if (hrecv[a]) (void)hipHostFree(hrecv[a]);
          FL_HIP(hipHostMalloc((void **)&hsend[a], sizeof(double) * m[a].count));
          FL_HIP(hipHostMalloc((void **)&hrecv[a], sizeof(double) * m[a].count));
          hcap[a] = m[a].count;
        }
        if (m[a].send) FL_HIP(hipMemcpyAsync(hsend[a], m[a].send, sizeof(double) * m[a].count, hipMemcpyDeviceToHost, st));
        peer[a] = m[a].peer;
        stag[a] = m[a].sendtag;
        rtag[a] = m[a].recvtag;
        sp[a]   = m[a].send ? hsend[a] : nullptr;
        rp[a]   = m[a].recv ? hrecv[a] : nullptr;
        nb[a]   = (int64_t)sizeof(double) * m[a].count;
      }
      FL_HIP(hipStreamSynchronize(st));
      if (xchg(ctx, n, peer.data(), stag.data(), rtag.data(), sp.data(), rp.data(), nb.data()) != 0) return FL_ERR_LIB;
      for (int a = 0; a < n; ++a)
        if (m[a].recv) FL_HIP(hipMemcpyAsync(m[a].recv, hrecv[a], sizeof(double) * m[a].count, hipMemcpyHostToDevice, st));
      return 0;
    }
    return FL_ERR_ARG_WRONGSTATE;
  }

  bool loopback = false;
  int  allreduce(hipStream_t st, double *dev, int n)
  {
    if (nranks == 1 && !loopback) return 0;
    if (oneshot_ready && n <= NSLOT && knob(K_allreduce) == 1) {
      launch_oneshot_allreduce(st, peers_dev, rank, nranks, ++oneshot_calls, dev, n);
      return 0;
    }
    if (kind == RCCL) {
      FL_NCCL(g_rccl.AllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, nccl, st));
      return 0;
    }
    if (kind == HOST) {
      if (hredcap < n) {
        if (hred) (void)hipHostFree(hred);
        FL_HIP(hipHostMalloc((void **)&hred, sizeof(double) * (size_t)std::max(n, 64)));
        hredcap = std::max(n, 64);
      }
      FL_HIP(hipMemcpyAsync(hred, dev, sizeof(double) * n, hipMemcpyDeviceToHost, st));
      FL_HIP(hipStreamSynchronize(st));
      if (allred(ctx, hred, n) != 0) return FL_ERR_LIB;
      FL_HIP(hipMemcpyAsync(dev, hred, sizeof(double) * n, hipMemcpyHostToDevice, st));
      return 0;
    }
    return FL_ERR_ARG_WRONGSTATE;
  }

  void destroy_oneshot()
  {
    if (!owns) {  // a multigrid level: the mailboxes belong to the fine handle
      box = nullptr;
      peers_dev = nullptr;
      oneshot_ready = false;
      return;
    }
    for (void *p : ipc_opened) (void)hipIpcCloseMemHandle(p);
    ipc_opened.clear();
    if (peers_dev) (void)hipFree(peers_dev);
    if (box) (void)hipFree(box);
    peers_dev = nullptr;
    box = nullptr;
    oneshot_ready = false;
    oneshot_calls = 0;
  }
  void destroy()
  {
    destroy_oneshot();
    if (kind == RCCL && nccl && owns) g_rccl.CommDestroy(nccl);
    for (double *p : hsend)
      if (p) (void)hipHostFree(p);
    for (double *p : hrecv)
      if (p) (void)hipHostFree(p);
    if (hred) (void)hipHostFree(hred);
    hsend.clear();
    hrecv.clear();
    hcap.clear();
    hred = nullptr;
    hredcap = 0;
    nccl = nullptr;
    kind = NONE;
    owns = true;
  }
};

}  // namespace fl

using namespace fl;

// ------------------------------------------------------------------------------------------------ handle

struct fl_mg;

// the scalar blocks of every sweep of one kind of smoothing call (fl_cheb_smooth_padded), precomputed on the host and kept on the device
struct SmoothSeq {
  int                  nu = 0;
  bool                 guess_zero = false, jac = false, fuse = false, want = false, zero3 = false;
  double               emin = 0., emax = 0.;
  std::vector<KspScal> host;
  KspScal             *dev = nullptr;
};

struct fl_poisson {
  int         device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  Axis        ax[3];
  int         bc[6];
  double      kappa = 0.;
  fl_decomp   dec;
  int         nbr[6];        // rank across each boundary, -1 = physical boundary
  bool        wrap_local[3]; // periodic axis held by a single rank: ghosts filled by a local copy
  bool        multi = false; // more than one rank (or loopback)
  bool        loopback = false;  // FLUCA_COMM_LOOPBACK=1: a single rank sends the ghost layers of its periodic axes to ITSELF through
                                 // the communicator instead of copying them -- exercises the RCCL path on a one-GPU box
  GridP       g;
  std::vector<void *> tables;
  int64_t     ncell = 0, nface[3] = {0, 0, 0};
  size_t      padlen = 0;
  int         nv_il = 1, sx0 = 0;  // row-interleave factor of the padded vectors and the un-interleaved row length
  int         gw = 1;              // ghost layers of the padded layout around the owned block (2 on several ranks: fl_fill_ghosts_deep)
  // fused two-step Chebyshev (fl_cheb2.hip) on several ranks: the ranks' AGREED answers to "legal on my block" [0] and "legal and large
  // enough to pay" [1], -1 = not asked yet (one all-reduce per handle / multigrid level: fl_cheb2_agree)
  int         cheb2_agreed[2] = {-1, -1};
  // solver workspace (padded vectors)
  double *r = nullptr, *P0 = nullptr, *P1 = nullptr, *q = nullptr, *xp = nullptr, *w0 = nullptr, *w1 = nullptr, *w2 = nullptr;
  double *cd1 = nullptr;  // second d buffer of the fused two-step Chebyshev kernel (fl_cheb2.hip)
  void   *sv_pack[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // packed 1-D rows of the fused DIAG / ROWSUM Schur product (fl_schur_var.hip; freed with h->tables)
  double *rb = nullptr;   // where the three-step sweep from a zero guess writes the updated right-hand side; swaps roles with r afterwards
  std::vector<void *> vec_bases;
  void               *slab = nullptr;
  // placement (fl_api.hip): one arena, the five CG vectors in a window found by probing, two side pools for the rest
  void  *arena = nullptr;
  size_t arena_bytes = 0, pool_vec = 0;
  size_t vec_bytes = 0;  // device memory behind vec_bases
  struct VmmArena *vmm = nullptr;  // placement window that lives in chunk-mapped virtual memory (fl_api.hip)
  char  *pool_next[2] = {nullptr, nullptr}, *pool_end[2] = {nullptr, nullptr};
  int    pool_flip = 0;
  bool   placed = false;
  bool   poisoned = false;  // a solve ended with NaN / Inf / divergence: work vectors are zeroed before the next one
  double placed_ms[2] = {0., 0.}, placed_at = 0.;
  int                 nvec = 0;
  double *partial = nullptr;
  int     partial_stride = 0;
  double *sums = nullptr;
  unsigned *tickets = nullptr;  // [2] device-scope arrival counters of the fused finalisation (k_cg_A, k_cg_B)
  KspScal *scal = nullptr, *scal_host = nullptr;
  double  *hist = nullptr;
  int      hist_cap = 0;
  double  *fsend[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *frecv[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double  *hiface[3] = {nullptr, nullptr, nullptr}, *loface_send[3] = {nullptr, nullptr, nullptr};
  double  *xsend[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *xrecv[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // extended faces of fl_fill_ghosts_full / fl_fill_ghosts_deep
  size_t   xcap = 0;  // doubles each of them holds
  Comm     comm;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // halo exchange overlapped with k_cg_B (fl_exchange_r_begin / _end): its own stream and the two events that order it
  hipStream_t comm_stream = nullptr;
  hipEvent_t  ev_packed = nullptr, ev_ghosts = nullptr;
  hipEvent_t  ev_upload = nullptr;  // behind the last fl_poisson_upload (fl_poisson_upload_fence waits for it)
  fl_mg     *mg = nullptr;  // multigrid hierarchy, built by the first solve with FL_PC_MG (fl_mg.hip)
  std::list<SmoothSeq> smooth_seq;  // (a list: entries are never moved once a sweep has been pointed at their host copy)
};


// HIP events bracketing the dominant kernel of every iteration (fl_ksp_opts.profile); released on every return path
struct ProfEvents {
  std::vector<hipEvent_t> ev;
  int create(size_t n)
  {
    ev.reserve(n);
    for (size_t a = 0; a < n; ++a) {
      hipEvent_t e = nullptr;
      FL_HIP(hipEventCreate(&e));
      ev.push_back(e);
    }
    return 0;
  }
  // mean of the first `pairs` (start, stop) pairs
  void mean(int pairs, double *ms_out, int *count_out) const
  {
    double tot = 0.;
    int    cnt = 0;
    for (int a = 0; a < pairs && (size_t)(2 * a + 1) < ev.size(); ++a) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev[2 * a], ev[2 * a + 1]) == hipSuccess) {
        tot += t;
        ++cnt;
      }
    }
    *ms_out    = cnt ? tot / cnt : 0.;
    *count_out = cnt;
  }
  // mean of ev[stride * a + o1] - ev[stride * a + o0] over the first n groups
  void mean_of(int n, int stride, int o0, int o1, double *ms_out, int *count_out) const
  {
    double tot = 0.;
    int    cnt = 0;
    for (int a = 0; a < n && (size_t)(stride * a + o1) < ev.size(); ++a) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev[stride * a + o0], ev[stride * a + o1]) == hipSuccess) {
        tot += t;
        ++cnt;
      }
    }
    *ms_out    = cnt ? tot / cnt : 0.;
    *count_out = cnt;
  }
  ~ProfEvents()
  {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};

// shared host helpers (fl_api.hip)
int  fl_dev_alloc(fl_poisson *h, void **p, size_t bytes, bool zero);
int  fl_ensure_vec(fl_poisson *h, double **v);
void fl_vmm_destroy(fl_poisson *h);
int  fl_ensure_partials(fl_poisson *h, int nblocks);
int  fl_ensure_hist(fl_poisson *h, int nhist);
int  fl_zero_vec(fl_poisson *h, double *v);
int  fl_fill_ghosts(fl_poisson *h, double *v);
int  fl_fill_ghosts_full(fl_poisson *h, double *v);  // edges and corners too (dimension by dimension)
int  fl_exchange_r_begin(fl_poisson *h, double *r, const double *q);
int  fl_exchange_r_end(fl_poisson *h, double *r);
bool fl_any_ghost_exchange(const fl_poisson *h);
int  fl_poll_scal(fl_poisson *h);
int  fl_fill_ghosts_deep(fl_poisson *h, double *v);
int  fl_exchange_sr_begin(fl_poisson *h, const double *r, const double *sb, const double *W, double *rn);
// fl_ksp.hip
int fl_apply_tiled(fl_poisson *h, const double *xpad, double *y, int unpadded_y);
int fl_residual(fl_poisson *h, const double *x, const double *b, double *r);
int fl_residual_padded(fl_poisson *h, double *xpad, const double *bpad, double *rpad);
int fl_residual_restrict_padded(fl_poisson *h, double *xpad, const double *bpad, const double *wx, const double *wy, const double *wz, fl_poisson *hc, double *cpad);
int fl_apply_padded_dot(fl_poisson *h, double *xpad, double *ypad, double *xy);
int fl_cheb_smooth_padded(fl_poisson *h, int nu, bool jac, bool guess_zero, bool *mgdots = nullptr, const double *subq = nullptr, const double *suba_dev = nullptr);
int fl_solve_bcgs(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st);
int fl_solve_cg_sr(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st);
int fl_ksp_begin(fl_poisson *h, const fl_ksp_opts *o);
int fl_bcgs_fin_step(fl_poisson *h, int mode, int nblocks, int nslot, int nhist);
int fl_ksp_finish(fl_poisson *h, const fl_ksp_opts *o, fl_ksp_stats *st);
int fl_allreduce_max(fl_poisson *h, double *v);
int fl_allreduce_sum(fl_poisson *h, double *v);
int fl_cheb_begin(fl_poisson *h, const fl_ksp_opts *o, double emin, double emax);
int fl_cheb_fin_step(fl_poisson *h, int nblocks, int nhist);
int fl_solve_cheb(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st);
double fl_gershgorin_bound(const fl_poisson *h, bool jac);
// fl_cheb2.hip
struct Cheb2Plan {
  int nw, tiles_x, tiles, nchunk, zc, nblocks;
};
bool      fl_cheb2_usable(const fl_poisson *h);
int       fl_cheb2_agree(fl_poisson *h);  // collective on several ranks; fills h->cheb2_agreed
Cheb2Plan fl_cheb2_plan(const GridP &g);
void      fl_launch_cheb2(fl_poisson *h, const Cheb2Plan &p, bool jac, double *X0, double *X1, const double *B, double *D0, double *D1, bool mgdots = false);
// the face-interpolation rows T (kind 0 of fl_momentum's FaceT) the fused DIAG / ROWSUM Schur product reads (fl_schur_var.hip)
namespace fl {
struct SchurVarT {
  const double *w0[3], *w1[3];
  const int    *c0[3];
};
}  // namespace fl
int       fl_schur_var_apply_fused(fl_poisson *h, const fl::SchurVarT &t, const double *ainv, const double *p_pad, double *y);
void      fl_launch_cheb2_from_zero(fl_poisson *h, const Cheb2Plan &p, bool jac, double *X0, double *X1, const double *B, double *D0, double *D1, const double *subq, const double *suba_dev, double *Bw);
// fl_mg.hip
int  fl_solve_cg_mg(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st);
void fl_mg_destroy(fl_poisson *h);
void fl_mg_set_stream(fl_poisson *h);
