/*
 * inproc_comm.c -- test transport: the fl_exchange_fn / fl_allreduce_fn pair of fl_poisson_comm_init_host (include/fluca_hip.h)
 * for SEVERAL RANKS INSIDE ONE PROCESS, one host thread per rank (what fluca_hip.h promises: "a handle is driven by one host
 * thread; different handles are independent").  Plain C + pthreads, no Python in the callbacks: the threads that drive the
 * handles may be Python threads (ctypes drops the GIL around every foreign call), the wire never takes it.
 *
 * Messages are copied into a heap buffer and queued at the destination (matched by source rank and tag, FIFO per pair), so a
 * sender never waits for its receiver; the all-reduce adds the ranks' contributions in rank order, once, and hands every rank
 * the same bits (what MPI_Allreduce guarantees on one communicator, what the gloo test transport does across processes).
 * A rank that fails calls inproc_abort: every wait then returns an error instead of hanging the others.
 *
 * Test infrastructure (tests/inproc.py loads it); not part of the product.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct Msg {
  int         src, tag;
  int64_t     nbytes;
  void       *data;
  struct Msg *next;
} Msg;

typedef struct World {
  int             n;
  pthread_mutex_t mu;
  pthread_cond_t  cv;
  Msg           **head, **tail; /* inbox per destination rank */
  /* all-reduce: contributions per rank, the sum, arrival / departure counters of a reusable two-phase barrier */
  double *contrib, *sum;
  int     cap, arrived, departing, red_n;
  long    generation;
  int     aborted;
  double  timeout_s;
  /* statistics (under mu) */
  long *n_allreduce, *n_exchange, *n_msgs;
  long long *bytes_sent;
} World;

typedef struct Ctx {
  World *w;
  int    rank;
} Ctx;

World *inproc_world_create(int nranks, double timeout_s)
{
  if (nranks < 1) return NULL;
  World *w = (World *)calloc(1, sizeof(World));
  if (!w) return NULL;
  w->n         = nranks;
  w->timeout_s = timeout_s > 0. ? timeout_s : 120.;
  pthread_mutex_init(&w->mu, NULL);
  pthread_condattr_t ca;
  pthread_condattr_init(&ca);
  pthread_condattr_setclock(&ca, CLOCK_MONOTONIC);
  pthread_cond_init(&w->cv, &ca);
  pthread_condattr_destroy(&ca);
  w->head        = (Msg **)calloc(nranks, sizeof(Msg *));
  w->tail        = (Msg **)calloc(nranks, sizeof(Msg *));
  w->cap         = 64;
  w->contrib     = (double *)calloc((size_t)nranks * w->cap, sizeof(double));
  w->sum         = (double *)calloc(w->cap, sizeof(double));
  w->n_allreduce = (long *)calloc(nranks, sizeof(long));
  w->n_exchange  = (long *)calloc(nranks, sizeof(long));
  w->n_msgs      = (long *)calloc(nranks, sizeof(long));
  w->bytes_sent  = (long long *)calloc(nranks, sizeof(long long));
  return w;
}

void inproc_world_destroy(World *w)
{
  if (!w) return;
  for (int r = 0; r < w->n; ++r)
    for (Msg *m = w->head[r]; m;) {
      Msg *nx = m->next;
      free(m->data);
      free(m);
      m = nx;
    }
  free(w->head);
  free(w->tail);
  free(w->contrib);
  free(w->sum);
  free(w->n_allreduce);
  free(w->n_exchange);
  free(w->n_msgs);
  free(w->bytes_sent);
  pthread_cond_destroy(&w->cv);
  pthread_mutex_destroy(&w->mu);
  free(w);
}

Ctx *inproc_ctx_create(World *w, int rank)
{
  if (!w || rank < 0 || rank >= w->n) return NULL;
  Ctx *c  = (Ctx *)malloc(sizeof(Ctx));
  c->w    = w;
  c->rank = rank;
  return c;
}
void inproc_ctx_destroy(Ctx *c) { free(c); }

void inproc_abort(World *w)
{
  pthread_mutex_lock(&w->mu);
  w->aborted = 1;
  pthread_cond_broadcast(&w->cv);
  pthread_mutex_unlock(&w->mu);
}
int inproc_aborted(World *w) { return w->aborted; }

/* counters of one rank: {all-reduces, exchanges, messages sent, bytes sent} */
void inproc_stats(World *w, int rank, long long out[4])
{
  pthread_mutex_lock(&w->mu);
  out[0] = w->n_allreduce[rank];
  out[1] = w->n_exchange[rank];
  out[2] = w->n_msgs[rank];
  out[3] = w->bytes_sent[rank];
  pthread_mutex_unlock(&w->mu);
}

static void deadline(World *w, struct timespec *ts)
{
  clock_gettime(CLOCK_MONOTONIC, ts);
  const double t = w->timeout_s;
  ts->tv_sec += (time_t)t;
  ts->tv_nsec += (long)((t - (double)(time_t)t) * 1e9);
  if (ts->tv_nsec >= 1000000000L) {
    ts->tv_sec += 1;
    ts->tv_nsec -= 1000000000L;
  }
}

/* mu held; returns 0, or 1 after abort / timeout (the world is then aborted for everybody) */
static int wait_cv(World *w, const struct timespec *until)
{
  if (w->aborted) return 1;
  if (pthread_cond_timedwait(&w->cv, &w->mu, until) != 0) {
    w->aborted = 1;
    pthread_cond_broadcast(&w->cv);
    return 1;
  }
  return w->aborted;
}

/* fl_exchange_fn */
int inproc_exchange(void *vctx, int nmsg, const int *peer, const int *sendtag, const int *recvtag, void *const *send, void *const *recv, const int64_t *nbytes)
{
  Ctx   *c = (Ctx *)vctx;
  World *w = c->w;
  for (int a = 0; a < nmsg; ++a) {
    if (!send[a]) continue;
    if (peer[a] < 0 || peer[a] >= w->n) return 2;
    Msg *m    = (Msg *)malloc(sizeof(Msg));
    m->src    = c->rank;
    m->tag    = sendtag[a];
    m->nbytes = nbytes[a];
    m->data   = malloc((size_t)nbytes[a] > 0 ? (size_t)nbytes[a] : 1);
    m->next   = NULL;
    memcpy(m->data, send[a], (size_t)nbytes[a]);
    pthread_mutex_lock(&w->mu);
    if (w->tail[peer[a]]) w->tail[peer[a]]->next = m;
    else w->head[peer[a]] = m;
    w->tail[peer[a]] = m;
    w->n_msgs[c->rank] += 1;
    w->bytes_sent[c->rank] += nbytes[a];
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
  }
  struct timespec until;
  deadline(w, &until);
  pthread_mutex_lock(&w->mu);
  w->n_exchange[c->rank] += 1;
  for (int a = 0; a < nmsg; ++a) {
    if (!recv[a]) continue;
    for (;;) {
      Msg *prev = NULL, *m = w->head[c->rank];
      while (m && !(m->src == peer[a] && m->tag == recvtag[a])) {
        prev = m;
        m    = m->next;
      }
      if (m) {
        if (prev) prev->next = m->next;
        else w->head[c->rank] = m->next;
        if (w->tail[c->rank] == m) w->tail[c->rank] = prev;
        const int ok = m->nbytes == nbytes[a];
        if (ok) memcpy(recv[a], m->data, (size_t)nbytes[a]);
        free(m->data);
        free(m);
        if (!ok) { /* the two sides disagree about a message size: a plan mismatch, fail everybody */
          w->aborted = 1;
          pthread_cond_broadcast(&w->cv);
          pthread_mutex_unlock(&w->mu);
          return 3;
        }
        break;
      }
      if (wait_cv(w, &until)) {
        pthread_mutex_unlock(&w->mu);
        return 1;
      }
    }
  }
  pthread_mutex_unlock(&w->mu);
  return 0;
}

/* fl_allreduce_fn: in-place sum of n doubles over the ranks, added in rank order, the same bits on every rank */
int inproc_allreduce(void *vctx, double *vals, int n)
{
  Ctx   *c = (Ctx *)vctx;
  World *w = c->w;
  if (n < 0) return 2;
  struct timespec until;
  deadline(w, &until);
  pthread_mutex_lock(&w->mu);
  /* the previous all-reduce must have been left by everybody before its buffers are reused */
  while (w->departing > 0)
    if (wait_cv(w, &until)) {
      pthread_mutex_unlock(&w->mu);
      return 1;
    }
  w->n_allreduce[c->rank] += 1;
  if (w->arrived == 0) {
    w->red_n = n;
    if (n > w->cap) { /* the first to arrive grows the staging (nobody is inside: departing == 0, arrived == 0) */
      free(w->contrib);
      free(w->sum);
      w->cap     = n;
      w->contrib = (double *)calloc((size_t)w->n * w->cap, sizeof(double));
      w->sum     = (double *)calloc(w->cap, sizeof(double));
    }
  } else if (w->red_n != n) { /* ranks disagree about the length: a protocol mismatch */
    w->aborted = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    return 3;
  }
  memcpy(w->contrib + (size_t)c->rank * w->cap, vals, sizeof(double) * (size_t)n);
  const long gen = w->generation;
  if (++w->arrived == w->n) {
    for (int i = 0; i < n; ++i) {
      double s = 0.;
      for (int r = 0; r < w->n; ++r) s += w->contrib[(size_t)r * w->cap + i];
      w->sum[i] = s;
    }
    w->arrived   = 0;
    w->departing = w->n;
    w->generation += 1;
    pthread_cond_broadcast(&w->cv);
  } else {
    while (w->generation == gen)
      if (wait_cv(w, &until)) {
        pthread_mutex_unlock(&w->mu);
        return 1;
      }
  }
  memcpy(vals, w->sum, sizeof(double) * (size_t)n);
  if (--w->departing == 0) pthread_cond_broadcast(&w->cv);
  pthread_mutex_unlock(&w->mu);
  return 0;
}

int inproc_barrier(Ctx *c)
{
  double z = 0.;
  return inproc_allreduce(c, &z, 1);
}
