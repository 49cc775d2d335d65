// fl_knobs.h -- every build-specific switch of libflucahip.so, in one table.
//
// PUBLIC knobs (fl_tuning_set / fl_tuning_get, include/fluca_hip.h): process-wide integers, atomics -- handles of several host threads read them
// while they run; set them before the solves they should affect.  Their initial value is the table's default or, if set, the environment
// variable FLUCA_<NAME IN CAPITALS>; the environment is read ONCE, in one place (knob_table_init, fl_api.hip).
//
// VARIANT switches choose between a shipped code path and a superseded or experimental one (A/B material of tools/kbench.py and
// tools/experiments/).  In the product build they are the shipped constant -- the other path is not compiled, its kernels are not in the
// library; a build with -DFL_KBENCH_VARIANTS (FL_KBENCH_VARIANTS=1 python -m fluca_amd.build) turns them into knobs of the same table, with the
// same FLUCA_* environment names the experiment scripts use.
#pragma once
#include <atomic>

#ifndef FL_DEFAULT_GAP
#define FL_DEFAULT_GAP 0
#endif
#ifndef FL_DEFAULT_INTERLEAVE
#define FL_DEFAULT_INTERLEAVE 0
#endif

// name, default
#define FL_PUBLIC_KNOBS(X)                                                                                                                   \
  X(cheb_fuse, 1)         /* two Chebyshev steps per sweep: 0 never, 1 where it pays (>= 32768 cells per rank), 2 wherever legal */          \
  X(placement, 0)         /* 1: the first solve of a large handle runs the placement search of fl_poisson_tune_placement */                  \
  X(cg_xbatch, 1)         /* CG: both x-updates of an iteration pair on the odd iteration */                                                  \
  X(cheb_zero3, 1)        /* a smoother call from a zero guess on one rank: its first three steps in one sweep (fl_cheb2.hip, Z) */          \
  X(mg_prolong, 1)        /* FL_PC_MG: 1 tri-linear prolongation, 0 piecewise constant */                                                     \
  X(mg_flexible, 1)       /* FL_PC_MG: 1 Polak-Ribiere beta (flexible CG), 0 KSPCG's */                                                      \
  X(mg_coarse, 1)         /* FL_PC_MG: 1 a coarsest level of <= 4096 cells on one rank is solved by one workgroup */                          \
  X(mg_post_smooth, 0)    /* FL_PC_MG: smoothing steps AFTER the coarse correction; 0 = as many as before it (fl_ksp_opts.mg_smooth_its) */           \
  X(schur_var_fused, 1)   /* PCABF schurainv DIAG / ROWSUM on one rank: 1 the product S p in one pass (fl_schur_var.hip), 0 the composition of seven */   \
  X(overlap, 1)           /* several ranks: the exchange of the new residual hidden behind the update kernel; 0 sequential (A/B, tests) */   \
  X(comm_loopback, 0)     /* 1: a single rank sends the ghost layers of its periodic axes to itself through the communicator (tests) */      \
  X(comm_trace, 0)        /* 1: every host-staged exchange / smoother stage on stderr; 2: with a stream wait per stage */                     \
  X(ghost_width, 0)       /* 0 automatic (2 where a neighbouring rank exists, else 1); 1 / 2 forced (tests) */                                \
  X(placement_verbose, 0) /* the placement search narrates on stderr */                                                                       \
  X(placement_vmm, 1)     /* placement arenas in chunk-mapped virtual memory (everything but the chosen window is released) */               \
  X(allreduce, 0)         /* several ranks: 0 RCCL / the host callbacks, 1 the one-shot all-reduce through peer-mapped buffers (fl_oneshot.h) */

#ifdef FL_KBENCH_VARIANTS
#define FL_VARIANT_KNOBS(X)                                                                                                     \
  X(cheb_staged, 1) X(project_fused, 3) X(cg_variant, -1) X(cg_qb, 1) X(cgbq_chunks, 0) X(fusedfin, 1)                          \
  X(interleave, FL_DEFAULT_INTERLEAVE) X(gap, FL_DEFAULT_GAP) X(slab, 0) X(bcgs_variant, -1) X(cheb2_nw, 0) X(cheb2_nchunk, 0)  \
  X(ibm_spread, 0) X(mg_fused_dots, 1) X(mg_fused_restrict, 1) X(mg_prolong_tile, 2) X(mom_chunks, 0) X(mom_kernel, 3)          \
  X(mom_nt, 1) X(mom_order, 1) X(mom_pw, 3) X(mom_pw_blocks, 0) X(cga_target, 0) X(print_ptrs, 0)
#else
#define FL_VARIANT_KNOBS(X)
#endif

namespace fl {

enum Knob : int {
#define X(n, d) K_##n,
  FL_PUBLIC_KNOBS(X) FL_VARIANT_KNOBS(X)
#undef X
      K_COUNT
};

int         knob(Knob k);              // relaxed atomic load; the first call of the process fills the table from the environment
void        knob_set(Knob k, int v);
int         knob_find(const char *name);  // index in the table, -1 if there is no such knob
const char *knob_name(int k);

#ifdef FL_KBENCH_VARIANTS
#define FL_VARIANT(n, d) (::fl::knob(::fl::K_##n))
const char *variant_env(const char *name);  // string / floating-point valued experiment switches (FLUCA_CG_PLAN = "ry,nw,nchunk", ...)
#else
#define FL_VARIANT(n, d) (d)
inline const char *variant_env(const char *) { return nullptr; }
#endif

}  // namespace fl
