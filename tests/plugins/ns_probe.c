/* ns_probe.c -- a second NS type, written the way a type is written against the reference's table
 * (fluca/include/fluca/private/nsimpl.h:21-31; constructor shape of cnlinear.c:164-187): "cnprobe" derives from NSCNLINEAR and
 * replaces ONE slot, formfunction.  Its version calls the parent's and then scales the momentum right-hand side by a factor
 * the test chooses, counting its calls -- so a step through NSStep shows whether the table is what the step goes through.
 * Test code (tests/test_host_plugins.py builds it with gcc against include/fluca_host_impl.h); not part of the product. */
#include "../../include/fluca_host_impl.h"

static int    probe_function_calls = 0, probe_jacobian_calls[2] = {0, 0};
static double probe_scale = 1.;
static FlErrorCode (*parent_formfunction)(NS, const NSVec *, NSVec *)            = 0;
static FlErrorCode (*parent_formjacobian)(NS, const NSVec *, NSMat, NSFormJacobianType) = 0;

static FlErrorCode NSFormFunction_Probe(NS ns, const NSVec *x, NSVec *f)
{
  ++probe_function_calls;
  FlErrorCode rc = parent_formfunction(ns, x, f);
  if (rc) return rc;
  if (probe_scale != 1.) {
    int64_t sz[4];
    rc = NSGetLocalSizes(ns, sz);
    if (rc) return rc;
    const int abi = fl_vec_lincomb(ns->poisson, 3 * sz[0], probe_scale, f->v, 0., 0, f->v); /* momrhs *= scale */
    if (abi) return -abi;
  }
  return 0;
}
static FlErrorCode NSFormJacobian_Probe(NS ns, const NSVec *x, NSMat J, NSFormJacobianType type)
{
  ++probe_jacobian_calls[type == NS_UPDATE_JACOBIAN];
  return parent_formjacobian(ns, x, J, type);
}

FlErrorCode NSCreate_Probe(NS ns)
{
  FlErrorCode rc = NSCreate_CNLinear(ns); /* the parent's constructor fills all nine slots */
  if (rc) return rc;
  parent_formfunction   = ns->ops->formfunction;
  parent_formjacobian   = ns->ops->formjacobian;
  ns->ops->formfunction = NSFormFunction_Probe;
  ns->ops->formjacobian = NSFormJacobian_Probe;
  return 0;
}

FlErrorCode ProbeRegister(void) { return NSRegister("cnprobe", NSCreate_Probe); }
void ProbeSetScale(double s) { probe_scale = s; }
int  ProbeFunctionCalls(void) { return probe_function_calls; }
int  ProbeJacobianCalls(int update) { return probe_jacobian_calls[update != 0]; }
int  ProbeSlotCount(void) { return (int)(sizeof(struct _NSOps) / sizeof(void *)); }
int  ProbeMeshSlotCount(void) { return (int)(sizeof(struct _MeshOps) / sizeof(void *)); }
