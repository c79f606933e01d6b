// fp64 instantiation (parity mode; compiled with -ffp-contract=off).
#include "step_kernel.hip.h"
namespace dmc {
hipError_t launch_step_f64(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<double>& o,
                           const int* g_mi, const double* g_mr, const int* g_mc, const StepIO<double>& io, int nstep, int legacy, int mode, int outmask, int nsub) {
  return launch_step_t<double>(g, stream, d_layout, o, g_mi, g_mr, g_mc, io, nstep, legacy, mode, outmask, nsub);
}
}  // namespace dmc
