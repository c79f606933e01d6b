// fp32 instantiation (production precision for BASELINE config "cheetah run fp32").
#include "step_kernel.hip.h"
namespace dmc {
hipError_t launch_step_f32(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<float>& o,
                           const int* g_mi, const float* g_mr, const int* g_mc, const StepIO<float>& io, int nstep, int legacy, int mode, int outmask, int nsub) {
  return launch_step_t<float>(g, stream, d_layout, o, g_mi, g_mr, g_mc, io, nstep, legacy, mode, outmask, nsub);
}
}  // namespace dmc
