// fp32 model-specialised kernels of the LARGE models (DMC_STATIC_INSTANCES_ILP: humanoid_CMU, the CMU walker on the
// floor, soccer 2v2), compiled with -mllvm -amdgpu-sched-strategy=max-ilp (build.py says why).
#define DMC_UNIT_ILP 1
#include "step_kernel.hip.h"
namespace dmc {
hipError_t launch_step_f32_ilp(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<float>& o,
                               const int* g_mi, const float* g_mr, const int* g_mc, const StepIO<float>& io, int nstep, int legacy, int mode, int outmask, int nsub) {
  return launch_step_t<float>(g, stream, d_layout, o, g_mi, g_mr, g_mc, io, nstep, legacy, mode, outmask, nsub);
}
}  // namespace dmc
