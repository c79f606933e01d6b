// fp32 instantiation (production precision for BASELINE config "cheetah run fp32"): the generic kernels and the
// model-specialised ones of the small models; the large models' live in step_kernels_f32_ilp.hip.
#define DMC_UNIT_STD 1
#define DMC_STATIC_FEATURES 0   // the small models' specialised kernels: no probe / step+forward / implicitfast (step_core.h kFeat)
#include "step_kernel.hip.h"
namespace dmc {
hipError_t launch_step_f32_ilp(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<float>& o,
                               const int* g_mi, const float* g_mr, const int* g_mc, const StepIO<float>& io, int nstep, int legacy, int mode, int outmask, int nsub);
hipError_t launch_step_f32(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<float>& o,
                           const int* g_mi, const float* g_mr, const int* g_mc, const StepIO<float>& io, int nstep, int legacy, int mode, int outmask, int nsub) {
#if DMC_NSTATIC > 0
#define DMC_X(SID, LPE) if (g.static_id == SID && g.lpe == LPE) return launch_step_f32_ilp(g, stream, d_layout, o, g_mi, g_mr, g_mc, io, nstep, legacy, mode, outmask, nsub);
  DMC_STATIC_INSTANCES_ILP(DMC_X)
#undef DMC_X
#endif
  return launch_step_t<float>(g, stream, d_layout, o, g_mi, g_mr, g_mc, io, nstep, legacy, mode, outmask, nsub);
}
}  // namespace dmc
