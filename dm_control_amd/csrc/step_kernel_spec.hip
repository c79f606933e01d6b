// step_kernel_spec.hip -- a model-specialised step kernel built ON DEMAND for one model (dm_control_amd/specialise.py).
//
// The library bakes specialised instantiations for eight assets (build.py _STATIC_MODELS); every other MJCF -- a composer
// recompile with another body count, a user model -- ran the generic kernel, 2.0 - 2.7 x slower (profiles/r05_generic_vs_static.log).
// This unit is compiled per model with its LDS layout as a compile-time constant (-DDMC_LAYOUTS_HEADER=<generated header>:
// the same header format as static_layouts.gen.h, one model) into a small shared object that the library loads next to a
// batch (dmc_batch_attach_specialised) and launches instead of its generic kernel.  Same StepCore, same flags as the
// library unit the model would have been baked into (specialise.py), so the results are those of a baked twin.
//   -DDMC_SPEC_PRECISION=32|64   -DDMC_STATIC_FEATURES=0|1 (0: the lean form of the small models, step_core.h kFeat)
#define DMC_UNIT_ILP 1      // (instance list = DMC_STATIC_INSTANCES_ILP of the generated header; no generic kernel in this unit)
#include "step_kernel.hip.h"
using namespace dmc;

#if DMC_SPEC_PRECISION == 64
typedef double spec_real;
#else
typedef float spec_real;
#endif

extern "C" const StepLayout* dmc_spec_layout() { static const StepLayout L = DMC_STATIC_LAYOUT_0; return &L; }
// what must agree with the library that loads this object: struct sizes, model blob version, precision, lanes, features
extern "C" void dmc_spec_info(int* out) {
  out[0] = (int)sizeof(StepLayout); out[1] = (int)sizeof(StepOpts<spec_real>); out[2] = (int)sizeof(StepIO<spec_real>);
  out[3] = DMC_MODEL_VERSION; out[4] = DMC_SPEC_PRECISION; out[5] = DMC_SPEC_LPE; out[6] = DMC_STATIC_FEATURES; out[7] = (int)sizeof(LaunchGeom);
}
extern "C" int dmc_spec_launch(const LaunchGeom* g, void* stream, const StepLayout* d_layout, const StepOpts<spec_real>* o,
                               const int* g_mi, const spec_real* g_mr, const int* g_mc, const StepIO<spec_real>* io, int nstep,
                               int legacy, int mode, int outmask, int nsub) {
  LaunchGeom gg = *g;
  gg.static_id = 0;
  return (int)launch_step_t<spec_real>(gg, (hipStream_t)stream, d_layout, *o, g_mi, g_mr, g_mc, *io, nstep, legacy, mode, outmask, nsub);
}
#ifdef DMC_TASK_HEADER
// the size of the task's argument block this kernel was generated with (dmc_batch_set_task_args checks it)
extern "C" int dmc_spec_task_args_bytes() { return (int)sizeof(dmc_task::PostArgs); }
#endif
