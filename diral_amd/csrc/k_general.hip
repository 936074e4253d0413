// k_general.hip - the general fused step kernel (step_kernel.hpp): any configuration, N <= 256.
#include "launch.hpp"
#include "step_kernel.hpp"

namespace diral {
namespace {
template <int VPL, bool FAST>
hipError_t launch_step(const StepParams& p, uint32_t lds, hipStream_t s) {
  hipLaunchKernelGGL((step_kernel<VPL, FAST>), dim3(p.B), dim3(Geo<VPL>::THREADS), lds, s, p);
  return hipGetLastError();
}
template <int VPL>
hipError_t set_lds_attr(uint32_t lds) {
  hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(step_kernel<VPL, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (r != hipSuccess) return r;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(step_kernel<VPL, false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
}  // namespace

hipError_t launch_general(int vpl, bool fast, const StepParams& p, uint32_t lds, hipStream_t s) {
  switch (vpl) {
    case 1: return fast ? launch_step<1, true>(p, lds, s) : launch_step<1, false>(p, lds, s);
    case 2: return fast ? launch_step<2, true>(p, lds, s) : launch_step<2, false>(p, lds, s);
    default: return fast ? launch_step<4, true>(p, lds, s) : launch_step<4, false>(p, lds, s);
  }
}

hipError_t set_attr_general(int vpl, uint32_t lds) {
  switch (vpl) {
    case 1: return set_lds_attr<1>(lds);
    case 2: return set_lds_attr<2>(lds);
    default: return set_lds_attr<4>(lds);
  }
}
}  // namespace diral
