// k_observe.hip - the instantiations of observe_kernel (stand-alone obtain_state) and their launcher.
#include "launch.hpp"
#include "observe_kernel.hpp"

namespace diral {
namespace {
struct LaunchObserve {
  const ObserveParams& p; const RichParams& r; dim3 g; uint32_t lds; hipStream_t s;
  template <bool FL, bool O>
  void operator()(std::integer_sequence<bool, FL, O>) const {
    hipLaunchKernelGGL((observe_kernel<FL, O>), g, dim3(kObserveThreads), lds, s, p, r);
  }
};
}  // namespace

hipError_t launch_observe(const ObserveParams& p, const RichParams& r, bool flat, bool out64, int B, hipStream_t s) {
  const LaunchObserve l{p, r, dim3(B), observe_lds_layout(p.N, p.K).total, s};
  bool_dispatch(l, std::integer_sequence<bool>{}, flat, out64);
  return hipGetLastError();
}
}  // namespace diral
