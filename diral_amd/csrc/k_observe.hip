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

// observe_kernel's dynamic LDS grows with N * K (ring rows + histogram rows: ~66 KB at N = 200 / K = 40, ~90 KB at
// N = 256 / K = 64) - past the 64 KB a kernel may use without asking; called once per handle at create
hipError_t set_attr_observe(int N, int K) {
  const int lds = (int)observe_lds_layout(N, K).total;
  hipError_t st = hipSuccess;
  auto one = [&](const void* fn) { if (st == hipSuccess) st = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); };
  one(reinterpret_cast<const void*>(observe_kernel<false, false>));
  one(reinterpret_cast<const void*>(observe_kernel<false, true>));
  one(reinterpret_cast<const void*>(observe_kernel<true, false>));
  one(reinterpret_cast<const void*>(observe_kernel<true, true>));
  return st;
}

hipError_t launch_observe(const ObserveParams& p, const RichParams& r, bool flat, bool out64, int B, hipStream_t s) {
  const LaunchObserve l{p, r, dim3(B), observe_lds_layout(p.N, p.K).total, s};
  bool_dispatch(l, std::integer_sequence<bool>{}, flat, out64);
  return hipGetLastError();
}
}  // namespace diral
