// k_fast64.hip - every instantiation of step_fast64_kernel (N <= 64) and its launcher.
#include "launch.hpp"
#include "step_fast64.hpp"

namespace diral {
namespace {
struct LaunchFast64 {
  const FastParams& f; const RichParams& r; const PolParams& q; dim3 g; uint32_t lds; hipStream_t s;
  template <bool FL, bool O, bool C, bool X, bool R>
  void operator()(std::integer_sequence<bool, FL, O, C, X, R>) const {
    hipLaunchKernelGGL((step_fast64_kernel<FL, O, C, X, R>), g, dim3(256), lds, s, f, r, q);
  }
};
}  // namespace

hipError_t launch_fast64(const FastParams& f, const RichParams& r, const KernelSel& k, int B, hipStream_t s) {
  static const PolParams no_policy{};
  const LaunchFast64 l{f, r, no_policy, dim3(B), fast_lds_layout(f.K, f.A, k.rich, k.out64, k.flat, k.ch || k.extra).total, s};
#ifdef DIRAL_FAST_BENCH_ONLY
  // tuning builds (profiles/build_variant.sh): only the instantiations the C2 / C4 bench lines run - seconds to compile
  if (!k.flat || k.out64 || k.ch || k.extra) return hipErrorInvalidValue;
  bool_dispatch(l, std::integer_sequence<bool, true, false, false, false>{}, k.rich);
#else
  bool_dispatch(l, std::integer_sequence<bool>{}, k.flat, k.out64, k.ch, k.extra, k.rich);
#endif
  return hipGetLastError();
}

// the POL instantiations (policy epilogue): the flat highway, my_step, no EXTRA switches, RICH
hipError_t launch_fast64_policy(const FastParams& f, const RichParams& r, const PolParams& q, bool out64, int B, hipStream_t s) {
  const uint32_t lds = fast_lds_layout(f.K, f.A, true, out64, true, false).total;
  if (out64) hipLaunchKernelGGL((step_fast64_kernel<true, true, false, false, true, true>), dim3(B), dim3(256), lds, s, f, r, q);
  else hipLaunchKernelGGL((step_fast64_kernel<true, false, false, false, true, true>), dim3(B), dim3(256), lds, s, f, r, q);
  return hipGetLastError();
}

// ... K slots per launch (PolParams::K > 1): step_fast64_slots_kernel, the same body with the env kept on the chip
hipError_t launch_fast64_slots(const FastParams& f, const RichParams& r, const PolParams& q, bool out64, int B, hipStream_t s) {
  const uint32_t lds = fast_lds_layout(f.K, f.A, true, out64, true, false, true).total;
  if (out64) hipLaunchKernelGGL((step_fast64_slots_kernel<true, true, false, false, true, true>), dim3(B), dim3(256), lds, s, f, r, q);
  else hipLaunchKernelGGL((step_fast64_slots_kernel<true, false, false, false, true, true>), dim3(B), dim3(256), lds, s, f, r, q);
  return hipGetLastError();
}
}  // namespace diral
