// k_wide2.hip - every instantiation of step_wide_kernel<2, ...> (64 < N <= 128) and its launcher.
#include "launch.hpp"
#include "step_wide.hpp"

namespace diral {
namespace {
constexpr int V = 2;
struct LaunchWide {
  const FastParams& f; const RichParams& r; dim3 g; uint32_t lds; hipStream_t s;
  template <bool O, bool F, bool C, bool X, bool R, bool P>
  void operator()(std::integer_sequence<bool, O, F, C, X, R, P>) const {
    hipLaunchKernelGGL((step_wide_kernel<V, O, F, C, X, R, P>), g, dim3(64 * wide_waves(V)), lds, s, f, r);
  }
};
struct AttrWide {
  int lds, lds_packed; hipError_t* st;
  template <bool O, bool F, bool C, bool X, bool R, bool P>
  void operator()(std::integer_sequence<bool, O, F, C, X, R, P>) const {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(step_wide_kernel<V, O, F, C, X, R, P>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, P ? lds_packed : lds);
    if (e != hipSuccess) *st = e;
  }
};
}  // namespace

hipError_t launch_wide2(const FastParams& f, const RichParams& r, const KernelSel& k, int B, hipStream_t s) {
  const LaunchWide l{f, r, dim3(B), wide_lds_layout(V, f.A, f.K, k.packed).total, s};
#ifdef DIRAL_WIDE_BENCH_ONLY
  // tuning builds (profiles/build_variant.sh): only the instantiations the C5 bench line runs - seconds to compile
  if (k.out64 || !k.full || k.ch || k.extra) return hipErrorInvalidValue;
  bool_dispatch(l, std::integer_sequence<bool, false, true, false, false>{}, k.rich, k.packed);
#else
  bool_dispatch(l, std::integer_sequence<bool>{}, k.out64, k.full, k.ch, k.extra, k.rich, k.packed);
#endif
  return hipGetLastError();
}

hipError_t set_attr_wide2(int A, int K) {
  hipError_t st = hipSuccess;
  const AttrWide a{(int)wide_lds_layout(V, A, K, false).total, (int)wide_lds_layout(V, A, K, true).total, &st};
#ifdef DIRAL_WIDE_BENCH_ONLY
  for (int m = 0; m < 4; ++m)
    bool_dispatch(a, std::integer_sequence<bool, false, true, false, false>{}, (m & 1) != 0, (m & 2) != 0);
#else
  for (int m = 0; m < 64; ++m)
    bool_dispatch(a, std::integer_sequence<bool>{}, (m & 1) != 0, (m & 2) != 0, (m & 4) != 0, (m & 8) != 0, (m & 16) != 0, (m & 32) != 0);
#endif
  return st;
}
}  // namespace diral
