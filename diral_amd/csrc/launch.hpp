// launch.hpp - host-side launchers of the step kernels.  Each kernel family lives in its own
// translation unit (k_fast64.hip, k_wide2.hip, k_wide4.hip, k_general.hip) so that hipcc builds
// them in parallel; diral_env.hip only sees these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "common.hpp"
#include "rich_out.hpp"

namespace diral {

struct FastParams;   // step_fast64.hpp
struct PolParams;    // policy_device.hpp
struct ObserveParams;   // observe_kernel.hpp

// which instantiation of a specialised kernel to launch
struct KernelSel {
  bool flat;    // every pos_y == 0 (step_fast64 only; step_wide requires it)
  bool out64;   // float64 outputs
  bool full;    // N == 64 * VPL (step_wide only)
  bool ch;      // my_step_ch
  bool extra;   // my_step_design / arrival stamps / trace replay compiled in
  bool rich;    // rich_out.hpp output tail
  bool packed;  // step_wide: the packed table form (codes + ages), else the (seq, age) plane
};

hipError_t launch_fast64(const FastParams& f, const RichParams& r, const KernelSel& k, int B, hipStream_t s);
hipError_t launch_fast64_policy(const FastParams& f, const RichParams& r, const PolParams& q, bool out64, int B, hipStream_t s);
hipError_t launch_fast64_slots(const FastParams& f, const RichParams& r, const PolParams& q, bool out64, int B, hipStream_t s);
hipError_t launch_wide2(const FastParams& f, const RichParams& r, const KernelSel& k, int B, hipStream_t s);
hipError_t launch_wide4(const FastParams& f, const RichParams& r, const KernelSel& k, int B, hipStream_t s);
hipError_t set_attr_wide2(int A, int K);
hipError_t set_attr_wide4(int A, int K);
hipError_t launch_observe(const ObserveParams& p, const RichParams& r, bool flat, bool out64, int B, hipStream_t s);
hipError_t set_attr_observe(int N, int K);
hipError_t launch_general(int vpl, bool fast, const StepParams& p, uint32_t lds, hipStream_t s);
hipError_t set_attr_general(int vpl, uint32_t lds);
// num_users > 256 / num_channels > 256 / num_bins > 64 (step_large.hpp): search + merge + histogram launches
hipError_t launch_large(const StepParams& p, const LargeScratch& g, hipStream_t s);
hipError_t set_attr_large(int N, int A, int K);
uint32_t large_lds_bytes(int N, int A, int K);   // the largest workgroup's LDS (must stay within 160 KB)

// run-time bools -> template arguments: f(std::integer_sequence<bool, ...>) is called with the
// values as a type
template <typename F, bool... Bs>
inline void bool_dispatch(F&& f, std::integer_sequence<bool, Bs...> seq) { f(seq); }
template <typename F, bool... Bs, typename... Rest>
inline void bool_dispatch(F&& f, std::integer_sequence<bool, Bs...>, bool b, Rest... rest) {
  if (b) bool_dispatch(f, std::integer_sequence<bool, Bs..., true>{}, rest...);
  else bool_dispatch(f, std::integer_sequence<bool, Bs..., false>{}, rest...);
}

}  // namespace diral
