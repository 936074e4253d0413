// k_large.hip - the step for num_users > 256 / num_channels > 256 / num_bins > 64 (step_large.hpp): three launches.
#include "launch.hpp"
#include "step_large.hpp"

namespace diral {

hipError_t launch_large(const StepParams& p, const LargeScratch& g, hipStream_t s) {
  const bool do_step = p.mode != kModeObserve;
  const bool piggy = (p.flags & DIRAL_F_ADD_POSDIST_PIGGY) != 0;
  const bool want_hist = piggy && p.posdist_type == 2 && p.state_out != nullptr && p.off_hist >= 0;
  if (do_step || p.state_out)
    hipLaunchKernelGGL(large_search_kernel, dim3(p.B), dim3(large_search_threads(p.N)), large_lds_layout(p.N, p.A).total, s, p, g);
  if (do_step && piggy) {
    if (p.N <= 1024) {                                                // two / four columns per wave, keys in registers
      const int nc = p.N <= 256 ? 4 : 2;   // (four columns at N <= 512 measured slower: 144 VGPRs, three waves per SIMD - 6.1 -> 6.7 ms at 512 / 64)
      const unsigned nblk = (unsigned)((((p.N + nc - 1) / nc) + 3) >> 2);
      const dim3 grid((unsigned)p.B * nblk), block(256);
      if (p.N <= 256) hipLaunchKernelGGL((large_mergen_kernel<4, 4>), grid, block, large_mergen_lds(4, 4), s, p, g);
      else if (p.N <= 512) hipLaunchKernelGGL((large_mergen_kernel<8, 2>), grid, block, large_mergen_lds(8, 2), s, p, g);
      else hipLaunchKernelGGL((large_mergen_kernel<16, 2>), grid, block, large_mergen_lds(16, 2), s, p, g);
    } else {
      const int w = large_merge_waves(p.N);
      const unsigned nblk = (unsigned)((p.N + w - 1) / w);
      hipLaunchKernelGGL(large_merge_kernel, dim3((unsigned)p.B * nblk), dim3(64 * w), large_merge_lds(p.N), s, p, g);
    }
  }
  if (want_hist) {
    const int vw = large_hist_viewers(p.K);
    const unsigned nblk = (unsigned)((p.N + vw - 1) / vw);
    hipLaunchKernelGGL(large_hist_kernel, dim3((unsigned)p.B * nblk), dim3(64 * kLargeHistWaves), large_hist_lds(p.K), s, p);
  }
  return hipGetLastError();
}

hipError_t set_attr_large(int N, int A, int K) {
  hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(large_search_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)large_lds_layout(N, A).total);
  if (r != hipSuccess) return r;
  r = hipFuncSetAttribute(reinterpret_cast<const void*>(large_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)large_merge_lds(N));
  if (r != hipSuccess) return r;
  if (N <= 1024 && N > 512)
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(large_mergen_kernel<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)large_mergen_lds(16, 2));
  else if (N <= 512 && N > 256)
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(large_mergen_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)large_mergen_lds(8, 2));
  if (r != hipSuccess) return r;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(large_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)large_hist_lds(K));
}

// what diral_env_validate needs to know without seeing the kernels
uint32_t large_lds_bytes(int N, int A, int K) {
  const uint32_t a = large_lds_layout(N, A).total, b = large_merge_lds(N), c = large_hist_lds(K);
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

}  // namespace diral
