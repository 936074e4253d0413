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
    if (p.N <= 1024) {
      // four columns per wave on thermometer codes (tables that stay fresh); the quads it flags - an entry 8 or more stamps
      // behind - on (rank, source) keys, two columns per wave, behind it
      const unsigned nquad = (unsigned)((p.N + 3) >> 2), cblk = (nquad + 3) >> 2;
      const unsigned npair = (unsigned)((p.N + 1) >> 1), rblk = (npair + 3) >> 2;
      const dim3 cgrid((unsigned)p.B * cblk), rgrid((unsigned)p.B * rblk), block(256);
      if (p.N <= 256) {
        hipLaunchKernelGGL((large_mergec_kernel<4>), cgrid, block, large_mergec_lds(4), s, p, g);
        hipLaunchKernelGGL((large_mergen_kernel<4, 2, true>), rgrid, block, large_mergen_lds(4, 2), s, p, g);
      } else if (p.N <= 512) {
        hipLaunchKernelGGL((large_mergec_kernel<8>), cgrid, block, large_mergec_lds(8), s, p, g);
        hipLaunchKernelGGL((large_mergen_kernel<8, 2, true>), rgrid, block, large_mergen_lds(8, 2), s, p, g);
      } else {
        // (512 < N <= 1024: the rank keys alone.  The codes form at 16 chunks per lane runs at 128 VGPRs with 23 spilled and
        // measured no gain where it applies - 1024 / 512: 23.7 -> 27.2 ms, 1024 / 64: 6.5 -> 6.8 - and a highway that long
        // keeps entries about far vehicles for more than 7 stamps anyway)
        hipLaunchKernelGGL((large_mergen_kernel<16, 2, false>), rgrid, block, large_mergen_lds(16, 2), s, p, g);
      }
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
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(large_mergen_kernel<16, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)large_mergen_lds(16, 2));
  else if (N <= 512 && N > 256)
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(large_mergen_kernel<8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
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
