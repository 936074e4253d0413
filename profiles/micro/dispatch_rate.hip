// dispatch_rate.hip - how fast does an MI355X start workgroups of the step kernel's shape?
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
// Launches G workgroups of 256 threads with L bytes of LDS; each workgroup busy-waits `spin` shader
// clocks (wall_clock64 at 100 MHz is too coarse: s_memtime).  Kernel time vs the ideal
// ceil(G / (256 CUs x resident)) x spin shows what the dispatcher adds at the bench's grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void spin_kernel(long long spin, int* out) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < spin) { __builtin_amdgcn_s_sleep(2); }
  if (out && lds[(threadIdx.x + 1) & 255] == -1) out[blockIdx.x] = 1;
}

int main() {
  int* out; hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int lds_sizes[] = {1024, 19712, 19712};
  for (int li = 0; li < 3; ++li)
    for (long long spin : {0LL, 2000LL, 10000LL, 48000LL})
      for (int G : {256, 1792, 4096, 8192, 32768}) {
        const int L = lds_sizes[li];
        if (li == 2) hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        const int lds = li == 2 ? 22528 : L;     // 7 per CU by LDS (160 KB / 22 KB)
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), lds, 0, spin, out);
        hipEventRecord(a);
        const int R = 20;
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), lds, 0, spin, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("lds %5d spin %6lld ticks  G %6d : %8.2f us per launch  (%.1f ns per WG)\n", lds, spin, G, ms / R * 1e3, ms / R * 1e6 / G);
      }
  return 0;
}
