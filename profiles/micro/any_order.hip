// any_order.hip - do two kernels of one stream overlap when the second is launched with hipExtAnyOrderLaunch?
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/any_order.hip -o /tmp/any_order && /tmp/any_order
// G workgroups of 256 threads (22 KB LDS: 7 per CU) spin for `spin` ticks; R launches back to back, in order vs any-order.
// G = 4096 is 2.29 rounds of the machine: overlapping launches would fill the last round (R launches -> ~2.29 R rounds
// instead of 3 R).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ __launch_bounds__(256) void spin_kernel(long long spin, int* out) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < spin) { __builtin_amdgcn_s_sleep(2); }
  if (out && lds[(threadIdx.x + 1) & 255] == -1) out[blockIdx.x] = 1;
}

int main() {
  int* out; hipMalloc(&out, 1 << 20);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  long long spin = 48000; int G = 4096;
  void* args[] = {&spin, &out};
  for (int G_ : {1792, 4096}) {
    G = G_;
    for (int flags : {0, 1}) {
      for (int w = 0; w < 3; ++w) hipExtLaunchKernel((const void*)spin_kernel, dim3(G), dim3(256), args, 22528, s, nullptr, nullptr, 0);
      hipStreamSynchronize(s);
      hipEventRecord(a, s);
      const int R = 40;
      for (int r = 0; r < R; ++r) {
        hipError_t e = hipExtLaunchKernel((const void*)spin_kernel, dim3(G), dim3(256), args, 22528, s, nullptr, nullptr, flags);
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
      }
      hipEventRecord(b, s); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("G %5d flags %d : %8.2f us per launch\n", G, flags, ms / R * 1e3);
    }
  }
  return 0;
}
