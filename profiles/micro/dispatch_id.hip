#include <hip/hip_runtime.h>
extern "C" __device__ unsigned long long diral_dispatch_id() __asm("llvm.amdgcn.dispatch.id");
__global__ void k(unsigned long long* o) {
  o[blockIdx.x] = diral_dispatch_id();
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 64);
  for (int i = 0; i < 5; ++i) { hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d + i); }
  hipStream_t s; (void)hipStreamCreate(&s);
  for (int i = 5; i < 8; ++i) { hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s, d + i); }
  unsigned long long h[8]; (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("%llu ", h[i]); printf("\n");
  return 0;
}
