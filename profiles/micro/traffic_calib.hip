// traffic_calib.hip - what do FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count, per access width?
//
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/traffic_calib.hip -o /tmp/traffic_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -o fetch -- /tmp/traffic_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir>/write -o write -- /tmp/traffic_calib
//   python profiles/traffic_calib.py <dir>          -> profiles/traffic_calibration.json
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE for 16 B per lane streaming reads only ("reports exactly 1/2 of the bytes")
// and leaves other widths and WRITE_SIZE "uncalibrated: calibrate on a known byte count in your own access pattern".
// The step kernels read and write the table one dword per lane (coalesced: 256 B per wave instruction), the ring and
// the per-vehicle arrays 8 B per lane, and write their outputs 16 B per lane with non-temporal stores.  Each kernel below
// streams NBYTES (256 MiB: past the 32 MiB of L2; Infinity-Cache hits are counted by the memory-side counters) once, with
// exactly one of those access forms; the kernel name carries the form and the byte count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr size_t NBYTES = 256ull << 20;

template <typename T>
__global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ src, T* sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned int acc = 0;
  for (size_t j = i; j < n; j += stride) {
    const T v = src[j];
    const unsigned int* w = reinterpret_cast<const unsigned int*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
  }
  if (acc == 0x12345u) reinterpret_cast<unsigned int*>(sink)[i] = acc;     // (never: keeps the loads alive)
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void write_kernel(T* __restrict__ dst, size_t n, unsigned int seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  T v;
  unsigned int* w = reinterpret_cast<unsigned int*>(&v);
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = seed + (unsigned int)i + k;
  for (size_t j = i; j < n; j += stride) {
    if constexpr (NT) __builtin_nontemporal_store(v, dst + j);
    else dst[j] = v;
  }
}

// the table pattern of step_fast64: a workgroup reads AND rewrites 16 rows of 64 dwords (256 B rows) of its own 4 KB block
__global__ __launch_bounds__(256) void rw_dword_rows_kernel(unsigned int* tab, size_t nblocks4k) {
  for (size_t b = blockIdx.x; b < nblocks4k; b += gridDim.x) {
    unsigned int* p = tab + b * 1024 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
    unsigned int v[4];
    for (int q = 0; q < 4; ++q) v[q] = p[q * 64];
    for (int q = 0; q < 4; ++q) p[q * 64] = (v[q] << 1) | 1u;
  }
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

int main() {
  void *a, *b;
  if (hipMalloc(&a, NBYTES) != hipSuccess || hipMalloc(&b, NBYTES) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, NBYTES); hipMemset(b, 2, NBYTES);
  const int G = 256 * 16;      // 16 workgroups per CU
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((read_kernel<unsigned int>), dim3(G), dim3(256), 0, 0, (const unsigned int*)a, (unsigned int*)b, NBYTES / 4);
    hipLaunchKernelGGL((read_kernel<u32x2>), dim3(G), dim3(256), 0, 0, (const u32x2*)a, (u32x2*)b, NBYTES / 8);
    hipLaunchKernelGGL((read_kernel<u32x4>), dim3(G), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, NBYTES / 16);
    hipLaunchKernelGGL((write_kernel<unsigned int, false>), dim3(G), dim3(256), 0, 0, (unsigned int*)b, NBYTES / 4, 7u + rep);
    hipLaunchKernelGGL((write_kernel<u32x2, false>), dim3(G), dim3(256), 0, 0, (u32x2*)b, NBYTES / 8, 7u + rep);
    hipLaunchKernelGGL((write_kernel<u32x4, false>), dim3(G), dim3(256), 0, 0, (u32x4*)b, NBYTES / 16, 7u + rep);
    hipLaunchKernelGGL((write_kernel<unsigned int, true>), dim3(G), dim3(256), 0, 0, (unsigned int*)b, NBYTES / 4, 7u + rep);
    hipLaunchKernelGGL((write_kernel<u32x4, true>), dim3(G), dim3(256), 0, 0, (u32x4*)b, NBYTES / 16, 7u + rep);
    hipLaunchKernelGGL(rw_dword_rows_kernel, dim3(G), dim3(256), 0, 0, (unsigned int*)a, NBYTES / 4096);
    hipDeviceSynchronize();
  }
  printf("traffic_calib: %zu bytes per kernel, 3 repetitions\n", NBYTES);
  return 0;
}
