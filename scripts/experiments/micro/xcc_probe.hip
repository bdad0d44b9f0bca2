// Which XCD runs workgroup i?  (The observable kernels map blockIdx -> (state, block) assuming linear id mod 8.)
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/xcc_probe.hip -o scripts/micro/xcc_probe.bin && scripts/micro/xcc_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;   // keep the workgroup resident for a while
  if (v == 12345.f) lds[threadIdx.x] = v;
}
int main() {
  const int n = 16384;
  unsigned* d;
  hipMalloc(&d, n * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
  for (int spin : {0, 20000}) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(512), 65536 + 64, 0, d, spin);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < n; ++i) ok += h[i] == unsigned(i % 8);
    printf("spin %d: %d of %d workgroups on XCD (id mod 8); first 32:", spin, ok, n);
    for (int i = 0; i < 32; ++i) printf(" %u", h[i]);
    printf("\n");
    int hist[16] = {0};
    for (int i = 0; i < n; ++i) hist[h[i] & 15]++;
    printf("  histogram:");
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf("\n");
  }
  return 0;
}
