// Probe: do literal LDS addresses + dynamic LDS (no __shared__ declaration) work on this stack?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define AS3 __attribute__((address_space(3)))
__global__ void probe(unsigned* out, int use_decl) {
  int t = threadIdx.x;
  *(AS3 unsigned*)(uintptr_t)(16 + t * 4) = 1000 + t;
  __syncthreads();
  out[t] = *(AS3 unsigned*)(uintptr_t)(16 + ((t + 1) & 63) * 4);
  unsigned long long v = 0x1234567800000000ull + t;
  *(AS3 unsigned long long*)(uintptr_t)(1024 + t * 8) = v;
  __syncthreads();
  out[64 + t] = (unsigned)(*(AS3 unsigned long long*)(uintptr_t)(1024 + t * 8) >> 32);
}
__global__ void probe_decl(unsigned* out) {
  extern __shared__ unsigned dyn[];
  int t = threadIdx.x;
  dyn[4 + t] = 2000 + t;
  __syncthreads();
  out[t] = *(AS3 unsigned*)(uintptr_t)(16 + ((t + 1) & 63) * 4);
  out[64 + t] = (unsigned)(uintptr_t)(AS3 unsigned*)dyn;
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipMemset(d, 0xff, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, 0);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("literal+dynamic: err=%d out[0]=%u out[63]=%u hi=%x\n", (int)e, h[0], h[63], h[64]);
  hipMemset(d, 0xff, sizeof(h));
  hipLaunchKernelGGL(probe_decl, dim3(1), dim3(64), 4096, 0, d);
  e = hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("extern decl: err=%d out[0]=%u out[63]=%u dynbase=%u\n", (int)e, h[0], h[63], h[64]);
  return 0;
}
