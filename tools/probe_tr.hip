// Probe of ds_read_b64_tr_b16 lane semantics on gfx950: fills LDS with element index, every lane
// supplies its own address, prints what each lane receives.  hipcc --offload-arch=gfx950 probe_tr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* addr_in, uint16_t* out) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned a = (unsigned)addr_in[threadIdx.x] * 2u + (unsigned)(uintptr_t)lds * 0;  // byte address inside lds[]
  uint2 v;
  unsigned base = (unsigned)(size_t)(&lds[0]);
  unsigned addr = base + a;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff;
  out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff;
  out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
  int h_addr[64];
  uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pattern = 0; pattern < 3; ++pattern) {
    for (int l = 0; l < 64; ++l) {
      if (pattern == 0) h_addr[l] = l * 4;                         // contiguous b64 per lane
      if (pattern == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 4;  // lane = row of 64 elements, group = col block
      if (pattern == 2) h_addr[l] = ((l & 15) >> 2) * 256 + (l & 3) * 4 + (l >> 4) * 16;  // 4 rows x 16 cols blocks, row stride 256
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pattern);
    for (int l = 0; l < 64; ++l)
      printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
