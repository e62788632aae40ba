// What does ISSUING a global_load_lds_dwordx4 cost the wave that issues it?  (round 6; dev tool, GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/glds_issue_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
// The halo weight-gradient kernels issue 10 (8 waves) or 19 (4 waves) LDS-DMA instructions per wave and tile in one burst and lose
// ~165 - 185 cycles of the wave PER INSTRUCTION (profiles/r06_wgrad_sw_ab.txt: 25 us / 53 us per launch of 32 tiles).  Variants, N
// instructions per burst, one wave per SIMD or two, every CU busy; s_memtime around the issue sequence only (no wait for the data)
// and around issue + s_waitcnt vmcnt(0):
//   0  m0 saved / set / restored around every instruction (the kernels' glds16)
//   1  m0 set before every instruction, not restored
//   2  m0 set ONCE (all N instructions write the same LDS range: timing only)
//   5  the MFMAs of 4 alone;  6 / 7 / 8: with the kernels' fragment reads in the loop -- see the code
//   4  as 1, with 8 independent v_mfma_f32_32x32x16 between two instructions (does the DMA issue hide under MFMA work?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V, int N>
__global__ __launch_bounds__(512) void probe(const uint4* src, long stride16, unsigned long long* t_issue, unsigned long long* t_done, float* sink) {
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];
  typedef __attribute__((address_space(3))) char* lds_cptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint4* g = src + ((long)blockIdx.x * 64 + wave) * stride16 + lane;
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  u32x4 fa = {threadIdx.x, 1u, 2u, 3u}, fb = {5u, 6u, 7u, threadIdx.x};
  uint4 regs[N];
  unsigned long long a0 = 0, a1 = 0, a2 = 0;
  for (int rep = 0; rep < 4; ++rep) {       // the last repetition is the one reported (code and data warm)
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a0)::"memory");
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint4* gp = g + (long)(rep * N + i) * 64;
      const unsigned dst = lds0 + ((wave * N + (V == 2 ? 0 : i)) % 144) * 1024;
      if (V == 0) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
      } else if (V == 1 || V == 4) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(dst) : "memory");
        if (V == 4) {
#pragma unroll
          for (int k = 0; k < 8; ++k) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(fa), "v"(fb) : "memory");
        }
      } else if (V == 2) {
        if (i == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory");
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp) : "memory");
      } else if (V == 6 || V == 7 || V == 8) {
        // 6: as 4, plus 8 ds_read_b64 of an LDS range no DMA writes, waited for before the MFMAs (the kernels' fragment reads)
        // 7: the same reads + MFMAs WITHOUT any DMA;  8: the DMA is issued by the ODD waves only, reads + MFMAs by the EVEN waves only
        const bool producer = V == 8 && (wave & 1), consumer = V != 8 || !(wave & 1);
        if (V == 6 || producer) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(dst) : "memory");
        if (producer) __builtin_amdgcn_s_sleep(4);
        if (consumer) {
          uint2 r[8];
          const unsigned ra = lds0 + 147456 + lane * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[k]) : "v"(ra), "n"(0) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < 8; ++k) fa[k & 3] ^= r[k].x;
#pragma unroll
          for (int k = 0; k < 8; ++k) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(fa), "v"(fb) : "memory");
        }
      } else {      // 5: the MFMAs of variant 4 alone
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(fa), "v"(fb) : "memory");
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a2)::"memory");
  }
  if (lane == 0) {
    t_issue[blockIdx.x * (blockDim.x >> 6) + wave] = a1 - a0;
    t_done[blockIdx.x * (blockDim.x >> 6) + wave] = a2 - a0;
  }
  float s = 0.f;
  if (V == 3)
    for (int i = 0; i < N; ++i) s += (float)regs[i].x;
  for (int k = 0; k < 8; ++k) s += acc[k][0];
  if (s == 123.456f) sink[0] = s + (float)sm[threadIdx.x].x;
}

template <int V, int N>
void run(const char* tag, int waves, const uint4* src, unsigned long long* ti, unsigned long long* td, float* sink) {
  const int nwg = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<V, N>), hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipLaunchKernelGGL((probe<V, N>), dim3(nwg), dim3(waves * 64), 150000, 0, src, (long)(4 * N * 64), ti, td, sink);   // 150000 B of LDS: one workgroup per CU
  { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("%s: %s\n", tag, hipGetErrorString(e)); return; } }
  std::vector<unsigned long long> a(nwg * waves), b(nwg * waves);
  hipMemcpy(a.data(), ti, a.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), td, b.size() * 8, hipMemcpyDeviceToHost);
  std::sort(a.begin(), a.end());
  std::sort(b.begin(), b.end());
  // s_memtime ticks: raw (the variant-5 line calibrates them: N x 8 MFMAs x 32 shader cycles at one wave per SIMD)
  if (V == 8) {      // consumers only (even waves)
    std::vector<unsigned long long> c;
    hipMemcpy(a.data(), ti, a.size() * 8, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < a.size(); i += 2) c.push_back(a[i]);
    std::sort(c.begin(), c.end());
    a = c;
  }
  printf("%-58s N=%2d waves/CU=%d  issue: median %6llu ticks (%6.1f / instr)   issue+land: median %6llu ticks\n", tag, N, waves, a[a.size() / 2],
         a[a.size() / 2] * 1.0 / N, b[b.size() / 2]);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint4* src;
  unsigned long long *ti, *td;
  float* sink;
  const size_t bytes = (size_t)256 * 64 * 4 * 19 * 64 * 16 + (1 << 20);
  if (hipMalloc(&src, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, bytes);
  hipMalloc(&ti, 8 * 4096);
  hipMalloc(&td, 8 * 4096);
  hipMalloc(&sink, 64);
  for (int waves : {4, 8}) {
    run<0, 10>("0 m0 save / set / restore per instruction (glds16)", waves, src, ti, td, sink);
    run<1, 10>("1 m0 set per instruction", waves, src, ti, td, sink);
    run<2, 10>("2 m0 set once", waves, src, ti, td, sink);
    run<4, 10>("4 m0 set per instruction + 8 MFMA 32x32x16 between", waves, src, ti, td, sink);
    run<5, 10>("5 the 8 MFMAs of variant 4 alone (10 x 8 x 32 cycles)", waves, src, ti, td, sink);
    run<7, 10>("7 8 ds_read_b64 + wait + 8 MFMA per round, NO DMA", waves, src, ti, td, sink);
    run<6, 10>("6 DMA + 8 ds_read_b64 (other range) + wait + 8 MFMA", waves, src, ti, td, sink);
    run<8, 10>("8 DMA from the odd waves, reads + MFMAs on the even waves", waves, src, ti, td, sink);
    run<4, 19>("4 m0 set per instruction + 8 MFMA 32x32x16 between", waves, src, ti, td, sink);
    run<5, 19>("5 the 8 MFMAs of variant 4 alone (19 x 8 x 32 cycles)", waves, src, ti, td, sink);
    run<0, 19>("0 m0 save / set / restore per instruction (glds16)", waves, src, ti, td, sink);
    run<1, 19>("1 m0 set per instruction", waves, src, ti, td, sink);
    run<2, 19>("2 m0 set once", waves, src, ti, td, sink);
  }
  return 0;
}
