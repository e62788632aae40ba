// Instruction-level A/B of the two dense bf16 MFMA shapes of gfx950 (VERDICT r4 next #2a): v_mfma_f32_16x16x32_bf16 against
// v_mfma_f32_32x32x16_bf16, (1) from registers alone and (2) fed from LDS at the forward kernel's wave tile (128 pixels x 64 channels
// per wave: 12 16-byte fragment reads per 32 K-columns for either shape), at one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_rate_probe tools/mfma_rate_probe.hip && tools/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) short ab_t;
typedef __attribute__((ext_vector_type(4))) float c4_t;
typedef __attribute__((ext_vector_type(16))) float c16_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (1) registers only: 64 accumulator registers either way
__global__ __launch_bounds__(512) void reg16_kernel(float* out, int iters) {
  ab_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  c4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = c4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}

__global__ __launch_bounds__(512) void reg32_kernel(float* out, int iters) {
  ab_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  c16_t acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.f) out[0] = s;
}

// (2) LDS-fed, wave tile 128 x 64, K step 32: every wave reads its own fragments (conflict-free 16-byte lanes) from a rotating window
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lds16_kernel(float* out, int iters) {
  extern __shared__ char sm[];
  for (int i = threadIdx.x; i < 16384; i += WAVES * 64) ((int*)sm)[i] = i * 2654435761u >> 20;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  c4_t acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = c4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const char* base = sm + ((it & 3) << 14) + lane * 16;     // 4 windows of 16 KB, 12 KB read from each
    ab_t fa[8], fb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *(const ab_t*)(base + i * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *(const ab_t*)(base + 8192 + j * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.f) out[0] = s;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lds32_kernel(float* out, int iters) {
  extern __shared__ char sm[];
  for (int i = threadIdx.x; i < 16384; i += WAVES * 64) ((int*)sm)[i] = i * 2654435761u >> 20;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  c16_t acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const char* base = sm + ((it & 3) << 14) + lane * 16;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {                            // two K halves of 16 columns: 6 fragment reads each
      ab_t fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const ab_t*)(base + (kh * 6 + i) * 1024);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *(const ab_t*)(base + (kh * 6 + 4 + j) * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
  if (s == 12345.f) out[0] = s;
}

template <class F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* out; CHECK(hipMalloc(&out, 64));
  hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const double clk = pr.clockRate * 1e3;
  printf("device %s, %d CUs, %.0f MHz\n", pr.name, cus, pr.clockRate / 1e3);
  const int iters = 20000;
  printf("%-44s %8s %10s %16s\n", "variant", "ms", "TFLOP/s", "cyc/SIMD/16KFLOP");
  auto report = [&](const char* name, double ms, int waves, double flop_per_wave_iter) {
    const double fl = (double)cus * waves * iters * flop_per_wave_iter;
    const double per_simd = waves / 4.0 * iters * flop_per_wave_iter;     // FLOP one SIMD executed
    printf("%-44s %8.3f %10.1f %16.2f\n", name, ms, fl / ms / 1e9, ms * 1e-3 * clk / (per_simd / 16384.0));
  };
  for (int waves : {4, 8}) {
    char nm[96];
    double ms = time_ms([&] { hipLaunchKernelGGL(reg16_kernel, dim3(cus), dim3(waves * 64), 0, 0, out, iters); });
    snprintf(nm, 96, "registers 16x16x32, %d wave(s)/SIMD", waves / 4); report(nm, ms, waves, 16 * 16384.0);
    ms = time_ms([&] { hipLaunchKernelGGL(reg32_kernel, dim3(cus), dim3(waves * 64), 0, 0, out, iters); });
    snprintf(nm, 96, "registers 32x32x16, %d wave(s)/SIMD", waves / 4); report(nm, ms, waves, 8 * 32768.0);
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL((lds16_kernel<4>), dim3(cus), dim3(256), 65536, 0, out, iters); });
    report("LDS-fed 128x64 tile 16x16x32, 1 wave/SIMD", ms, 4, 32 * 16384.0);
    ms = time_ms([&] { hipLaunchKernelGGL((lds32_kernel<4>), dim3(cus), dim3(256), 65536, 0, out, iters); });
    report("LDS-fed 128x64 tile 32x32x16, 1 wave/SIMD", ms, 4, 16 * 32768.0);
    ms = time_ms([&] { hipLaunchKernelGGL((lds16_kernel<8>), dim3(cus), dim3(512), 65536, 0, out, iters); });
    report("LDS-fed 128x64 tile 16x16x32, 2 waves/SIMD", ms, 8, 32 * 16384.0);
    ms = time_ms([&] { hipLaunchKernelGGL((lds32_kernel<8>), dim3(cus), dim3(512), 65536, 0, out, iters); });
    report("LDS-fed 128x64 tile 32x32x16, 2 waves/SIMD", ms, 8, 16 * 32768.0);
  }
  return 0;
}
