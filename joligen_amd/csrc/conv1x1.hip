// Streaming 1x1 convolution for the wide, shallow layers of the UNet (skip / shortcut convs at full resolution: 64 -> 128,
// 64 -> 192, 128 -> 64 ... channels on 32 x 256^2 pixels).  These are HBM-bound (a few hundred FLOP per byte at most) and the
// generic LDS-DMA implicit-GEMM kernel reaches only ~2 TB/s on them: one K step per tile leaves its pipeline nothing to overlap.
//
// Here nothing goes through LDS: MFMA computes D^T = W . X^T per 16-pixel tile,
//   A operand = weights (row = output channel), loaded ONCE per wave into registers for the whole kernel;
//   B operand = pixels  (col = pixel): lane (l & 15, l >> 4) loads 16 contiguous bytes of pixel row (l & 15) -- a wave instruction
//               covers 16 full 64-byte segments, no im2col, no barrier;
//   D         = [16 output channels] x [16 pixels]: a lane owns 4 consecutive channels of ONE pixel; the weight rows of two MFMA tiles
//               are interleaved so that a lane's 2 x 4 channels are 8 consecutive ones -> one 16-byte store (and one 16-byte read of
//               the residual) per lane and tile pair.
// A wave walks pixel tiles with a stride of the total wave count and keeps the fragments of the next tile in flight.
//   y = alpha * conv + bias + res_scale * res     (same epilogue contract as the other convolution kernels)
#include <stdlib.h>

#include "conv_params.h"

namespace {

__device__ __forceinline__ float act_rt(float u, int act) {
  return act == JG_ACT_SILU ? silu_f(u) : act == JG_ACT_RELU ? fmaxf(u, 0.f) : act == JG_ACT_LRELU ? (u > 0.f ? u : 0.2f * u) : u;
}

// APPLY: the launch also writes act(a x + b) of its INPUT (the GroupNorm apply pass of x, ConvP.aab / ay): the block column
// blockIdx.y == 0 transforms every 16-byte fragment it has loaded anyway and stores it; the (a, b) rows of the wave's current image
// are staged in a wave-private LDS slot (2 Cin floats) and re-staged when the wave's tile sequence crosses an image boundary.
// GNB: the epilogue adds the GroupNorm-backward apply step of the output tensor (ConvP.bgx ...): per lane and tile pair two more 16-byte
// loads (gx, gdy; + the optional addends) and the per-channel coefficients (a, b, P, Q, R) of the block's PAIRS * 32 output channels from
// a wave-private LDS slot laid out [5][PAIRS * 32] (re-staged at image boundaries).
template <typename T, int PAIRS, int KS, bool APPLY = false, bool GNB = false>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(ConvP p, int ntiles) {
  __shared__ __attribute__((aligned(16))) float s_ab[APPLY ? 4 * 2 * KS * 32 : (GNB ? 4 * 5 * PAIRS * 32 : 4)];
  const int lane = threadIdx.x & 63;
  const int a = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * (PAIRS * 32);
  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ w = (const T*)p.w;
  const T* __restrict__ res = (const T*)p.res;
  T* __restrict__ y = (T*)p.y;

  // weights: tile (pair, half) row a  <->  output channel n0 + pair*32 + (a >> 2)*8 + half*4 + (a & 3)
  uint4 wf[PAIRS][2][KS];
#pragma unroll
  for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int n = n0 + pr * 32 + (a >> 2) * 8 + hf * 4 + (a & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[pr][hf][ks] = *reinterpret_cast<const uint4*>(w + (long)n * p.ldw + ks * 32 + g * 8);
    }
  // bias of this lane's 8 channels per pair: n0 + pair*32 + g*8 + 0..7
  float bs[PAIRS][8];
#pragma unroll
  for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[pr][j] = (!GNB && p.bias) ? p.bias[n0 + pr * 32 + g * 8 + j] : 0.f;      // GNB: an input gradient has no bias

  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  auto load_tile = [&](int t, uint4* xf) {
    const T* px = x + ((long)t * 16 + a) * p.ldx + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const uint4*>(px + ks * 32);
  };
  const bool do_apply = APPLY && blockIdx.y == 0;
  float* wab = s_ab + (threadIdx.x >> 6) * (APPLY ? 2 * KS * 32 : (GNB ? 5 * PAIRS * 32 : 1));       // this wave's coefficient rows
  const int tiles_per_img = (p.Ho * p.Wo) >> 4;
  int cur_b = -1;
  // PF tiles of input fragments in flight per wave (a memory-bound stream needs ~64 KB in flight per CU)
  constexpr int PF = KS <= 2 ? 4 : (KS <= 4 ? 2 : 1);
  uint4 ring[PF][KS];
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (wave + s * nwaves < ntiles) load_tile(wave + s * nwaves, ring[s]);
  for (int base = wave; base < ntiles; base += PF * nwaves) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int t = base + s * nwaves;
      if (t >= ntiles) break;
      const long prow = ((long)t * 16 + a);
      if (GNB) {
        const int b = t / tiles_per_img;
        if (b != cur_b) {               // wave-uniform: (a, b, P, Q, R) of this block's channels for the new image -> [5][PAIRS * 32]
          cur_b = b;
          constexpr int NC = PAIRS * 32;
          for (int c = lane; c < NC; c += 64) {
            const long row = (long)b * p.N + n0 + c;
            wab[c] = p.bab[row * 2];
            wab[NC + c] = p.bab[row * 2 + 1];
            wab[2 * NC + c] = p.bpqr[row * 3];
            wab[3 * NC + c] = p.bpqr[row * 3 + 1];
            wab[4 * NC + c] = p.bpqr[row * 3 + 2];
          }
        }
      }
      uint4 rf[PAIRS];
      if (res) {
#pragma unroll
        for (int pr = 0; pr < PAIRS; ++pr) rf[pr] = *reinterpret_cast<const uint4*>(res + prow * p.ldres + n0 + pr * 32 + g * 8);
      }
      f32x4 acc[PAIRS][2];
#pragma unroll
      for (int pr = 0; pr < PAIRS; ++pr) {
        acc[pr][0] = acc[pr][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          acc[pr][0] = Mfma<T>::run(wf[pr][0][ks], ring[s][ks], acc[pr][0]);
          acc[pr][1] = Mfma<T>::run(wf[pr][1][ks], ring[s][ks], acc[pr][1]);
        }
      }
      if (do_apply) {
        const int b = t / tiles_per_img;
        if (b != cur_b) {               // wave-uniform: stage the image's coefficient rows (same-wave LDS writes are read back in order)
          cur_b = b;
          const float* src = p.aab + (long)b * (2 * KS * 32);
          for (int i = lane; i < 2 * KS * 32 / 4; i += 64) reinterpret_cast<float4*>(wab)[i] = reinterpret_cast<const float4*>(src)[i];
        }
        T* ay = (T*)p.ay + prow * p.lday + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          float f[8];
          unpack8<T>(ring[s][ks], f);
          const float4* c4 = reinterpret_cast<const float4*>(wab + (ks * 32 + g * 8) * 2);
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const float4 c = c4[k4];      // (a, b) of channels 2 k4, 2 k4 + 1 of the octet
            const float u0 = c.x * f[2 * k4] + c.y, u1 = c.z * f[2 * k4 + 1] + c.w;
            f[2 * k4] = p.aact == JG_ACT_SILU ? silu_f(u0) : act_rt(u0, p.aact);
            f[2 * k4 + 1] = p.aact == JG_ACT_SILU ? silu_f(u1) : act_rt(u1, p.aact);
          }
          *reinterpret_cast<uint4*>(ay + ks * 32) = pack8<T>(f);
        }
      }
      const int tn = t + PF * nwaves;       // refill this slot as soon as its fragments have been consumed
      if (tn < ntiles) load_tile(tn, ring[s]);
#pragma unroll
      for (int pr = 0; pr < PAIRS; ++pr) {
        float o[8] = {acc[pr][0][0], acc[pr][0][1], acc[pr][0][2], acc[pr][0][3], acc[pr][1][0], acc[pr][1][1], acc[pr][1][2], acc[pr][1][3]};
        float r8[8];
        if (res) unpack8<T>(rf[pr], r8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = GNB ? p.alpha * o[j] : p.alpha * o[j] + bs[pr][j] + (res ? p.res_scale * r8[j] : 0.f);
        if (GNB) {
          constexpr int NC = PAIRS * 32;
          const int cl = pr * 32 + g * 8, cg = n0 + cl;
          float fx[8], fd[8], ca[8], cb[8], cP[8], cQ[8], cR[8];
          unpack8<T>(*reinterpret_cast<const uint4*>((const T*)p.bgx + prow * p.bldgx + cg), fx);
          unpack8<T>(*reinterpret_cast<const uint4*>((const T*)p.bgdy + prow * p.bldgdy + cg), fd);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(ca + 4 * h) = *reinterpret_cast<const float4*>(wab + cl + 4 * h);
            *reinterpret_cast<float4*>(cb + 4 * h) = *reinterpret_cast<const float4*>(wab + NC + cl + 4 * h);
            *reinterpret_cast<float4*>(cP + 4 * h) = *reinterpret_cast<const float4*>(wab + 2 * NC + cl + 4 * h);
            *reinterpret_cast<float4*>(cQ + 4 * h) = *reinterpret_cast<const float4*>(wab + 3 * NC + cl + 4 * h);
            *reinterpret_cast<float4*>(cR + 4 * h) = *reinterpret_cast<const float4*>(wab + 4 * NC + cl + 4 * h);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float du = fd[j];
            if (p.bact != JG_ACT_NONE) du *= act_grad_rt(ca[j] * fx[j] + cb[j], p.bact);
            o[j] += du * cP[j] + fx[j] * cQ[j] + cR[j];
          }
          if (p.badd1) {
            float fa[8];
            unpack8<T>(*reinterpret_cast<const uint4*>((const T*)p.badd1 + prow * p.bldadd1 + cg), fa);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += p.bsc1 * fa[j];
          }
          if (p.badd2) {
            float fa[8];
            unpack8<T>(*reinterpret_cast<const uint4*>((const T*)p.badd2 + prow * p.bldadd2 + cg), fa);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += p.bsc2 * fa[j];
          }
        }
        *reinterpret_cast<uint4*>(y + prow * p.ldy + n0 + pr * 32 + g * 8) = pack8<T>(o);
        if (GNB) __builtin_amdgcn_sched_barrier(0);      // one channel group at a time: its operand vectors and coefficient octets die here
      }
    }
  }
}

// ---- the same streaming scheme for the 3x3 stem (and the input gradient of the 3x3 head): Cin = 8 (one 16-byte chunk per tap) ----
// K = 9 taps x 8 channels = 72 -> 3 MFMA k-steps of 4 taps (the last three tap slots are zero); the B operand of lane (pixel a, group g)
// for k-step ks is the 16-byte pixel at tap ks*4 + g, read straight from global memory (zero outside the image).  The generic
// im2col kernel spends 375 us on 32 x 256^2 x (8 -> 64); the layer moves 33 MB in and 268 MB out.
template <typename T, int PAIRS>
__global__ __launch_bounds__(256, 2) void conv3x3_c8_stream_kernel(ConvP p, int ntiles) {
  constexpr int KS = 3;
  const int lane = threadIdx.x & 63;
  const int a = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * (PAIRS * 32);
  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ w = (const T*)p.w;
  const T* __restrict__ res = (const T*)p.res;
  T* __restrict__ y = (T*)p.y;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  uint4 wf[PAIRS][2][KS];
#pragma unroll
  for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int n = n0 + pr * 32 + (a >> 2) * 8 + hf * 4 + (a & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int tap = ks * 4 + g;
        wf[pr][hf][ks] = tap < 9 ? *reinterpret_cast<const uint4*>(w + (long)n * p.ldw + tap * 8) : zero4;
      }
    }
  float bs[PAIRS][8];
#pragma unroll
  for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[pr][j] = p.bias ? p.bias[n0 + pr * 32 + g * 8 + j] : 0.f;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  auto load_tile = [&](int t, uint4* xf) {
    const long pix = (long)t * 16 + a;
    const int px = pix % p.W, py = (pix / p.W) % p.H;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int tap = ks * 4 + g;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool ok = tap < 9 && (unsigned)(py + dy) < (unsigned)p.H && (unsigned)(px + dx) < (unsigned)p.W;
      xf[ks] = ok ? *reinterpret_cast<const uint4*>(x + (pix + (long)dy * p.W + dx) * 8) : zero4;
    }
  };
  // a wave owns a CONTIGUOUS run of tiles, so the fused GroupNorm statistics (sum, sum of squares per image and channel, from the
  // fp32 values) stay in registers until the image changes: one shuffle reduction + atomic pair per channel and image per wave
  const int per = (ntiles + nwaves - 1) / nwaves;
  const int t0 = wave * per, t1 = min(ntiles, t0 + per);
  const int tiles_per_img = (p.H * p.W) >> 4;
  float s1[PAIRS][8], s2[PAIRS][8];
#pragma unroll
  for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[pr][j] = s2[pr][j] = 0.f;
  int cur_b = t0 < t1 ? t0 / tiles_per_img : 0;
  auto flush = [&](int b) {
    float* base = p.stats + ((long)(b * p.nslots + wave % p.nslots) * p.ldstats) * 2;
#pragma unroll
    for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float u = s1[pr][j], v = s2[pr][j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          u += __shfl_xor(u, o);
          v += __shfl_xor(v, o);
        }
        if (a == 0) {
          atomicAdd(base + (n0 + pr * 32 + g * 8 + j) * 2, u);
          atomicAdd(base + (n0 + pr * 32 + g * 8 + j) * 2 + 1, v);
        }
        s1[pr][j] = s2[pr][j] = 0.f;
      }
  };
  constexpr int PF = 2;
  uint4 ring[PF][KS];
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (t0 + s < t1) load_tile(t0 + s, ring[s]);
  for (int base = t0; base < t1; base += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int t = base + s;
      if (t < t1) {
        const long prow = ((long)t * 16 + a);
        if (p.stats) {
          const int b = t / tiles_per_img;
          if (b != cur_b) {
            flush(cur_b);
            cur_b = b;
          }
        }
        f32x4 acc[PAIRS][2];
#pragma unroll
        for (int pr = 0; pr < PAIRS; ++pr) {
          acc[pr][0] = acc[pr][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            acc[pr][0] = Mfma<T>::run(wf[pr][0][ks], ring[s][ks], acc[pr][0]);
            acc[pr][1] = Mfma<T>::run(wf[pr][1][ks], ring[s][ks], acc[pr][1]);
          }
        }
        if (t + PF < t1) load_tile(t + PF, ring[s]);
#pragma unroll
        for (int pr = 0; pr < PAIRS; ++pr) {
          float o[8] = {acc[pr][0][0], acc[pr][0][1], acc[pr][0][2], acc[pr][0][3], acc[pr][1][0], acc[pr][1][1], acc[pr][1][2], acc[pr][1][3]};
          float r8[8];
          if (res) unpack8<T>(*reinterpret_cast<const uint4*>(res + prow * p.ldres + n0 + pr * 32 + g * 8), r8);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            o[j] = p.alpha * o[j] + bs[pr][j] + (res ? p.res_scale * r8[j] : 0.f);
            s1[pr][j] += o[j];
            s2[pr][j] += o[j] * o[j];
          }
          *reinterpret_cast<uint4*>(y + prow * p.ldy + n0 + pr * 32 + g * 8) = pack8<T>(o);
        }
      }
    }
  }
  if (p.stats && t0 < t1) flush(cur_b);
}

template <typename T>
void launch_c8(const ConvP& p, hipStream_t st) {
  const int ntiles = p.M / 16;
  const int ngroups = p.N / 64;
  int bx = (ntiles + 3) / 4;
  const int cap = 256 * 8 / ngroups > 64 ? 256 * 8 / ngroups : 64;
  if (bx > cap) bx = cap;
  hipLaunchKernelGGL((conv3x3_c8_stream_kernel<T, 2>), dim3(bx, ngroups), dim3(256), 0, st, p, ntiles);
}

// (tried: the same LDS-free scheme for the 3x3 / 64 -> 64 layers at full resolution, weights of 32 output channels in registers and the
//  nine shifted pixel reads served by L1 / L2: 714 us against 208 us for the halo-resident kernel -- the 9x operand re-read through the
//  cache hierarchy costs far more than the halo kernel's unoverlapped phases; not kept)

template <typename T, int PAIRS, int KS>
void launch_1x1(const ConvP& p, hipStream_t st) {
  const int ntiles = p.M / 16;
  const int ngroups = p.N / (PAIRS * 32);
  int bx = (ntiles + 3) / 4;                 // one tile per wave at least
  const int cap = 256 * 8 / ngroups > 64 ? 256 * 8 / ngroups : 64;
  if (bx > cap) bx = cap;
  if (p.aab) {
    jg_note_kernel("conv1x1_stream_kernel+gn_apply");
    hipLaunchKernelGGL((conv1x1_stream_kernel<T, PAIRS, KS, true>), dim3(bx, ngroups), dim3(256), 0, st, p, ntiles);
  } else if (p.bgx) {
    jg_note_kernel("conv1x1_stream_kernel+gn_bwd_apply");
    hipLaunchKernelGGL((conv1x1_stream_kernel<T, PAIRS, KS, false, true>), dim3(bx, ngroups), dim3(256), 0, st, p, ntiles);
  } else {
    jg_note_kernel("conv1x1_stream_kernel");
    hipLaunchKernelGGL((conv1x1_stream_kernel<T, PAIRS, KS>), dim3(bx, ngroups), dim3(256), 0, st, p, ntiles);
  }
}

// the plain streaming kernel only (no fused GroupNorm passes): the extra (PAIRS, KS) combinations of round 5
template <typename T, int PAIRS, int KS>
void launch_1x1_plain(const ConvP& p, hipStream_t st) {
  const int ntiles = p.M / 16;
  const int ngroups = p.N / (PAIRS * 32);
  int bx = (ntiles + 3) / 4;
  const int cap = 256 * 8 / ngroups > 64 ? 256 * 8 / ngroups : 64;
  if (bx > cap) bx = cap;
  jg_note_kernel("conv1x1_stream_kernel");
  hipLaunchKernelGGL((conv1x1_stream_kernel<T, PAIRS, KS>), dim3(bx, ngroups), dim3(256), 0, st, p, ntiles);
}

template <typename T>
bool dispatch_1x1(const ConvP& p, hipStream_t st) {
  const int ks = p.Cin / 32;
  const bool wide = p.N % 128 == 0;
  // round 5: 32-channel inputs (KS 1) and outputs that are a multiple of 32 but not of 64 (PAIRS 1) -- the stage-1 / stage-2 linear layers
  // of the SegFormer generator (32 -> 32 / 128, 128 -> 32, 64 -> 64 / 256, 256 -> 64 on 65 k - 262 k tokens), which the implicit-GEMM
  // kernel streams at 2.0 TB/s (profiles/r05_cut_pmc_hbm_traffic.md)
  if (!p.aab && !p.bgx) {
    if (ks == 1) {
      if (wide) launch_1x1_plain<T, 4, 1>(p, st);
      else if (p.N % 64 == 0) launch_1x1_plain<T, 2, 1>(p, st);
      else launch_1x1_plain<T, 1, 1>(p, st);
      return true;
    }
    if (p.N % 64) {
      switch (ks) {
        case 2: launch_1x1_plain<T, 1, 2>(p, st); return true;
        case 4: launch_1x1_plain<T, 1, 4>(p, st); return true;
        case 8: launch_1x1_plain<T, 1, 8>(p, st); return true;
        default: return false;
      }
    }
  } else if (ks == 1 || p.N % 64) {
    return false;
  }
  switch (ks) {
    case 2: wide ? launch_1x1<T, 4, 2>(p, st) : launch_1x1<T, 2, 2>(p, st); return true;
    case 4: launch_1x1<T, 2, 4>(p, st); return true;      // <4, 4> would need 256+ VGPRs
    case 6: launch_1x1<T, 2, 6>(p, st); return true;
    case 8: launch_1x1<T, 2, 8>(p, st); return true;
    default: return false;
  }
}

}  // namespace

// true when the shape was handled here (gemm_nt.hip falls through to the generic kernel otherwise)
bool jg_conv1x1_try(int dtype, const ConvP& p, int nbatch, hipStream_t st) {
  if (p.res_up || p.x_up || p.y_pool) return false;   // half-resolution residuals / inputs: LDS-staged kernels only
  const bool off = jg_tune(JG_TUNE_CONV1X1) == 0;
  if (off && !p.aab && !p.bgx) return false;
  if (!p.aab && !p.bgx && nbatch == 1 && p.nh == 1 && p.R == 3 && p.S == 3 && p.pad == 1 && p.stride == 1 && !p.out_f32 && !(p.stats && p.stats_mode) && !p.reflect &&
      p.Cin == 8 && ((p.H * p.W) & 15) == 0 &&
      p.ldx == 8 && p.ldw == 72 && p.N % 64 == 0 && !(p.M & 15) && p.H == p.Ho && p.W == p.Wo && p.ldy % 8 == 0 && (!p.res || p.ldres % 8 == 0) &&
      (long)p.M >= 65536) {
    if (dtype == JG_F16) launch_c8<f16_t>(p, st);
    else if (dtype == JG_BF16) launch_c8<bf16_t>(p, st);
    else return false;
    return true;
  }
  if (nbatch != 1 || p.nh != 1 || p.R != 1 || p.S != 1 || p.pad != 0 || p.stride != 1 || p.out_f32 || p.stats || p.reflect) return false;
  if (p.Cin % 32 || p.Cin > 256 || p.N % 32 || (p.M & 15)) return false;
  if (p.ldx % 8 || p.ldy % 8 || p.ldw % 8 || (p.res && p.ldres % 8)) return false;
  // only where the layer is memory-bound: few hundred FLOP per byte; deep / low-resolution layers stay on the MFMA-tiled GEMM
  if ((long)p.M < 65536) return false;
  // round 6: 256 -> >= 256 channels is GEMM-shaped (128 FLOP per byte): the LDS-tiled kernel is faster there (the point-wise convolutions of the
  // mobile ResNet blocks at 64 x 64: 66 -> ~45 us, mobile_resnet_attn + [projected_d, basic] 347.5 -> 354.2 images/s); the fused GroupNorm forms
  // exist only here and stay
  if (!p.aab && !p.bgx && p.Cin >= 256 && p.N >= 256 && jg_tune(JG_TUNE_CONV1X1) < 2) return false;
  if (dtype == JG_F16) return dispatch_1x1<f16_t>(p, st);
  if (dtype == JG_BF16) return dispatch_1x1<bf16_t>(p, st);
  return false;
}
