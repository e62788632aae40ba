// Persistent 3x3 convolution for Cin == 64 (stride 1, pad 1): the full-resolution layers of the UNet (64 -> 64 at 256x256 is the most
// expensive shape of the training step: 14 launches) and the input gradients of the layers with 64 output channels.
//
// Why a second kernel.  With ONE 64-channel chunk a tile of conv_halo.hip is prologue -> 9 K-steps -> epilogue with nothing to overlap:
// per tile (measured, tools/halo_ablate.py) ~2.4 us of exposed halo-DMA latency, 9 K-steps that wait for their weight tile (the MFMAs of
// a step are shorter than the L2 -> LDS latency of the weight ring), and an epilogue as long as the K loop; two workgroups per CU run
// these phases in lockstep.  264 us per launch = 0.23 of the MFMA peak at 2.2 TB/s: neither roof.
//
// Here a workgroup is PERSISTENT (one per CU) and keeps, for its 64 output channels,
//   * the whole weight set resident in LDS (64 rows x 9 taps x 128 B = 72 KB, loaded once): no weight ring, no barrier per tap;
//   * two halo buffers (2 x 40.5 KB), one per WAVE GROUP.  The 8 waves form two groups of 4 (one wave per SIMD each) that work on
//     alternating tiles in ANTI-PHASE: in every phase one group runs the 288 MFMAs per wave of its tile out of its halo buffer while the
//     other group drains the tile it computed in the previous phase and DMAs its next halo into its (now idle) buffer.  One workgroup barrier per phase; both groups
//     execute the same number of barriers whatever the tile count.  The MFMA pipes see a computing wave in every phase, HBM sees the
//     other group's stores / residual reads / halo loads underneath.
// A wave owns 64 pixels (4 tile rows) x 64 channels, like the 64-wide configuration of conv_halo.hip (same swizzled halo image, same
// fragment reads).  Its epilogue works from the accumulator layout without an LDS transposition (8-byte accesses: a drain phase has a
// whole MFMA phase of the other group to issue them), so the group's halo buffer is free as soon as its MFMA phase ends and the DMA
// of its next tile is in flight underneath the epilogue.
//
// The residual enters through the accumulators (requested one phase ahead into the registers they vacate), and the fused GroupNorm
// statistics stay in registers as per-lane partials for as long as the workgroup's contiguous run of tiles stays in one image.
// Supported: bias, residual (same resolution), alpha / res_scale, fused GroupNorm statistics of the output (stats_mode 0).  Everything
// else (upsample-on-read, pooled stores, reflect padding, reduction mode, fp32 output) stays on conv_halo.hip.
#include "conv_params.h"
#include "conv_epilogue.h"

namespace {

constexpr int HW_ = 18;
constexpr int HALO_CH = HW_ * HW_ * 8;     // 2592 16-byte chunks per halo buffer
constexpr int W_CH = 64 * 9 * 8;           // 4608 chunks: [tap][row][8]
constexpr int REGION = HALO_CH / 4;        // 648 chunks of a halo buffer are loaded (and later used as scratch) by one wave
constexpr int A_RD = (REGION + 63) / 64;   // 11 LDS-DMA rounds per wave and tile

__device__ uint4 jg_p64_zero_page = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct TileRef { int b, oh0, ow0, sp; };

template <typename T>
__global__ __launch_bounds__(512, 1) void conv3x3_p64_kernel(ConvP p, int nsp, int Gs, int tilesN) {
  __shared__ uint4 sm[W_CH + 2 * HALO_CH + 16];     // weights | halo of group 0 | halo of group 1 | bias
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, w = wave & 3;
  const int nb = blockIdx.x % tilesN, stream = blockIdx.x / tilesN;
  const int n0 = nb * 64;
  // a workgroup owns a CONTIGUOUS run of spatial tiles (uniform over its waves): consecutive tiles share halo columns / rows in L2, and
  // the run stays inside one image for long stretches, so the GroupNorm statistics can stay in registers (see the drain phase)
  const int per = (nsp + Gs - 1) / Gs;
  const int ntl = stream * per < nsp ? (nsp - stream * per < per ? nsp - stream * per : per) : 0;
  const int tw = p.W >> 4, th = p.H >> 4;

  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ wt = (const T*)p.w;
  const T* zp = reinterpret_cast<const T*>(&jg_p64_zero_page);
  typedef __attribute__((address_space(3))) char* lds_cptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0];
  char* smc = reinterpret_cast<char*>(&sm[0]);
  const unsigned hbyte = (W_CH + g * HALO_CH) * 16;          // this group's halo buffer (byte offset in sm)

  auto tile_of = [&](int k) {
    TileRef t;
    t.sp = stream * per + k;
    t.ow0 = (t.sp % tw) << 4;
    t.oh0 = ((t.sp / tw) % th) << 4;
    t.b = t.sp / (tw * th);
    return t;
  };

  // ---- halo DMA: wave w of a group fills chunks [w * 648, (w + 1) * 648) of the group's buffer --------------------
  // position -> (halo pixel, 16-byte chunk) is fixed; only the tile origin moves
  auto issue_halo = [&](const TileRef& t) {
    const long org = (((long)t.b * p.H + t.oh0) * p.W + t.ow0) * p.ldx;
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));     // opaque per call: keeps hipcc from hoisting the 11 rounds of position arithmetic out of the
                                         // tile loop (it then holds ~50 values across the MFMA phase and spills them)
#pragma unroll
    for (int rd = 0; rd < A_RD; ++rd) {
      const int q = rd * 64 + lane_v;
      if (q < REGION) {
        const int pos = w * REGION + q;
        const int hp = pos >> 3, cpos = pos & 7;
        const int hy = (hp * 3641) >> 16, hx = hp - hy * HW_;       // hp / 18 for hp < 324
        const int kc = cpos ^ ((hx >> 1) & 7);
        const int ih = t.oh0 - 1 + hy, iw = t.ow0 - 1 + hx;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const T* src = ok ? x + org + ((long)(hy - 1) * p.W + (hx - 1)) * p.ldx + kc * 8 : zp;
        glds16(src, lds0 + hbyte + (w * REGION + rd * 64) * 16);
      }
    }
  };

  // ---- prologue: weights of the 64 output channels (9 rounds of 512 chunks = one tap each), first halo of both groups ----
  {
    const int row = tid >> 3, cpos = tid & 7;
    const int kc = cpos ^ ((row >> 1) & 7);
    const bool ok = n0 + row < p.N;
    const T* src0 = wt + (long)(n0 + row) * p.ldw + kc * 8;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) glds16(ok ? src0 + tap * 64 : zp, lds0 + (tap * 512 + wave * 64) * 16);
  }
  if (g == 0 && ntl > 0) issue_halo(tile_of(0));     // group 1 fetches tile 1 in phase 0

  // fragment / accumulator coordinates of this lane
  const int l15 = lane & 15, lk = lane >> 4;
  // bias of the 64 output channels: kept in LDS (re-read per tile; 16 more live registers would push the MFMA phase into spilling)
  float* sbias = reinterpret_cast<float*>(smc + (W_CH + 2 * HALO_CH) * 16);
  if (tid < 64) sbias[tid] = (p.bias && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;

  // fragment offsets (as in conv_halo.hip, 64-wide configuration: TM = 4 tile rows of this wave, TN = 4 channel tiles)
  int bfrag[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = j * 16 + l15;
    bfrag[j] = row * 128 + ((lk ^ ((row >> 1) & 7)) << 4);
  }
  int afrag[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int hx = jg_pixperm(l15) + s3;
    afrag[s3] = ((w * 4) * HW_ + hx) * 128 + ((lk ^ ((hx >> 1) & 7)) << 4);
  }

  f32x4 acc[4][4];
  T* y = (T*)p.y;
  const T* res = (p.dbg & 32) ? nullptr : (const T*)p.res;
  uint2 rraw[4][4];          // residual rows of the tile this wave computes next (live only between a drain and the next MFMA phase)
  auto load_res = [&](const TileRef& t) {
    const long m0 = ((long)t.b * p.H + t.oh0 + w * 4) * p.W + t.ow0 + jg_pixperm(l15);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        rraw[i][j] = *reinterpret_cast<const uint2*>(res + (m0 + (long)i * p.W) * p.ldres + n0 + lk * 4 + j * 16);
  };
  // fused GroupNorm statistics: per-lane partial (sum, sum^2) of the lane's 16 channels over every pixel this wave has drained since
  // the last flush; reduced over the 16 pixel lanes and added to the global rows only when the image changes or the run ends
  // (a per-tile cross-lane reduction cost as much as the 288 MFMAs of the tile)
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) s1[j][q] = s2[j][q] = 0.f;
  int stat_b = -1;
  auto flush_stats = [&]() {
    if (stat_b < 0 || !p.stats) return;
    float* dst = p.stats + (((long)stat_b * p.nslots + (blockIdx.x * 8 + wave) % p.nslots) * p.ldstats + n0) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = s1[j][q], b = s2[j][q];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          b += __shfl_xor(b, o);
        }
        if (l15 == 0 && !(p.dbg & 64)) {
          atomicAdd(dst + (j * 16 + lk * 4 + q) * 2, a);
          atomicAdd(dst + (j * 16 + lk * 4 + q) * 2 + 1, b);
        }
        s1[j][q] = s2[j][q] = 0.f;
      }
  };
  wait_vmcnt<0>();
  __syncthreads();

  // ---- phases: in phase ph group (ph & 1) computes tile ph, the other group drains tile ph - 1 and fetches tile ph + 1 ----
  for (int ph = 0; ph <= ntl; ++ph) {
    if ((ph & 1) == g) {
      if (ph < ntl) {
        // ---------------- MFMA phase: 9 taps x 2 k-halves x (4 x 4) tiles, nothing in LDS changes meanwhile ----------------
        // accumulators start from the residual: y = alpha * (conv + (res_scale / alpha) * res) + bias.  Its rows were requested at
        // the end of this group's previous drain phase (the accumulator registers are dead in between), so they are here by now;
        // only the first tile of a group fetches them on the spot.
        if (res) {
          if (ph < 2) load_res(tile_of(ph));
          const float rs = p.res_scale / p.alpha;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float rf[4];
              unpack4<T>(rraw[i][j], rf);
              acc[j][i] = (f32x4){rs * rf[0], rs * rf[1], rs * rf[2], rs * rf[3]};
            }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (!(p.dbg & 2)) {
          const char* hb = smc + hbyte;
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s3 = tap % 3;
            asm volatile("" ::: "memory");     // keep the fragment reads of later taps behind this point (register pressure)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
              if (sub) asm volatile("" ::: "memory");
              uint4 fa[4], fb[4];
              const int a0 = afrag[s3] ^ (sub * 64);
#pragma unroll
              for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(hb + a0 + (i + r) * (HW_ * 128));
#pragma unroll
              for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(smc + tap * 8192 + (bfrag[j] ^ (sub * 64)));
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = Mfma<T>::run(fb[j], fa[i], acc[j][i]);
            }
          }
        }
      }
    } else {
      const int kt = ph - 1;     // the tile this group computed in the previous phase
      const bool drain = kt >= 0 && kt < ntl && !(p.dbg & 1);
      const TileRef t = tile_of(drain ? kt : 0);
      const long mrow0 = ((long)t.b * p.H + t.oh0 + w * 4) * p.W + t.ow0 + jg_pixperm(l15);
      // ---------------- drain phase.  The epilogue works straight from the accumulator layout (a lane holds 4 consecutive channels
      // of a pixel: 8-byte accesses) and needs NO LDS scratch, so this group's halo buffer is free from the start of the phase:
      // order = halo DMA of the group's next tile -> epilogue arithmetic and stores -> residual rows of the next tile (into the
      // registers the accumulators just left) -> wait for everything.
      const bool fetch = ph + 1 < ntl && !(p.dbg & 4);
      if (fetch) issue_halo(tile_of(ph + 1));
      if (drain) {
        if (t.b != stat_b) {
          flush_stats();
          stat_b = t.b;
        }
        T* ybase = y + mrow0 * p.ldy + n0 + lk * 4;
        const long ystep = (long)p.W * p.ldy;       // wave-uniform: one tile row down
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 bj = *reinterpret_cast<const float4*>(sbias + j * 16 + lk * 4);
          const float bq[4] = {bj.x, bj.y, bj.z, bj.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = p.alpha * acc[j][i][q] + bq[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              s1[j][q] += v[q];
              s2[j][q] += v[q] * v[q];
            }
            if (!(p.dbg & 8)) *reinterpret_cast<uint2*>(ybase + i * ystep + j * 16) = pack4<T>(v[0], v[1], v[2], v[3]);
          }
        }
      }
      // The halo of the next tile must have landed before the barrier; the residual rows need not.  vmcnt retires in order and the
      // DMA pieces are OLDER than the 16 residual requests issued here, so "at most 16 outstanding" == DMA landed, with the
      // residual rows still in flight across the barrier (hipcc waits for them, conservatively, at their first use).
      if (res && ph + 1 < ntl) {
        load_res(tile_of(ph + 1));
        wait_vmcnt<16>();
      } else {
        wait_vmcnt<0>();
      }
    }
    // raw barrier: what crosses it is LDS state only (halo landed: explicit counts above; fragment reads of the MFMA group retired
    // with its MFMAs).  __syncthreads() would add a release fence, i.e. vmcnt(0): the residual rows in flight would be waited for.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  flush_stats();
}

}  // namespace

bool jg_conv_p64_try(int dtype, const ConvP& p0, int nbatch, hipStream_t st) {
  // JG_PERSIST64: 0 off; 1 auto (>= 1024 spatial tiles, launches without a residual: with one the kernel is HBM-latency bound at the
  // same ~2.8 TB/s as conv_halo.hip, measured 282 vs 272 us); -1 auto including residual launches; >= 2: forced at any size with at
  // most that many spatial streams (tests: small grids where the workgroups still walk over several, ragged, numbers of tiles)
  int mode = jg_tune(JG_TUNE_PERSIST64);
  if (!mode) return false;
  if (mode == 1 && p0.res) return false;
  if (mode == -1) mode = 1;
  ConvP p = p0;
  p.dbg = jg_tune(JG_TUNE_HALO_DBG);
  if (nbatch != 1 || p.R != 3 || p.S != 3 || p.pad != 1 || p.stride != 1 || p.out_f32) return false;
  if (p.Cin != 64 || p.N % 64 || (p.H & 15) || (p.W & 15) || p.H != p.Ho || p.W != p.Wo) return false;
  if (p.x_up || p.y_pool || p.res_up || p.reflect || (p.stats && p.stats_mode != 0) || p.alpha == 0.f) return false;
  if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return false;
  const int tilesN = p.N / 64;
  const int nsp = p.B * (p.H >> 4) * (p.W >> 4);
  if ((mode == 1 && nsp < 1024) || tilesN > 4) return false;       // persistence pays from a few tiles per workgroup on
  int Gs = 256 / tilesN;
  if (mode >= 2 && Gs > mode) Gs = mode;
  if (Gs > nsp) Gs = nsp;
  const int grid = Gs * tilesN;
  jg_note_kernel("conv3x3_p64_kernel");
  if (dtype == JG_F16) hipLaunchKernelGGL((conv3x3_p64_kernel<f16_t>), dim3(grid), dim3(512), 0, st, p, nsp, Gs, tilesN);
  else if (dtype == JG_BF16) hipLaunchKernelGGL((conv3x3_p64_kernel<bf16_t>), dim3(grid), dim3(512), 0, st, p, nsp, Gs, tilesN);
  else return false;
  return true;
}
