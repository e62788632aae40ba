// Sliding-window form of the halo-resident 3x3 weight gradient (round 6): the 16-row x 64-co tile of wgrad_halo.hip with the LDS
// fragment traffic of its MFMA loop cut by 2.3x.
//
//   dw[co][r][s][ci] += alpha * sum_{b,oh,ow} dy[b,oh,ow,co] * x[b,oh+r-1,ow+s-1,ci]
//
// wgrad3x3_halo_kernel<16 rows, 64 co> reads 11 operand fragments (1 KB each, two ds_read_b64_tr_b16 per fragment) for every 18
// v_mfma_f32_16x16x32: 611 B of LDS reads per 16-cycle MFMA, 153 B/clk/CU at the MFMA peak, from 8-byte reads that need ~4 waves per
// SIMD to reach their rate (MI355X_MICROARCH.md, LDS) while the tile's 72 accumulators allow two.  Its MFMA loop alone runs at 118 us
// where the pipe time is 70 us, and the LDS-DMA of the next tile does not hide under it (profiles/r05_wgrad_cfg_ab.txt): the LDS is the
// busy unit, not the MFMA pipe.
//
// Here the K step is ONE pixel row of the tile (16 pixels = the K of v_mfma_f32_32x32x16) and a wave owns a 32 co x 32 ci block of all
// nine taps (9 x 16 accumulators).  The x fragment of halo row h at column shift s is the operand of tap (r, s) for the dy row h - r,
// r = 0, 1, 2: it is read ONCE and multiplied with the three dy-row fragments of a register window that slides down the tile.  Per halo
// row a wave reads 3 x fragments + 1 dy fragment (4 KB) for 9 MFMAs of 32 cycles: 14 B per MFMA cycle and wave against 38 before.
// The 8 waves of a workgroup are 2 row halves x 2 co blocks x 2 ci blocks; the two row halves of a block are summed through LDS (the
// tile buffers are free by then) so that the atomic volume is that of the kernel it replaces.
//
// LDS images, LDS-DMA issue, tile walk, split-K policy and the XMODE forms (mirrored borders, upsample-on-read) are those of
// wgrad_halo.hip; the swizzle differs because a 32-lane service group of the transposing read now covers 4 pixels x 64 B (two 32-byte
// channel blocks of one pixel side by side): the 64-byte block index of a 128-byte pixel row is XORed with bit 1 of the pixel column,
// which with the parity of the column (the pixel stride is 128 B = half a bank row) spreads any four consecutive columns over the four
// 64-byte bank quarters -- for every tap shift.
#include "conv_params.h"
#include "wgrad_params.h"
#include <type_traits>

namespace {

__device__ uint4 jg_sw_zero_page = {0u, 0u, 0u, 0u};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ONE v_mfma_f32_32x32x16 as a pinned statement (mfma_pipe.h::jg_mfma_pinned for the 32x32 shape): the compiler keeps the source order
// of the transposing LDS reads and the MFMAs and still counts lgkmcnt for the reads (they stay builtins)
template <typename T>
__device__ __forceinline__ void mfma32_pinned(f32x16& c, const uint4& a, const uint4& b) {
  const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  if constexpr (std::is_same<T, bf16_t>::value) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv) : "memory");
  } else {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv) : "memory");
  }
}

// 64-byte block swizzle of a 128-byte pixel row, by pixel column
__device__ __forceinline__ int fx(int c) { return (c >> 1) & 1; }

// sum of the 8 16-bit values of a fragment register quad (bias gradient: the dy fragment of a lane is 8 pixels of one output channel)
template <typename T> __device__ __forceinline__ float sum8(const uint4& v) {
  float f[8];
  unpack8<T>(v, f);
  return ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
}

// XMODE 0: zero padding; 1: mirrored borders; 2: x is the half-resolution tensor read through the nearest-upsample map
//
// Measured and taken out again (profiles/r06_wgrad_sw_ab.txt): a 4-wave form (one wave per SIMD walks all 16 rows; the MFMA loop
// alone is as fast as with two waves per SIMD, 109 vs 114 us, but its 19 LDS-DMA rounds per wave cost 53 us per launch and hipcc spills
// around them, each reload draining vmcnt and with it the DMA in flight), and the LDS-DMA rounds of the next tile spread between the MFMA
// steps of the current one (tools/glds_issue_probe.hip: one round in front of 8 MFMAs costs 14 - 28 cycles in isolation; inside this
// kernel the launch time did not move, 157.0 vs 157.3 us).
template <typename T, int XMODE>
__global__ __launch_bounds__(512, 1) void wgrad3x3_sw_kernel(WgP p, int ntiles, int per, int npairs, int ncot, int dbg) {
  constexpr bool REFLECT = XMODE == 1, UP = XMODE == 2;
  constexpr int TH = 16, NT = 512, BCO = 64;
  constexpr int RH = 8;                        // tile rows per wave (two row halves)
  constexpr int HW_ = 18, HPX = (TH + 2) * HW_;
  constexpr int HALO_CH = HPX * 8;             // 16-byte chunks of the halo (64 channels = 128 B / pixel)
  constexpr int DY_CH = TH * 16 * 8;           // dy tile: 256 pixels x 64 channels
  constexpr int A_ROUNDS = (HALO_CH + NT - 1) / NT, A_FULL = HALO_CH / NT, D_ROUNDS = DY_CH / NT;
  constexpr int BUF_CH = HALO_CH + DY_CH;
  constexpr int HROW = HW_ * 128, DROW = 16 * 128;   // bytes per halo row / dy tile row
  static_assert(2 * BUF_CH * 16 <= 163840, "LDS per CU");
  static_assert(4 * 144 * 64 * 4 <= 2 * BUF_CH * 16, "row-half reduction fits the tile buffers");
  __shared__ uint4 sm[2 * BUF_CH];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;    // row half, 32-co block, 32-ci block

  int id;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  }
  const int pair = id % npairs, slice = id / npairs;
  const int co0 = (pair % ncot) * BCO, ci0 = (pair / ncot) * 64;
  const int t0 = slice * per;
  const int t1 = min(ntiles, t0 + per);
  if (t0 >= t1) return;

  const T* __restrict__ xg = (const T*)p.x + ci0;
  const T* __restrict__ dyg = (const T*)p.dy + co0;
  const T* zp = reinterpret_cast<const T*>(&jg_sw_zero_page);
  typedef __attribute__((address_space(3))) char* lds_cptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0];
  const char* smb = reinterpret_cast<const char*>(&sm[0]);
  const int tw = p.W >> 4, th = p.H / TH;

  // ---- per-thread LDS-DMA source coordinates relative to the tile origin (LDS-DMA writes lane-linearly: the swizzle is on the source).
  // Kept PACKED -- halo rounds as (hy << 8 | hx) halves of a register, the dy rounds as one offset + a uniform stride: every register
  // this workgroup does not hold is one the waves of the other stream can use ----
  constexpr int A_PK = (A_ROUNDS + 1) / 2;
  unsigned a_pk[A_PK];
#pragma unroll
  for (int k = 0; k < A_PK; ++k) a_pk[k] = 0u;
#pragma unroll
  for (int rd = 0; rd < A_ROUNDS; ++rd) {
    const int pos = rd * NT + tid;
    const int hp = pos >> 3;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const unsigned v = (pos < HALO_CH) ? (unsigned)((hy << 8) | hx) : 0xffffu;
    a_pk[rd >> 1] |= v << ((rd & 1) * 16);
  }
  const int cpos = tid & 7;
  // dy: pos = rd * NT + tid -> pixel (rd * NT / 128 + tid / 128, (tid / 8) % 16), chunk cpos: the same column and chunk in every round
  const int d_xx = (tid >> 3) & 15;
  const int d_chunk = (((cpos >> 2) ^ fx(d_xx)) << 2) | (cpos & 3);
  const int d_rel0 = co0 + d_chunk * 8 < p.Cout ? ((tid >> 7) * p.W + d_xx) * (int)p.lddy + d_chunk * 8 : -1;     // narrow heads: missing channels read the zero page
  const int d_step = (NT / 128) * p.W * (int)p.lddy;

  // tile origin of the next LDS-DMA target (wave-uniform: scalar registers)
  struct TileO { const T* xb; const T* db; int oh0, ow0; unsigned l0; };
  auto tile_origin = [&](int t, int buf) -> TileO {
    const int tx = t % tw;
    const int r2 = t / tw;
    const int ty = r2 % th;
    const int b = r2 / th;
    TileO o;
    o.oh0 = ty * TH; o.ow0 = tx << 4;
    const long pix = ((long)b * p.H + o.oh0) * p.W + o.ow0;
    o.xb = UP ? xg + (((long)b * (p.H >> 1) + (o.oh0 >> 1)) * (p.W >> 1) + (o.ow0 >> 1)) * p.ldx : xg + pix * p.ldx;
    o.db = dyg + pix * p.lddy;
    o.l0 = lds0 + (buf * BUF_CH + wave * 64) * 16;
    return o;
  };
  // ONE LDS-DMA round (rd < D_ROUNDS: dy, then the halo rounds); rd is a compile-time constant at every call site
  auto issue_round = [&](const TileO& o, int rd) {
    if (rd < D_ROUNDS) {
      glds16(d_rel0 >= 0 ? o.db + (d_rel0 + rd * d_step) : zp, o.l0 + (HALO_CH + rd * NT) * 16);
    } else {
      const int ra = rd - D_ROUNDS;
      const int yx = (int)((a_pk[ra >> 1] >> ((ra & 1) * 16)) & 0xffffu);
      if (ra < A_FULL || yx != 0xffff) {
        const int hy = yx >> 8, hx = yx & 255;
        const int chunk = (((cpos >> 2) ^ fx(hx)) << 2) | (cpos & 3);
        int ih = o.oh0 - 1 + hy, iw = o.ow0 - 1 + hx;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        if constexpr (REFLECT) {
          ih = JG_REFLECT1(ih, p.H);
          iw = JG_REFLECT1(iw, p.W);
        }
        const int rel = UP ? (((ih >> 1) - (o.oh0 >> 1)) * (p.W >> 1) + ((iw >> 1) - (o.ow0 >> 1))) : ((ih - o.oh0) * p.W + (iw - o.ow0));
        glds16((ok || REFLECT) ? o.xb + (rel * (int)p.ldx + chunk * 8) : zp, o.l0 + ra * NT * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);       // one round's address arithmetic at a time (hoisted, the address pairs of a tile spill)
  };
  auto issue_tile = [&](int t, int buf) {
    const TileO o = tile_origin(t, buf);
#pragma unroll
    for (int rd = 0; rd < D_ROUNDS + A_ROUNDS; ++rd) issue_round(o, rd);
  };
  // LDS-DMA instructions a wave issues per tile (wave-uniform): the count its vmcnt wait leaves in flight
  const bool partial = (A_ROUNDS > A_FULL) && (A_FULL * NT + wave * 64 < HALO_CH);

  // ---- fragment byte offsets inside a buffer: lane = 16 j + i; 16-lane group j reads the 4-pixel x 16-channel block of channel half
  // (j & 1) and pixel octet (j >> 1) and leaves lane i with channel i of it (MFMA row / column = lane % 32, k group = lane / 32) ----
  const int i16 = lane & 15, j = lane >> 4;
  int abase[2], bbase[3][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int xq = (j >> 1) * 8 + rd * 4 + (i16 >> 2);
    abase[rd] = HALO_CH * 16 + (wr * 8 * 16 + xq) * 128 + (((wm ^ fx(xq)) << 6) | ((j & 1) << 5)) + (i16 & 3) * 8;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int hx = xq + s3;
      bbase[s3][rd] = (wr * 8 * HW_ + hx) * 128 + (((wn ^ fx(hx)) << 6) | ((j & 1) << 5)) + (i16 & 3) * 8;
    }
  }
  auto tr_frag = [&](int off0, int off1) -> uint4 {
    const uint2 u0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smb + off0)));
    const uint2 u1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smb + off1)));
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
  };

  f32x16 acc[9];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t9][q] = 0.f;
  float bsum = 0.f;
  const bool do_bias = p.dbias != nullptr && ci0 == 0 && wn == 0;

  issue_tile(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = t + 1 < t1 && !(dbg & 4);      // JG_HALO_DBG 4: timing without the per-tile LDS-DMA (first tile only)
    if (more) issue_tile(t + 1, buf ^ 1);
    if (more) {
      if (partial) wait_vmcnt<D_ROUNDS + A_FULL + 1>(); else wait_vmcnt<D_ROUNDS + A_FULL>();
    } else {
      wait_vmcnt<0>();
    }
    if (!(dbg & 8)) __builtin_amdgcn_s_barrier();              // JG_HALO_DBG 8 (with 4): timing without the per-tile barriers
    const int boff = buf * (BUF_CH * 16);
    if (dbg & 2) {                                            // JG_HALO_DBG 2: timing without the MFMAs and fragment reads
      __builtin_amdgcn_s_barrier();
      continue;
    }
    // halo rows hl = 0..9 of this wave's row half (tile rows y = 0..7): step hl multiplies the x fragments (hl, s) with the dy rows
    // hl - r; the fragments of step hl + 1 are requested before the MFMAs of step hl issue
    // Single-buffered x fragments (round 6b: every register this kernel gives back is room for the waves of the other stream): the
    // MFMAs of a step go column by column -- (r = 2, 1, 0; s) for s = 0, 1, 2 -- and xf[s] is reloaded for the next halo row right behind its
    // column, a whole step (9 MFMAs) ahead of its next use; the dy rows sit in a ring of four, row hl + 1 requested at the top of step hl.
    uint4 xf[3], dyf[4];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) xf[s3] = tr_frag(boff + bbase[s3][0], boff + bbase[s3][1]);
    dyf[0] = tr_frag(boff + abase[0], boff + abase[1]);
#pragma unroll
    for (int hl = 0; hl < RH + 2; ++hl) {
      if (hl + 1 < RH) dyf[(hl + 1) & 3] = tr_frag(boff + abase[0] + (hl + 1) * DROW, boff + abase[1] + (hl + 1) * DROW);
      if (hl < RH && do_bias) bsum += sum8<T>(dyf[hl & 3]);
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
        for (int r = 2; r >= 0; --r) {
          const int y = hl - r;
          if (y >= 0 && y < RH) mfma32_pinned<T>(acc[r * 3 + s3], dyf[y & 3], xf[s3]);
        }
        if (hl + 1 < RH + 2) xf[s3] = tr_frag(boff + bbase[s3][0] + (hl + 1) * HROW, boff + bbase[s3][1] + (hl + 1) * HROW);
      }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // the compiler does not see the MFMAs: keep its next VALU access off their results
    if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: the two row halves of a (co block, ci block) are summed through LDS -- the upper half keeps taps 5..8 and hands over
  // 0..4, the lower half the other way round -- then D row = co = 8 (q / 4) + 4 (lane / 32) + q % 4, col = ci = lane % 32 --------------
  if (dbg & 1) return;      // JG_HALO_DBG 1: timing without the atomic epilogue
  float* dw = (float*)p.dw;
  const int ci = ci0 + wn * 32 + (lane & 31);
  const int cob = co0 + wm * 32 + (lane >> 5) * 4;
  {
    float4* red = reinterpret_cast<float4*>(&sm[0]) + (size_t)(wm * 2 + wn) * (9 * 4 * 64) + lane;
    if (wr == 0) {
#pragma unroll
      for (int t9 = 5; t9 < 9; ++t9)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) red[(t9 * 4 + q4) * 64] = make_float4(acc[t9][q4 * 4], acc[t9][q4 * 4 + 1], acc[t9][q4 * 4 + 2], acc[t9][q4 * 4 + 3]);
    } else {
#pragma unroll
      for (int t9 = 0; t9 < 5; ++t9)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) red[(t9 * 4 + q4) * 64] = make_float4(acc[t9][q4 * 4], acc[t9][q4 * 4 + 1], acc[t9][q4 * 4 + 2], acc[t9][q4 * 4 + 3]);
    }
    __syncthreads();
    auto flush = [&](int t9) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 o = red[(t9 * 4 + q4) * 64];
        const float v[4] = {acc[t9][q4 * 4] + o.x, acc[t9][q4 * 4 + 1] + o.y, acc[t9][q4 * 4 + 2] + o.z, acc[t9][q4 * 4 + 3] + o.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cob + q4 * 8 + q;
          if (co < p.Cout_out && ci < p.Cin_out) atomicAdd(dw + (long)co * p.lddw + (long)t9 * p.Cin_out + ci, p.alpha * v[q]);
        }
      }
    };
    if (wr == 0) {
#pragma unroll
      for (int t9 = 0; t9 < 5; ++t9) flush(t9);
    } else {
#pragma unroll
      for (int t9 = 5; t9 < 9; ++t9) flush(t9);
    }
  }
  if (do_bias) {
    const int co = co0 + wm * 32 + (lane & 31);
    if (co < p.Cout_out) atomicAdd(p.dbias + co, p.dbias_scale * bsum);
  }
}

// Split-K choice of wgrad_halo.hip: one workgroup per CU, a block walks `per` tiles and pays a fixed prologue + epilogue of about OVH tiles
static void pick_split(int npairs, int ntiles, int ovh, int slots, int* per_out, int* splitk_out) {
  long best = -1;
  int bper = ntiles, bsk = 1;
  if (jg_tune(JG_TUNE_DETERMINISTIC) != 0) {     // no split over the tiles: one workgroup, one thread per element of dw -- a reproducible sum
    *per_out = ntiles; *splitk_out = 1;
    return;
  }
  const int skmax = ntiles < 2048 / npairs + 1 ? ntiles : 2048 / npairs + 1;
  for (int sk = 1; sk <= skmax; ++sk) {
    const int per = (ntiles + sk - 1) / sk;
    const int ske = (ntiles + per - 1) / per;
    const long blocks = (long)npairs * ske;
    const long rounds = (blocks + slots - 1) / slots;
    const long cost = (long)(per + ovh) * rounds;
    if (best < 0 || cost < best) { best = cost; bper = per; bsk = ske; }
  }
  *per_out = bper; *splitk_out = bsk;
}

template <typename T, int XMODE>
void launch_sw(const WgP& p, hipStream_t st) {
  const int ncot = (p.Cout + 63) / 64, npairs = ncot * (p.Cin / 64);
  const int ntiles = p.B * (p.H / 16) * (p.W >> 4);
  int per, splitk;
  pick_split(npairs, ntiles, 6, 256, &per, &splitk);
  hipLaunchKernelGGL((wgrad3x3_sw_kernel<T, XMODE>), dim3(npairs * splitk), dim3(512), 0, st, p, ntiles, per, npairs, ncot, jg_tune(JG_TUNE_HALO_DBG));
}

template <typename T>
void dispatch_sw(const WgP& p, hipStream_t st) {
  if (p.reflect) launch_sw<T, 1>(p, st);
  else if (p.x_up) launch_sw<T, 2>(p, st);
  else launch_sw<T, 0>(p, st);
}

}  // namespace

// Called by wgrad_halo.hip's dispatch for the shapes it has already validated (3x3, stride 1, pad 1, Cin % 64 == 0, H, W % 16 == 0).
void jg_wgrad_sw_launch(int dtype, const WgP& p, hipStream_t st) {
  if (dtype == JG_F16) dispatch_sw<f16_t>(p, st);
  else dispatch_sw<bf16_t>(p, st);
}
