// Halo-resident weight gradient of a KS x KS / stride 1 convolution with a LARGE kernel (7x7: the content / output heads of the CUT
// generators, resnet_generator.py:247-263, attn_network.py:6-54) on gfx950 MFMA.
//
//   dw[co][r][s][ci] += alpha * sum_{b,oh,ow} dy[b,oh,ow,co] * x[b,oh+r-pad,ow+s-pad,ci]          dbias[co] += sum dy[b,oh,ow,co]
//
// Why not gemm_tn.hip: its im2col tiling gives every 128 columns of (r, s, ci) -- two taps -- their own workgroups, so the input tensor
// is fetched once per tap pair: 49 x 268 MB for the 64 -> 27 content head at 256x256 x 32 images, 3.1 ms at 136 TFLOP/s, L2-traffic
// bound.  Why not wgrad_halo.hip as it is: 49 taps x its accumulator tile do not fit the register file.
//
// Here a workgroup (4 waves) owns 32 output channels x one 64-channel input chunk x a GROUP of RT tap rows (RT x KS taps: 14 accumulator
// tiles per co-block for 7x7 with RT = 2), and walks over a slice of TH x 16 spatial tiles (split over tiles, fp32 atomics at the end).
// Per tile the dy tile [TH*16 px][32] and the x halo [(TH+RT-1) x (16+KS-1) px][64] come into LDS once (LDS-DMA, double buffered) and all
// RT x KS taps read the halo at shifted positions: the input is fetched ceil(KS / RT) x 1.5 times instead of KS^2 / 2 times.
// LDS images, swizzles and the transposing fragment reads (ds_read_b64_tr_b16) are those of wgrad_halo.hip.
#include "wgrad_params.h"

namespace {

__device__ uint4 jg_wk_zero_page = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;

// 2-bit swizzle of a 128-byte halo pixel row (4 blocks of 32 B) by pixel column (wgrad_halo.hip)
__device__ __forceinline__ int f4(int c) { return ((c >> 1) & 1) | (((c >> 3) & 1) << 1); }
// 1-bit swizzle of a 64-byte dy pixel row (2 blocks of 32 B): a 32-lane service group reads pixels {h..h+3, h+8..h+11}
__device__ __forceinline__ int f2(int c) { return (c >> 3) & 1; }

template <typename T, int KS, int RT, int TH, int CPW>
__global__ __launch_bounds__(256, 2) void wgrad_kxk_halo_kernel(WgP p, int ntiles, int per, int npairs, int ncot, int ngroups) {
  constexpr int NT = 256;
  constexpr int BCO = CPW * 16;
  static_assert(BCO == 32, "dy rows of 64 bytes (the f2 swizzle)");
  constexpr int HW_ = 16 + KS - 1, HR = TH + RT - 1;
  constexpr int HALO_CH = HR * HW_ * 8;        // 16-byte chunks of the halo (64 channels = 128 B / pixel)
  constexpr int CPP = BCO / 8;                 // chunks per dy pixel
  constexpr int DY_ROWB = BCO * 2;
  constexpr int DY_CH = TH * 16 * CPP;
  constexpr int A_ROUNDS = (HALO_CH + NT - 1) / NT, A_FULL = HALO_CH / NT;
  constexpr int D_ROUNDS = DY_CH / NT;
  constexpr int BUF_CH = HALO_CH + DY_CH;
  constexpr int NSUB = TH / 2;                 // K-steps of 32 pixels (2 tile rows)
  static_assert(DY_CH % NT == 0 && A_ROUNDS - A_FULL <= 1, "staging rounds");
  static_assert(2 * BUF_CH * 16 * 2 <= 163840, "two workgroups per CU");
  __shared__ uint4 sm[2 * BUF_CH];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = the wave's 16-channel block of the input chunk
  const int id = blockIdx.x;
  const int pair = id % npairs, rest = id / npairs;
  const int grp = rest % ngroups, slice = rest / ngroups;
  const int co0 = (pair % ncot) * BCO, ci0 = (pair / ncot) * 64;
  const int r0 = grp * RT;                     // first tap row of this workgroup
  const int t0 = slice * per, t1 = min(ntiles, t0 + per);
  if (t0 >= t1) return;

  const T* __restrict__ xg = (const T*)p.x + ci0;
  const T* __restrict__ dyg = (const T*)p.dy + co0;
  const T* zp = reinterpret_cast<const T*>(&jg_wk_zero_page);
  typedef __attribute__((address_space(3))) char* lds_cptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0];
  const char* smb = reinterpret_cast<const char*>(&sm[0]);
  const int tw = p.Wo >> 4, th = p.Ho / TH;

  auto issue_tile = [&](int t, int buf) {
    const int tx = t % tw, r2 = t / tw, ty = r2 % th, b = r2 / th;
    const int oh0 = ty * TH, ow0 = tx << 4;
    const T* db = dyg + (((long)b * p.Ho + oh0) * p.Wo + ow0) * p.lddy;
    const T* xb = xg + (long)b * p.H * p.W * p.ldx;
    const unsigned l0 = lds0 + (buf * BUF_CH + wave * 64) * 16;
#pragma unroll
    for (int rd = 0; rd < D_ROUNDS; ++rd) {
      const int pos = rd * NT + tid;
      const int px = pos / CPP, cpos = pos % CPP;
      const int yy = px >> 4, xx = px & 15;
      const int chunk = (((cpos >> 1) ^ f2(xx)) << 1) | (cpos & 1);
      const bool ok = co0 + chunk * 8 < p.Cout;
      glds16(ok ? db + (yy * p.Wo + xx) * (int)p.lddy + chunk * 8 : zp, l0 + (HALO_CH + rd * NT) * 16);
    }
#pragma unroll
    for (int rd = 0; rd < A_ROUNDS; ++rd) {
      const int pos = rd * NT + tid;
      if (pos < HALO_CH) {
        const int hp = pos >> 3, cpos = pos & 7;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int chunk = (((cpos >> 1) ^ f4(hx)) << 1) | (cpos & 1);
        const int ih = oh0 + hy + r0 - p.pad, iw = ow0 + hx - p.pad;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        glds16(ok ? xb + ((long)ih * p.W + iw) * p.ldx + chunk * 8 : zp, l0 + rd * NT * 16);
      }
    }
  };
  // LDS-DMA instructions a wave issues per tile (wave-uniform): the count its vmcnt wait leaves in flight
  const bool partial = (A_ROUNDS > A_FULL) && (A_FULL * NT + wave * 64 < HALO_CH);

  // ---- fragment byte offsets inside a buffer (as wgrad_halo.hip) ---------------------------------------------------------
  const int i16 = lane & 15, g = lane >> 4;
  const int trow = g >> 1;
  int abase[CPW][2], bbase[KS][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int xq = (g & 1) * 8 + rd * 4 + (i16 >> 2);
#pragma unroll
    for (int i = 0; i < CPW; ++i) abase[i][rd] = HALO_CH * 16 + (trow * 16 + xq) * DY_ROWB + ((i ^ f2(xq)) << 5) + (i16 & 3) * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int hx = xq + s;
      bbase[s][rd] = (trow * HW_ + hx) * 128 + ((wave ^ f4(hx)) << 5) + (i16 & 3) * 8;
    }
  }
  auto tr_frag = [&](int off0, int off1) -> uint4 {
    const uint2 u0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smb + off0)));
    const uint2 u1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smb + off1)));
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
  };

  f32x4 acc[RT * KS][CPW], accb[CPW];
#pragma unroll
  for (int t9 = 0; t9 < RT * KS; ++t9)
#pragma unroll
    for (int i = 0; i < CPW; ++i) acc[t9][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CPW; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.dbias != nullptr && ci0 == 0 && grp == 0 && wave == 0;
  const uint32_t one1 = to_bits<T>(from_f32<T>(1.0f));
  const uint32_t one2 = one1 | (one1 << 16);
  const uint4 ones = make_uint4(one2, one2, one2, one2);

  issue_tile(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = t + 1 < t1;
    if (more) {
      issue_tile(t + 1, buf ^ 1);
      if (partial) wait_vmcnt<D_ROUNDS + A_FULL + 1>(); else wait_vmcnt<D_ROUNDS + A_FULL>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    const int boff = buf * (BUF_CH * 16);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      uint4 fa[CPW];
#pragma unroll
      for (int i = 0; i < CPW; ++i) fa[i] = tr_frag(boff + abase[i][0] + sub * 32 * DY_ROWB, boff + abase[i][1] + sub * 32 * DY_ROWB);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) accb[i] = Mfma<T>::run(fa[i], ones, accb[i]);
      }
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const uint4 fb = tr_frag(boff + bbase[s][0] + (2 * sub + r) * (HW_ * 128), boff + bbase[s][1] + (2 * sub + r) * (HW_ * 128));
#pragma unroll
          for (int i = 0; i < CPW; ++i) acc[r * KS + s][i] = Mfma<T>::run(fa[i], fb, acc[r * KS + s][i]);
        }
    }
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: D row = co = g*4 + q, col = ci = i16 ------------------------------------------------------------------------------
  float* dw = (float*)p.dw;
  const int ci = ci0 + wave * 16 + i16;
  if (ci < p.Cin_out) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      if (r0 + r >= KS) continue;       // the last group of a kernel height that RT does not divide
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < CPW; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int co = co0 + i * 16 + g * 4 + q;
            if (co < p.Cout_out) atomicAdd(dw + (long)co * p.lddw + (long)((r0 + r) * KS + s) * p.Cin_out + ci, p.alpha * acc[r * KS + s][i][q]);
          }
    }
  }
  if (do_bias) {
    // accb columns are all equal (B = ones): take column 0
    if (i16 == 0) {
#pragma unroll
      for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = co0 + i * 16 + g * 4 + q;
          if (co < p.Cout_out) atomicAdd(p.dbias + co, p.dbias_scale * accb[i][q]);
        }
    }
  }
}

// split over tiles: blocks run 2 per CU; a block pays a prologue + atomic epilogue worth about `ovh` tiles
void pick_split(int nwork, int ntiles, int ovh, int slots, int* per_out, int* splitk_out) {
  long best = -1;
  int bper = ntiles, bsk = 1;
  const int skmax = ntiles < 4096 / nwork + 1 ? ntiles : 4096 / nwork + 1;
  for (int sk = 1; sk <= skmax; ++sk) {
    const int per = (ntiles + sk - 1) / sk;
    const int ske = (ntiles + per - 1) / per;
    const long rounds = ((long)nwork * ske + slots - 1) / slots;
    const long cost = (long)(per + ovh) * rounds;
    if (best < 0 || cost < best) { best = cost; bper = per; bsk = ske; }
  }
  *per_out = bper; *splitk_out = bsk;
}

template <typename T, int KS, int RT>
void launch_kxk(const WgP& p, hipStream_t st) {
  constexpr int TH = 8, CPW = 2, BCO = 32;
  // (p.R == 1: the 1 x KS convolution of the row-packed heads -- one tap row, one group)
  const int ncot = (p.Cout + BCO - 1) / BCO, npairs = ncot * (p.Cin / 64), ngroups = p.R == 1 ? 1 : (KS + RT - 1) / RT;
  const int ntiles = p.B * (p.Ho / TH) * (p.Wo >> 4);
  int per, splitk;
  pick_split(npairs * ngroups, ntiles, 12, 512, &per, &splitk);
  hipLaunchKernelGGL((wgrad_kxk_halo_kernel<T, KS, RT, TH, CPW>), dim3(npairs * ngroups * splitk), dim3(256), 0, st, p, ntiles, per, npairs, ncot, ngroups);
}

}  // namespace

bool jg_wgrad_kxk_try(int dtype, const WgP& p, int nbatch, hipStream_t st, bool dry_run) {
  // 1 x 7 (round 6, the row-packed heads): pad 0; dy may carry zero rows below the image (Ho >= H: its row count is a multiple of the 8-row tile)
  const bool row1 = p.R == 1 && p.S == 7 && p.pad == 0 && p.Ho >= p.H;
  if (nbatch != 1 || (!row1 && (p.R != 7 || p.S != 7)) || p.stride != 1 || p.out_mode != JG_OUT_ATOMIC_F32 || p.reflect || p.x_up) return false;
  if (p.Cin % 64 || p.Cout % 8 || (p.Ho & 7) || (p.Wo & 15) || (!row1 && p.Ho != p.H + 2 * p.pad - 6) || p.Wo != p.W + 2 * p.pad - 6) return false;
  if ((long)p.B * p.Ho * p.Wo < 65536) return false;        // small maps: the generic kernel's split over pixels fills the chip better
  if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.B * p.Ho * p.Wo * p.lddy >= (1L << 31)) return false;
  if (dry_run) return dtype == JG_F16 || dtype == JG_BF16;
  if (row1) {
    if (dtype == JG_F16) launch_kxk<f16_t, 7, 1>(p, st);
    else if (dtype == JG_BF16) launch_kxk<bf16_t, 7, 1>(p, st);
    else return false;
    jg_note_kernel("wgrad_kxk_halo_kernel<1x7>");
    return true;
  }
  if (dtype == JG_F16) launch_kxk<f16_t, 7, 2>(p, st);
  else if (dtype == JG_BF16) launch_kxk<bf16_t, 7, 2>(p, st);
  else return false;
  jg_note_kernel("wgrad_kxk_halo_kernel<7x7,2 tap rows>");
  return true;
}
