// Halo-resident weight gradient of a 3x3 / stride 1 / pad 1 convolution on gfx950 MFMA.
//
//   dw[co][r][s][ci] += alpha * sum_{b,oh,ow} dy[b,oh,ow,co] * x[b,oh+r-1,ow+s-1,ci]
//   dbias[co]        +=         sum_{b,oh,ow} dy[b,oh,ow,co]
//
// The reduction index is the PIXEL index.  A workgroup owns BCO output channels x one 64-channel
// input chunk x ALL 9 taps, and walks over a slice of TH x 16 spatial tiles (split-K over tiles,
// fp32 atomics into the gradient arena at the end).  Per tile it brings the dy tile
// [TH*16 px][BCO] and the x HALO [(TH+2) x 18 px][64] into LDS once (LDS-DMA, double buffered across
// tiles) and every tap multiplies the same dy fragments with the halo read at a shifted position:
// 9 x fewer L2->LDS bytes per MFMA than one im2col tile per tap (gemm_tn.hip), no address math or
// bounds checks in the MFMA loop.
//
// Both operands are pixel-major in LDS (the NHWC order of global memory) while an MFMA fragment
// wants 8 consecutive k (= pixels) per lane: fragments come from the transposing LDS read
// ds_read_b64_tr_b16 (lane i of a 16-lane group addresses 4 channels of pixel k0 + i/4 and receives
// channel c0 + i of pixels k0..k0+3).  A 32-lane service group of that read touches 8 pixels
// {h..h+3, h+8..h+11} x 32 B; the 32-byte block index of a pixel row is XOR-swizzled with a function
// of the pixel COLUMN (so that vertical tap shifts are plain address offsets) that is injective on
// every such set: conflict-free for every tap.  LDS-DMA writes lane-linearly, so the swizzle is
// applied to the SOURCE chunk index (cdna_hip_programming.md 5.4 rule 21).
#include "conv_params.h"
#include "wgrad_params.h"
#include "mfma_pipe.h"
#include <cstdlib>

namespace {

__device__ uint4 jg_wg_zero_page = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;

// 2-bit swizzle of a 128-byte pixel row (4 blocks of 32 B), by pixel column c
__device__ __forceinline__ int f4(int c) { return ((c >> 1) & 1) | (((c >> 3) & 1) << 1); }
// 3-bit swizzle of a 256-byte pixel row (8 blocks of 32 B)
__device__ __forceinline__ int f8(int c) { return (c & 3) | (((c >> 3) & 1) << 2); }
// 1-bit swizzle of a 64-byte pixel row (2 blocks of 32 B): the 8 pixels {h..h+3, h+8..h+11} of a service group are four bank quarters
// (pixel % 4) x the two blocks, told apart by bit 3 of the column
__device__ __forceinline__ int f2(int c) { return (c >> 3) & 1; }
// block swizzle of a dy pixel row of BCO channels
template <int BCO> __device__ __forceinline__ int fdy(int c) { return BCO == 32 ? f2(c) : BCO == 64 ? f4(c) : f8(c); }

// XMODE 0: plain zero padding; 1: mirrored borders (REFLECT); 2: x is the half-resolution tensor read through the nearest-upsample map
//
// PIPE: the MFMA loop of a tile in software-pipelined form.  hipcc orders a sub-step as {reads of a few taps; s_waitcnt; 2 MFMAs; ...}: with
// only CPW MFMAs per weight-side fragment the LDS latency of every tap is exposed (SQ_VALU_MFMA_BUSY 0.37 for the 16-row tile,
// profiles/r02_mfma_busy.md).  Here the (sub-step, tap) pairs of a tile form ONE flat sequence of 9 * NSUB steps; the halo fragment of
// step i + LOOKAHEAD is requested before the MFMAs of step i are issued, and the dy fragments of the next sub-step at the start of the
// current one.  The MFMAs are `asm volatile` statements (with a memory clobber), which pins the source order of reads and MFMAs; the
// reads stay compiler-visible builtins, so every s_waitcnt lgkmcnt(N) is still the compiler's (LDS returns in order: exact counts).
template <typename T, int TH, int CPW, int WMR, int XMODE, bool PIPE = false>
__global__ __launch_bounds__(WMR * 256, 3 - WMR) void wgrad3x3_halo_kernel(WgP p, int ntiles, int per, int npairs, int ncot, int dbg) {
  constexpr bool REFLECT = XMODE == 1, UP = XMODE == 2;
  constexpr int NT = WMR * 256;                // WMR wave rows (output-channel groups) x 4 wave columns (16-channel ci tiles)
  constexpr int BCO = WMR * CPW * 16;          // output channels per block
  constexpr int HW_ = 18, HPX = (TH + 2) * HW_;
  constexpr int HALO_CH = HPX * 8;             // 16-byte chunks of the halo (64 channels = 128 B / pixel)
  constexpr int CPP = BCO / 8;                 // chunks per dy pixel row
  constexpr int DY_ROWB = BCO * 2;
  constexpr int DY_CH = TH * 16 * CPP;
  constexpr int A_ROUNDS = (HALO_CH + NT - 1) / NT;
  constexpr int A_FULL = HALO_CH / NT;         // rounds every wave takes part in
  constexpr int D_ROUNDS = DY_CH / NT;
  constexpr int BUF_CH = HALO_CH + DY_CH;
  constexpr int NSUB = TH / 2;                 // K-steps of 32 pixels (2 tile rows) per tile
  static_assert(DY_CH % NT == 0, "dy tile vs block size");
  static_assert(A_ROUNDS - A_FULL <= 1, "at most one partial halo round");

  static_assert(2 * BUF_CH * 16 * (3 - WMR) <= 163840, "LDS per CU");
  __shared__ uint4 sm[2 * BUF_CH];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // wm < WMR

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
  const T* zp = reinterpret_cast<const T*>(&jg_wg_zero_page);
  typedef __attribute__((address_space(3))) char* lds_cptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0];
  const char* smb = reinterpret_cast<const char*>(&sm[0]);
  const int tw = p.W >> 4, th = p.H / TH;

  // Two forms of the per-tile LDS-DMA issue; which one keeps the MFMA loop spill-free depends on the
  // configuration (hipcc hoists the recomputed coordinates of the 8-wave form back into registers).
  constexpr bool RECOMP = (WMR == 1);
  // ---- per-thread LDS-DMA source coordinates relative to the tile origin ---------------------------
  int a_rel[A_ROUNDS], a_yx[A_ROUNDS];
  if constexpr (!RECOMP) {
#pragma unroll
  for (int rd = 0; rd < A_ROUNDS; ++rd) {
    const int pos = rd * NT + tid;
    const int hp = pos >> 3, cpos = pos & 7;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int chunk = (((cpos >> 1) ^ f4(hx)) << 1) | (cpos & 1);
    // UP: tile origins are even, so (oh0 + hy - 1) >> 1 = oh0 / 2 + ((hy - 1) >> 1): offsets relative to the half-resolution origin
    a_rel[rd] = UP ? (((hy - 1) >> 1) * (p.W >> 1) + ((hx - 1) >> 1)) * (int)p.ldx + chunk * 8
                   : ((hy - 1) * p.W + (hx - 1)) * (int)p.ldx + chunk * 8;
    a_yx[rd] = (pos < HALO_CH) ? ((hy << 8) | hx) : -1;
  }
  }
  int d_rel[D_ROUNDS];
  if constexpr (!RECOMP) {
#pragma unroll
  for (int rd = 0; rd < D_ROUNDS; ++rd) {
    const int pos = rd * NT + tid;
    const int px = pos / CPP, cpos = pos % CPP;
    const int ty = px >> 4, xx = px & 15;
    const int blk = (cpos >> 1) ^ fdy<BCO>(xx);
    // (round 5: output-channel counts below the tile width -- the 8-channel head of the UNet -- read the zero page for the missing chunks)
    d_rel[rd] = co0 + ((blk << 1) | (cpos & 1)) * 8 < p.Cout ? (ty * p.W + xx) * (int)p.lddy + ((blk << 1) | (cpos & 1)) * 8 : -1;
  }
  }

  auto issue_tile = [&](int t, int buf) {
    if constexpr (RECOMP) {
      const int tx = t % tw;
      const int r2 = t / tw;
      const int ty = r2 % th;
      const int b = r2 / th;
      const int oh0 = ty * TH, ow0 = tx << 4;
      const long pix = ((long)b * p.H + oh0) * p.W + ow0;
      const T* xb = UP ? xg + (((long)b * (p.H >> 1) + (oh0 >> 1)) * (p.W >> 1) + (ow0 >> 1)) * p.ldx : xg + pix * p.ldx;
      const T* db = dyg + pix * p.lddy;
      const unsigned l0 = lds0 + (buf * BUF_CH + wave * 64) * 16;
  #pragma unroll
      for (int rd = 0; rd < D_ROUNDS; ++rd) {
        const int pos = rd * NT + tid;
        const int px = pos / CPP, cpos = pos % CPP;
        const int yy = px >> 4, xx = px & 15;
        const int blk = (cpos >> 1) ^ fdy<BCO>(xx);
        const int dch = ((blk << 1) | (cpos & 1)) * 8;
        glds16(co0 + dch < p.Cout ? db + (yy * p.W + xx) * (int)p.lddy + dch : zp, l0 + (HALO_CH + rd * NT) * 16);
      }
  #pragma unroll
      for (int rd = 0; rd < A_ROUNDS; ++rd) {
        const int pos = rd * NT + tid;
        if (pos < HALO_CH) {
          const int hp = pos >> 3, cpos = pos & 7;
          const int hy = hp / HW_, hx = hp - hy * HW_;
          const int chunk = (((cpos >> 1) ^ f4(hx)) << 1) | (cpos & 1);
          int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
          if constexpr (REFLECT) {
            ih = JG_REFLECT1(ih, p.H);
            iw = JG_REFLECT1(iw, p.W);
          }
          const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          const int rel = UP ? (((ih >> 1) - (oh0 >> 1)) * (p.W >> 1) + ((iw >> 1) - (ow0 >> 1))) : ((ih - oh0) * p.W + (iw - ow0));
          glds16(ok ? xb + rel * (int)p.ldx + chunk * 8 : zp, l0 + rd * NT * 16);
        }
      }
    
    } else {
      const int tx = t % tw;
      const int r2 = t / tw;
      const int ty = r2 % th;
      const int b = r2 / th;
      const int oh0 = ty * TH, ow0 = tx << 4;
      const long pix = ((long)b * p.H + oh0) * p.W + ow0;
      const T* xb = UP ? xg + (((long)b * (p.H >> 1) + (oh0 >> 1)) * (p.W >> 1) + (ow0 >> 1)) * p.ldx : xg + pix * p.ldx;
      const T* db = dyg + pix * p.lddy;
      const unsigned l0 = lds0 + (buf * BUF_CH + wave * 64) * 16;
  #pragma unroll
      for (int rd = 0; rd < D_ROUNDS; ++rd) glds16(d_rel[rd] >= 0 ? db + d_rel[rd] : zp, l0 + (HALO_CH + rd * NT) * 16);
  #pragma unroll
      for (int rd = 0; rd < A_ROUNDS; ++rd) {
        if (a_yx[rd] >= 0) {
          const int ih = oh0 - 1 + (a_yx[rd] >> 8), iw = ow0 - 1 + (a_yx[rd] & 255);
          const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          int rel = a_rel[rd];
          if constexpr (REFLECT) {
            if (!ok) rel += ((JG_REFLECT1(ih, p.H) - ih) * p.W + (JG_REFLECT1(iw, p.W) - iw)) * (int)p.ldx;
          }
          glds16((ok || REFLECT) ? xb + rel : zp, l0 + rd * NT * 16);
        }
      }
    
    }
  };
  // LDS-DMA instructions a wave issues per tile (wave-uniform): the count its vmcnt wait leaves in flight
  const bool partial = (A_ROUNDS > A_FULL) && (A_FULL * NT + wave * 64 < HALO_CH);

  // ---- fragment byte offsets inside a buffer ------------------------------------------------------------
  const int i16 = lane & 15, g = lane >> 4;
  const int trow = g >> 1;
  int abase[CPW][2], bbase[3][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int xq = (g & 1) * 8 + rd * 4 + (i16 >> 2);
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int cb = wm * CPW + i;
      abase[i][rd] = HALO_CH * 16 + (trow * 16 + xq) * DY_ROWB + ((cb ^ fdy<BCO>(xq)) << 5) + (i16 & 3) * 8;
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int hx = xq + s3;
      bbase[s3][rd] = (trow * HW_ + hx) * 128 + ((wn ^ f4(hx)) << 5) + (i16 & 3) * 8;
    }
  }
  auto tr_frag = [&](int off0, int off1) -> uint4 {
    const uint2 u0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smb + off0)));
    const uint2 u1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smb + off1)));
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
  };

#ifdef JG_WG_BLOAT
  asm volatile("" ::: "v255");     // experiment: the register footprint of the sliding-window kernel on this one
#endif
  f32x4 acc[9][CPW], accb[CPW];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int i = 0; i < CPW; ++i) acc[t9][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CPW; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.dbias != nullptr && ci0 == 0 && wn == 0;
  // eight 1.0 values: bf16 0x3F80, fp16 0x3C00
  const uint32_t one1 = to_bits<T>(from_f32<T>(1.0f));
  const uint32_t one2 = one1 | (one1 << 16);
  const uint4 ones = make_uint4(one2, one2, one2, one2);

  issue_tile(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = t + 1 < t1;
    if (more && !(dbg & 4)) issue_tile(t + 1, buf ^ 1);      // JG_HALO_DBG 4: timing without the per-tile LDS-DMA (first tile only)
    if (more && !(dbg & 4)) {
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
    if constexpr (PIPE) {
      constexpr int LA = CPW >= 4 ? 3 : 5;          // steps of lookahead: a step is CPW MFMAs (x 2 waves per SIMD) = 64 .. 128 cycles
      constexpr int NSTEP = NSUB * 9;
      auto load_fb = [&](int step) -> uint4 {
        const int sub = step / 9, t9 = step % 9, r = t9 / 3, s3 = t9 % 3;
        return tr_frag(boff + bbase[s3][0] + (2 * sub + r) * (HW_ * 128), boff + bbase[s3][1] + (2 * sub + r) * (HW_ * 128));
      };
      uint4 ring[LA + 1], fa[2][CPW];
#pragma unroll
      for (int i = 0; i < CPW; ++i) fa[0][i] = tr_frag(boff + abase[i][0], boff + abase[i][1]);
#pragma unroll
      for (int s = 0; s < LA; ++s) ring[s] = load_fb(s);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        const int sub = step / 9, t9 = step % 9;
        if (step + LA < NSTEP) ring[(step + LA) % (LA + 1)] = load_fb(step + LA);
        if (t9 == 1 && sub + 1 < NSUB) {       // dy fragments of the next sub-step: 8 steps ahead of their first use
#pragma unroll
          for (int i = 0; i < CPW; ++i)
            fa[(sub + 1) & 1][i] = tr_frag(boff + abase[i][0] + (sub + 1) * 32 * DY_ROWB, boff + abase[i][1] + (sub + 1) * 32 * DY_ROWB);
        }
        if (t9 == 0 && do_bias) {
#pragma unroll
          for (int i = 0; i < CPW; ++i) jg_mfma_pinned<T>(accb[i], fa[sub & 1][i], ones);
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) jg_mfma_pinned<T>(acc[t9][i], fa[sub & 1][i], ring[step % (LA + 1)]);
      }
      asm volatile("s_nop 7" ::: "memory");     // the compiler does not see the MFMAs: keep its next VALU write off their operands
    } else
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      uint4 fa[CPW];
#pragma unroll
      for (int i = 0; i < CPW; ++i)
        fa[i] = tr_frag(boff + abase[i][0] + sub * 32 * DY_ROWB, boff + abase[i][1] + sub * 32 * DY_ROWB);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) accb[i] = Mfma<T>::run(fa[i], ones, accb[i]);
      }
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int r = t9 / 3, s3 = t9 % 3;
        const uint4 fb = tr_frag(boff + bbase[s3][0] + (2 * sub + r) * (HW_ * 128), boff + bbase[s3][1] + (2 * sub + r) * (HW_ * 128));
#pragma unroll
        for (int i = 0; i < CPW; ++i) acc[t9][i] = Mfma<T>::run(fa[i], fb, acc[t9][i]);
      }
    }
    if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: D row = co = g*4 + q, col = ci = i16 ----------------------------------------------------
  if (dbg & 1) return;      // JG_HALO_DBG 1: timing without the atomic epilogue (tools/wgrad_pipe_ab.py --dbg)
  float* dw = (float*)p.dw;
  const int ci = ci0 + wn * 16 + i16;
  if (ci < p.Cin_out) {
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = co0 + (wm * CPW + i) * 16 + g * 4 + q;
          if (co < p.Cout_out) atomicAdd(dw + (long)co * p.lddw + (long)t9 * p.Cin_out + ci, p.alpha * acc[t9][i][q]);
        }
      }
    }
  }
  if (do_bias && i16 == 0) {
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + (wm * CPW + i) * 16 + g * 4 + q;
        if (co < p.Cout_out) atomicAdd(p.dbias + co, p.dbias_scale * accb[i][q]);
      }
  }
}

// Split-K choice: `slots` workgroups run at a time (1 or 2 per CU by LDS), so the launch runs in rounds; every block
// walks `per` tiles and pays a fixed prologue + atomic epilogue worth about OVH tiles.
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

template <typename T, int TH, int CPW, int WMR, int XMODE = 0>
void launch_wg(const WgP& p, hipStream_t st) {
  // JG_WGRAD_PIPE (1): the software-pipelined MFMA loop for the CPW == 2 tiles (the CPW == 4 tiles have no registers left for the ring)
  const bool pipe = CPW == 2 && jg_tune(JG_TUNE_WGRAD_PIPE) != 0;
  constexpr int BCO = WMR * CPW * 16;
  const int ncot = (p.Cout + BCO - 1) / BCO, npairs = ncot * (p.Cin / 64);
  const int ntiles = p.B * (p.H / TH) * (p.W >> 4);
  int per, splitk;
  pick_split(npairs, ntiles, TH == 16 ? 6 : 10, WMR == 1 && jg_tune(JG_TUNE_WGRAD_LDS_PAD) < 4096 ? 512 : 256, &per, &splitk);
  // JG_WGRAD_LDS_PAD: extra (unused) dynamic LDS bytes, an occupancy lever for A/B runs: e.g. the 4-wave configuration at one workgroup per
  // CU instead of two leaves half of every SIMD's register file to the kernels of the other stream
  const size_t pad = (size_t)jg_tune(JG_TUNE_WGRAD_LDS_PAD);
  if (pipe) hipLaunchKernelGGL((wgrad3x3_halo_kernel<T, TH, CPW, WMR, XMODE, CPW == 2>), dim3(npairs * splitk), dim3(WMR * 256), pad, st, p, ntiles, per, npairs, ncot, jg_tune(JG_TUNE_HALO_DBG));
  else hipLaunchKernelGGL((wgrad3x3_halo_kernel<T, TH, CPW, WMR, XMODE>), dim3(npairs * splitk), dim3(WMR * 256), pad, st, p, ntiles, per, npairs, ncot, jg_tune(JG_TUNE_HALO_DBG));
}

template <typename T>
void dispatch_wg(const WgP& p, hipStream_t st) {
  const int cfg = jg_tune(JG_TUNE_WGRAD_HALO_CFG);   // 0: auto; 1: 16x16 tiles x 64 co; 2: 8x16 tiles x 128 co; 3: 8x16 x 64 co, 4 waves; 4: 8x16 x 32 co, 4 waves; 6: sliding-window form (wgrad_sw.hip) for every launch
  // the 128-channel x 8-row configuration halves the L2->LDS bytes per MFMA but doubles the atomic
  // volume: it pays once a block has >= 64 (16-row) tiles to walk
  const long per1 = (long)p.B * (p.H >> 4) * (p.W >> 4) * (p.Cout / 64) * (p.Cin / 64) / 256;
  const bool big = cfg == 2 || (cfg == 0 && per1 >= 64);
  // mirrored borders (pad_mode 1) are compiled only into the 16x16 / 64-co configuration: the extra address arithmetic would push
  // the register-tight 8-row configurations into spilling
  if (p.Cout < 64) jg_note_kernel("wgrad3x3_halo_kernel<16 rows,<64 co>");     // its own row in the kernel tables: 8 of 64 channel rows live, HBM-bound
  else
  jg_note_kernel(p.reflect || !(big && p.Cout % 128 == 0) ? (cfg == 3 && !p.reflect && !p.x_up ? "wgrad3x3_halo_kernel<8 rows,64 co,4 waves>" :
                                                             (cfg == 4 || cfg == 5) && !p.reflect && !p.x_up ? "wgrad3x3_halo_kernel<8 rows,32 co,4 waves>" : "wgrad3x3_halo_kernel<16 rows,64 co>")
                                                          : "wgrad3x3_halo_kernel<8 rows,128 co>");
  // round 6: the sliding-window form (wgrad_sw.hip), OFF by default.  Alone it beats both tiles it could replace (10.94 vs 11.49 ms over
  // the 3x3 layers of the step); inside the step it loses: in place of the 16-row tile by 1.0 - 1.4 ms (243 VGPRs x 8 waves against 192: the
  // waves of the main stream find no room beside it -- the same +1.3 ms appear when the round-5 tile is merely COMPILED to 256 VGPRs), in
  // place of the 8-row x 128-co tile by 0.1 - 0.3 ms (the same footprint, but twice the L2 -> LDS bytes per MFMA).  profiles/r06_wgrad_sw_ab.txt.
  // JG_WGRAD_SW 1: in place of the 8-row x 128-co tile; 2: wherever the shape allows; JG_WGRAD_HALO_CFG 6: every launch (tests, tools).
  const int swm = jg_tune(JG_TUNE_WGRAD_SW);
  const bool bigt = big && p.Cout % 128 == 0;
  const bool sw = p.Cout >= 64 && (cfg == 6 || (cfg == 0 && ((swm == 1 && bigt && !p.reflect) || swm == 2)));
  if (sw) {
    jg_note_kernel("wgrad3x3_sw_kernel<16 rows,64 co>");
    jg_wgrad_sw_launch(std::is_same<T, bf16_t>::value ? JG_BF16 : JG_F16, p, st);
    return;
  }
  if (p.reflect) launch_wg<T, 16, 2, 2, 1>(p, st);
  else if (p.x_up && big && p.Cout % 128 == 0) launch_wg<T, 8, 4, 2, 2>(p, st);   // upsample-on-read: the two shapes the UNet up-blocks use
  else if (p.x_up) launch_wg<T, 16, 2, 2, 2>(p, st);
  else if (cfg == 3) launch_wg<T, 8, 4, 1>(p, st);   // 4 waves, 64 co x 8-row tiles, 2 workgroups / CU
  else if (cfg == 4 || (cfg == 5 && !(big && p.Cout % 128 == 0)))
    launch_wg<T, 8, 2, 1>(p, st);   // 4 waves, 32 co x 8-row tiles, pipelined loop, 2 INDEPENDENT workgroups / CU (round 5); 5 = instead of the 16-row tile only
  else if (big && p.Cout % 128 == 0) launch_wg<T, 8, 4, 2>(p, st);
  else launch_wg<T, 16, 2, 2>(p, st);
}

}  // namespace

bool jg_wgrad_halo_try(int dtype, const WgP& p, int nbatch, hipStream_t st, bool dry_run) {
  if (nbatch != 1 || p.R != 3 || p.S != 3 || p.pad != 1 || p.stride != 1 || p.out_mode != JG_OUT_ATOMIC_F32) return false;
  // Cout: a multiple of 64, or fewer than 64 (one partly filled channel tile: the 3 -> 8-channel head of the UNet at 256 x 256 x 32 images, whose
  // im2col weight gradient fetched the input once per tap pair: 582 us for 60 us of bytes)
  if (p.Cin % 64 || (p.Cout % 64 && (p.Cout > 64 || (p.Cout & 7))) || (p.H & 15) || (p.W & 15) || p.H != p.Ho || p.W != p.Wo) return false;
  if (p.Cout < 64 && ((long)p.B * p.H * p.W < 262144 || p.reflect || p.x_up)) return false;      // small launches stay on the im2col kernel
  if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.B * p.H * p.W * p.lddy >= (1L << 31)) return false;
  if (dry_run) return dtype == JG_F16 || dtype == JG_BF16;
  if (dtype == JG_F16) dispatch_wg<f16_t>(p, st);
  else if (dtype == JG_BF16) dispatch_wg<bf16_t>(p, st);
  else return false;
  return true;
}
