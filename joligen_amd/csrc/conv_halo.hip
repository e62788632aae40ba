// Halo-resident implicit-GEMM 3x3 convolution (stride 1, pad 1) on gfx950 MFMA.
//
//   y[b,oh,ow,n] = alpha * sum_{r,s,ci} x[b,oh+r-1,ow+s-1,ci] * w[n][r][s][ci] + bias[n] + res_scale*res
//
// A workgroup owns a 16x16 tile of output pixels of ONE image and BN output channels.  The K loop
// runs over 64-channel chunks; for each chunk the 18x18-pixel input HALO of the tile (324 pixels x
// 128 B = 40.5 KB) is brought into LDS ONCE by LDS-DMA and all 9 filter taps read their MFMA
// operand fragments from it at shifted pixel positions.  Compared with re-fetching a 256x64 im2col
// tile per tap (gemm_nt.hip) this divides the activation traffic L2->LDS by 9/1.27 = 7 and removes
// every per-tap bounds check / address computation from the loop: the padding is materialised once,
// when the halo is loaded (out-of-image pixels fetch a zero page).
//
// The weight tile of a (tap, chunk) K-step (BN rows x 128 B) is streamed through an NBBUF-deep LDS
// ring with counted `s_waitcnt vmcnt(N)` so that its loads stay in flight across the one raw
// `s_barrier` per K-step; the halo of the NEXT chunk is prefetched one LDS-DMA round per tap.
//
// LDS images are lane-linear (LDS-DMA writes wave_base + lane*16), so the bank swizzle lives on the
// SOURCE side: the 16-byte chunk landing at physical position c of a 128-byte row fetches logical
// k-chunk c ^ ((row >> 1) & 7), and fragment reads apply the same XOR (row = halo pixel index /
// weight row).  MFMA: v_mfma_f32_16x16x32, weights as operand A so that a lane's 4 accumulators are
// 4 consecutive output channels of one pixel (8-byte epilogue stores).
//
// Addressing modes (all decided in the prologue / epilogue, the K loop is the same code):
//   reflect  out-of-image halo pixels fetch the mirrored interior pixel (ReflectionPad2d(1) + pad-0 conv as one launch)
//   x_up 1   x is the half-resolution tensor: halo pixel (ih, iw) fetches (ih >> 1, iw >> 1)   -- conv over Upsample_nearest(x)
//   x_up 2   the same function in its sub-pixel form (PHASE instantiation below): 4 taps on folded weights per output phase
//   res_up   the residual is at half resolution and is read through the same index map in the epilogue
//   y_pool   the epilogue stores the 2x2 sum-pool of the tile (adjoint of x_up 1 when this launch computes an input gradient)
#include "conv_params.h"
#include "conv_epilogue.h"
#include "mfma_pipe.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ uint4 jg_halo_zero_page = {0u, 0u, 0u, 0u};

constexpr int HW_ = 18;                 // halo width / height
constexpr int HALO_PX = HW_ * HW_;      // 324
constexpr int HALO_CH = HALO_PX * 8;    // 16-byte chunks per halo buffer

// LDS-DMA issued from inline asm: hipcc does not model it, so it neither drains it with a
// vmcnt(0) before the next ds_read (what it does for __builtin_amdgcn_global_load_lds) nor counts
// it -- every wait on these loads is an explicit wait_vmcnt<N>() below.  M0 (the LDS destination
// base) is compiler-reserved: saved, written and restored inside the one statement
// (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PHASE (sub-pixel form of conv3x3 over a nearest-x2-upsampled input, x_mode 2): p.H / p.W are the LOW-resolution dimensions of x,
// y is [B, 2H, 2W, N].  A workgroup owns a 16x16 tile of low-resolution pixels, BN channels and ONE of the four output phases
// (py, px): output pixel (2h + py, 2w + px) = sum over the 2x2 taps (a, b) of  Wp[py][px][a][b] . x[h + a + py - 1][w + b + px - 1],
// i.e. halo offsets (a + py, b + px) of the same 18x18 halo tile, with the folded weights Wp = [4][N][2][2][Cin] (jg_subpixel_fold):
// 4 instead of 9 K-steps per chunk.
//
// PIPE: the K loop in its software-pipelined form (mfma_pipe.h): the fragment reads of sub-step s + 1 (and, across the per-step barrier,
// the pixel fragments of the next K-step -- the halo is stable for a whole chunk) are in flight under the MFMAs of sub-step s; only the
// four weight-fragment reads that follow a barrier are exposed.  Same LDS images, same accumulators, bit-identical results.
template <typename T, int BN, int NT, int WAVES_M, int WAVES_N, int NABUF, int NBBUF, int MINB, bool PHASE = false, int PIPE_MODE = 0>
__global__ __launch_bounds__(NT, MINB) void conv3x3_halo_kernel(ConvP p) {
  constexpr int NTAP = PHASE ? 4 : 9;
  constexpr bool PIPE = PIPE_MODE != 0;
  static_assert(!PIPE || !PHASE, "the pipelined K loop covers the 9-tap form");
  static_assert(!PHASE || NABUF == 1, "the phase form reloads its halo between chunks");
  static_assert(!PHASE || NBBUF - 1 < NTAP, "weight ring prologue");
  constexpr int NWAVES = NT / 64;
  static_assert(WAVES_M * WAVES_N == NWAVES, "wave grid");
  constexpr int TM = 16 / WAVES_M;                 // 16-pixel tile rows per wave
  constexpr int WN = BN / WAVES_N, TN = WN / 16;   // output channels per wave
  constexpr int A_ROUNDS = (HALO_CH + NT - 1) / NT;
  constexpr int B_ROUNDS = BN * 8 / NT;
  constexpr int B_BUF = BN * 8;
  static_assert(B_ROUNDS >= 1 && BN * 8 % NT == 0, "weight tile vs block size");
  static_assert(NABUF == 1 || A_ROUNDS <= 9, "halo prefetch is spread over the 9 taps");
  static_assert(NBBUF >= 2 && NBBUF <= 5, "weight ring depth");

  // PIPE: the block's BN bias values are parked behind the ring (the pipelined loop has no registers to carry them across)
  __shared__ uint4 sm[NABUF * HALO_CH + NBBUF * B_BUF + (PIPE ? BN / 4 : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- block -> (image, tile row, tile col, channel tile); XCD-aware: consecutive ids share an L2 ----
  const int nwg = gridDim.x;
  int id;
  {
    const int q = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  }
  int py = 0, px = 0;
  if constexpr (PHASE) {        // the four phases of a tile are neighbours in the id order: they share the halo in L2
    py = (id >> 1) & 1;
    px = id & 1;
    id >>= 2;
  }
  const int tilesN = (p.N + BN - 1) / BN;
  const int n0 = (id % tilesN) * BN;
  const int sp = id / tilesN;
  const int tw = p.W >> 4, th = p.H >> 4;
  const int ow0 = (sp % tw) << 4;
  const int oh0 = ((sp / tw) % th) << 4;
  const int b = sp / (tw * th);

  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ w = (const T*)p.w + (PHASE ? (long)(py * 2 + px) * p.N * p.ldw : 0L);
  const T* zp = reinterpret_cast<const T*>(&jg_halo_zero_page);
  typedef __attribute__((address_space(3))) char* lds_cptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0];   // LDS byte address of sm
  const char* smb = reinterpret_cast<const char*>(&sm[0]);

  // ---- per-thread LDS-DMA source offsets (elements), fixed for the whole kernel --------------------
  int aoff[A_ROUNDS];
#pragma unroll
  for (int rd = 0; rd < A_ROUNDS; ++rd) {
    const int pos = rd * NT + tid;
    const int hp = pos >> 3, cpos = pos & 7;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
    if (p.reflect) {
      ih = JG_REFLECT1(ih, p.H);
      iw = JG_REFLECT1(iw, p.W);
    }
    const bool ok = pos < HALO_CH && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    const int kc = cpos ^ ((hx >> 1) & 7);   // swizzle by the COLUMN inside the halo row: the same for every row
    // x_up: the halo pixel (ih, iw) of the (virtual) upsampled image lives at (ih >> 1, iw >> 1) of the half-resolution tensor
    const int ush = p.x_up ? 1 : 0;
    aoff[rd] = ok ? (int)((((long)b * (p.H >> ush) + (ih >> ush)) * (p.W >> ush) + (iw >> ush)) * p.ldx) + kc * 8 : -1;
  }
  int boff[B_ROUNDS];
#pragma unroll
  for (int rd = 0; rd < B_ROUNDS; ++rd) {
    const int row = (tid >> 3) + rd * (NT / 8);
    const int cpos = tid & 7;
    const int kc = cpos ^ ((row >> 1) & 7);
    const int n = n0 + row;
    boff[rd] = (n < p.N) ? (int)((long)n * p.ldw) + kc * 8 : -1;
  }

  auto issue_a_round = [&](int abuf, int cc, int rd) {
    const int pos = rd * NT + tid;
    if (pos < HALO_CH) {
      const T* src = aoff[rd] >= 0 ? x + aoff[rd] + cc * 64 : zp;
      glds16(src, lds0 + (abuf * HALO_CH + rd * NT + wave * 64) * 16);
    }
  };
  auto issue_b = [&](int bbuf, int koff) {
#pragma unroll
    for (int rd = 0; rd < B_ROUNDS; ++rd) {
      const T* src = boff[rd] >= 0 ? w + boff[rd] + koff : zp;
      glds16(src, lds0 + (NABUF * HALO_CH + bbuf * B_BUF + rd * NT + wave * 64) * 16);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lk = lane >> 4;
  // weight fragment byte offsets inside a ring slot, per N-tile (k-half 1 = this ^ 64)
  int bfrag[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WN + j * 16 + l15;
    bfrag[j] = row * 128 + ((lk ^ ((row >> 1) & 7)) << 4);
  }
  // halo fragment byte offsets for the three horizontal tap shifts s (tile row 0 of this wave, k-half 0);
  // tile row i and vertical shift r add the constant (i + r) * 18 * 128, k-half 1 is ^ 64
  int afrag[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int hx = jg_pixperm(l15) + s3;   // MFMA row l15 <-> pixel column jg_pixperm(l15) (conv_epilogue.h)
    afrag[s3] = ((wm * TM) * HW_ + hx) * 128 + ((lk ^ ((hx >> 1) & 7)) << 4);
  }

  auto compute = [&](int abyte, int bbyte, int r, int s3) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint4 fa[TM], fb[TN];
      const int af = PHASE ? (s3 == 0 ? afrag[0] : (s3 == 1 ? afrag[1] : afrag[2])) : afrag[s3];   // PHASE: s3 is a run-time value
      const int a0 = (af ^ (sub * 64)) + abyte;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(smb + a0 + (i + r) * (HW_ * 128));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(smb + (bfrag[j] ^ (sub * 64)) + bbyte);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = Mfma<T>::run(fb[j], fa[i], acc[j][i]);
    }
  };

  const int nch = p.Cin >> 6;       // 64-channel chunks
  const int nk = nch * NTAP;
  static_assert(TN == 4, "one 64-channel epilogue pass per wave");
  float bias_pre[8];                // epilogue bias of this lane, in flight during the K loop
  float* sbias = reinterpret_cast<float*>(&sm[NABUF * HALO_CH + NBBUF * B_BUF]);
  if constexpr (PIPE) {
    for (int i = tid; i < BN; i += NT) sbias[i] = (p.bias && n0 + i < p.N) ? p.bias[n0 + i] : 0.f;   // visible after the prologue barrier
  } else {
    jg_epilogue_bias(p, lane, n0 + wn * WN, bias_pre);
  }

  // ---- prologue: halo of chunk 0, first NBBUF-1 weight tiles -----------------------------------------
  if (!(p.dbg & 4)) {
#pragma unroll
    for (int rd = 0; rd < A_ROUNDS; ++rd) issue_a_round(0, 0, rd);
  }
#pragma unroll
  for (int pb = 0; pb < NBBUF - 1; ++pb) issue_b(pb, pb * p.Cin);   // taps 0 .. NBBUF-2 of chunk 0 (nk >= 9 always)
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  int kpre = NBBUF - 1;             // K-step whose weight tile is issued next
  int pre_tap = NBBUF - 1, pre_cc = 0;
  int bslot = 0;                    // ring slot of the current K-step
  if constexpr (PIPE) {
    static_assert(TM == 8 && TN == 4 && NTAP == 9, "the pipelined K loop is written for 128-pixel x 64-channel wave tiles");
    static_assert(NABUF == 1 || A_ROUNDS <= 7, "the last halo piece must have been waited for (vmcnt(0) of tap >= A_ROUNDS) before tap 8 reads the buffer");
    constexpr int RS = HW_ * 128;   // one halo row down
    u32x4 faP[8], faQ[8], fb[5];     // weight fragments: k-half 0 (fbP) in fb[0..3]; k-half 1 (fbQ) re-uses fb[0], fb[1], fb[2] as they die, + fb[4]
    // LDS-side order of one K-step (P = k-half 0, Q = k-half 1; every group = 8 MFMAs of one weight fragment):
    //   [faP in flight from the previous step]  fbP[0..3] | G(P,0)+faQ[0..3] | G(P,1)+faQ[4..7] | G(P,2)+fbQ[0,1] | G(P,3)+fbQ[2,3]
    //   | G(Q,0)+faP'[0..3] | G(Q,1)+faP'[4..7] | G(Q,2) | G(Q,3) | vmcnt | barrier          (faP' = pixel fragments of the NEXT step)
    // LDS returns in order, so the lgkmcnt in front of each group is the number of reads issued after the fragment it needs.
    auto read_fa = [&](unsigned abyte) {      // pixel fragments of (tap 0, k-half 0)
      const unsigned a0 = lds0 + abyte + afrag[0];
      jg_rd4<0, RS, 2 * RS, 3 * RS>(faP[0], faP[1], faP[2], faP[3], a0);
      jg_rd4<4 * RS, 5 * RS, 6 * RS, 7 * RS>(faP[4], faP[5], faP[6], faP[7], a0);
    };
    auto kstep = [&](auto tapc, int cc, bool next_chunk) {
      constexpr int tap = decltype(tapc)::value;
      constexpr int r = tap / 3, s3 = tap % 3;
      constexpr int ntap = (tap + 1) % 9, nr = ntap / 3, ns3 = ntap % 3;
      const unsigned abyte = (NABUF == 2 ? (cc & 1) : 0) * (HALO_CH * 16);
      // halo buffer the NEXT step reads: the other one after tap 8 (it has landed: its last LDS-DMA round went out at tap <= 5 and
      // every wave has waited for a younger weight tile and passed two barriers since).  With a single halo buffer the fragments
      // prefetched at tap 8 are dropped and re-read after the reload (below).
      const unsigned nbyte = (NABUF == 2 && tap == 8) ? ((cc + 1) & 1) * (HALO_CH * 16) : abyte;
      const unsigned bbyte = (NABUF * HALO_CH + bslot * B_BUF) * 16;
      const unsigned b0 = lds0 + bbyte + bfrag[0], b1 = lds0 + bbyte + (bfrag[0] ^ 64);
      const unsigned a1 = lds0 + abyte + (afrag[s3] ^ 64), an = lds0 + nbyte + afrag[ns3];
      jg_rd4<0, 2048, 4096, 6144>(fb[0], fb[1], fb[2], fb[3], b0);
      // weight tile of K-step k + NBBUF - 1 into the slot freed at the previous barrier; one halo round of the next chunk.
      // (measured, profiles/r03_halo_pipe_ablation.txt: spreading the pieces between the MFMA groups, staggering them between the two waves
      // of a SIMD, or leaving the halo piece in flight across the barrier are all slower than issuing them right here)
      const bool b_iss = kpre < nk && !(p.dbg & 128);
      int slot = bslot + NBBUF - 1;
      if (slot >= NBBUF) slot -= NBBUF;
      const int koff = pre_tap * p.Cin + pre_cc * 64;
      if (kpre < nk) {
        ++kpre;
        if (++pre_tap == NTAP) { pre_tap = 0; ++pre_cc; }
      }
      const bool a_iss = NABUF == 2 && tap < A_ROUNDS && next_chunk && tap * NT + wave * 64 < HALO_CH && !(p.dbg & 4);
      if (b_iss) issue_b(slot, koff);
      if (a_iss) issue_a_round((cc + 1) & 1, cc + 1, tap);
      jg_g8r4<T, 3, (0 + r) * RS, (1 + r) * RS, (2 + r) * RS, (3 + r) * RS>(acc[0], fb[0], faP, faQ[0], faQ[1], faQ[2], faQ[3], a1);
      jg_g8r4<T, 6, (4 + r) * RS, (5 + r) * RS, (6 + r) * RS, (7 + r) * RS>(acc[1], fb[1], faP, faQ[4], faQ[5], faQ[6], faQ[7], a1);
      jg_g8r2<T, 9, 0, 2048>(acc[2], fb[2], faP, fb[0], fb[1], b1);
      jg_g8r2<T, 10, 4096, 6144>(acc[3], fb[3], faP, fb[2], fb[4], b1);
      jg_g8r4<T, 3, (0 + nr) * RS, (1 + nr) * RS, (2 + nr) * RS, (3 + nr) * RS>(acc[0], fb[0], faQ, faP[0], faP[1], faP[2], faP[3], an);
      jg_g8r4<T, 6, (4 + nr) * RS, (5 + nr) * RS, (6 + nr) * RS, (7 + nr) * RS>(acc[1], fb[1], faQ, faP[4], faP[5], faP[6], faP[7], an);
      jg_g8r0<T, 9>(acc[2], fb[2], faQ);
      jg_g8r0<T, 8>(acc[3], fb[4], faQ);
      if (NBBUF >= 3 && b_iss) {
        if (a_iss) wait_vmcnt<(NBBUF - 2) * B_ROUNDS + 1>(); else wait_vmcnt<(NBBUF - 2) * B_ROUNDS>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (++bslot == NBBUF) bslot = 0;
    };
    read_fa(0);
    for (int cc = 0; cc < nch; ++cc) {
      const bool next_chunk = cc + 1 < nch;
      kstep(std::integral_constant<int, 0>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 1>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 2>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 3>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 4>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 5>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 6>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 7>{}, cc, next_chunk);
      kstep(std::integral_constant<int, 8>{}, cc, next_chunk);
      if (NABUF == 1 && next_chunk) {
#pragma unroll
        for (int rd = 0; rd < A_ROUNDS; ++rd) issue_a_round(0, cc + 1, rd);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_fa(0);
      }
    }
    // the last MFMAs are still in the pipe when the epilogue's first VALU reads the accumulators: the compiler does not see them
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  } else
  for (int cc = 0; cc < nch; ++cc) {
    const int abyte = (NABUF == 2 ? (cc & 1) : 0) * (HALO_CH * 16);
    const bool next_chunk = cc + 1 < nch;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      // 1. prefetch: weight tile of K-step k + NBBUF - 1 into the slot freed at the previous barrier
      const bool b_iss = kpre < nk;
      if (b_iss) {
        int slot = bslot + NBBUF - 1;
        if (slot >= NBBUF) slot -= NBBUF;
        issue_b(slot, pre_tap * p.Cin + pre_cc * 64);
        ++kpre;
        if (++pre_tap == NTAP) { pre_tap = 0; ++pre_cc; }
      }
      // 2. one LDS-DMA round of the next chunk's halo per tap
      bool a_iss = false;
      if (NABUF == 2 && tap < A_ROUNDS && next_chunk && tap * NT + wave * 64 < HALO_CH) {  // wave-uniform
        issue_a_round((cc + 1) & 1, cc + 1, tap);
        a_iss = true;
      }
      // 3. MFMAs of this K-step
      if (!(p.dbg & 2)) {
        if constexpr (PHASE) compute(abyte, (NABUF * HALO_CH + bslot * B_BUF) * 16, (tap >> 1) + py, (tap & 1) + px);
        else compute(abyte, (NABUF * HALO_CH + bslot * B_BUF) * 16, tap / 3, tap % 3);
      }
      // 4. the weight tile of the NEXT K-step (and, at tap 8, the whole next halo) must have landed;
      //    what was issued in this step may stay in flight (NBBUF == 3)
      if (NBBUF >= 3 && b_iss) {
        // in flight: the weight tiles of K-steps k+2 .. k+NBBUF-1 (+ the halo round issued in this step; halo rounds
        // of earlier steps are older than the tile that must land, so they are covered)
        if (a_iss) wait_vmcnt<(NBBUF - 2) * B_ROUNDS + 1>(); else wait_vmcnt<(NBBUF - 2) * B_ROUNDS>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (++bslot == NBBUF) bslot = 0;
    }
    if (NABUF == 1 && next_chunk) {
      // single halo buffer: reload between chunks (all waves are past their reads of it)
#pragma unroll
      for (int rd = 0; rd < A_ROUNDS; ++rd) issue_a_round(0, cc + 1, rd);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
  }

  if (p.dbg & 1) return;
  // ---- epilogue: LDS-transposed, full-line stores (conv_epilogue.h) ---------------------------------------
  static_assert(TN % 4 == 0, "the shared epilogue works on 64-channel wave tiles");
  static_assert(sizeof(sm) >= NWAVES * 16384 + BN * 8, "epilogue scratch");
  char* smc = reinterpret_cast<char*>(&sm[0]);          // every wave is past its last LDS read (K-loop barrier)
  float* sred = reinterpret_cast<float*>(smc + NWAVES * 16384);
  if (p.stats) {
    for (int i = tid; i < BN * 2; i += NT) sred[i] = 0.f;
    __syncthreads();
  }
  const long mrow0 = PHASE ? ((long)b * 2 * p.H + 2 * (oh0 + wm * TM) + py) * (2 * p.W) + 2 * ow0 + px
                           : ((long)b * p.H + oh0 + wm * TM) * p.W + ow0;
  // half-resolution residual (res_up): tile origins and the wave's row offset are even, so the nearest-upsample map is two shifts
  // (PHASE: the kernel's own grid IS the half resolution, the residual row of a pixel is its low-resolution index)
  const int wres = PHASE ? p.W : (p.res_up ? (p.W >> 1) : p.W), rsh = (!PHASE && p.res_up) ? 1 : 0;
  const long rrow0 = PHASE ? ((long)b * p.H + oh0 + wm * TM) * p.W + ow0
                           : (p.res_up ? ((long)b * (p.H >> 1) + ((oh0 + wm * TM) >> 1)) * wres + (ow0 >> 1) : mrow0);
  if constexpr (PIPE) {
#pragma unroll
    for (int q = 0; q < 8; ++q) bias_pre[q] = sbias[wn * WN + (lane & 7) * 8 + q];
  }
#pragma unroll
  for (int h = 0; h < TN / 4; ++h) {     // 64 output channels of the wave tile at a time
    jg_epilogue_lds<T, TM, true>(
        p, reinterpret_cast<f32x4(&)[4][TM]>(acc[4 * h]), smc + wave * 16384, lane, n0 + wn * WN + 64 * h, b,
        [&](int lp) -> long { return PHASE ? mrow0 + (long)(lp >> 4) * (4 * p.W) + 2 * (lp & 15) : mrow0 + (long)(lp >> 4) * p.W + (lp & 15); },
        [&](int lp, long m) -> long {
          if (PHASE && !p.res_up) return m;      // full-resolution residual: the output row itself
          return rrow0 + (long)((lp >> 4) >> rsh) * wres + ((lp & 15) >> rsh);
        },
        [&](int slab, int r2, int c2) -> long {
          return ((long)b * (p.H >> 1) + ((oh0 + wm * TM) >> 1) + slab * 2 + r2) * (p.W >> 1) + (ow0 >> 1) + c2;
        },
        [&](int nch, const float* s1, const float* s2) {
          // wave partials -> LDS (ds_add_f32) -> ONE global atomic pair per channel per block
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            atomicAdd(&sred[(nch - n0 + q) * 2], s1[q]);
            atomicAdd(&sred[(nch - n0 + q) * 2 + 1], s2[q]);
          }
        },
        PIPE ? bias_pre : nullptr);
  }
  if (p.stats && !(p.dbg & 64)) {
    __syncthreads();
    float* dst = p.stats + (((long)b * p.nslots + sp % p.nslots) * p.ldstats + n0) * 2;
    for (int i = tid; i < BN * 2; i += NT) atomicAdd(dst + i, sred[i]);
  }
}

template <typename T, int BN, int NT, int WMv, int WNv, int NABUF, int NBBUF, int MINB>
void launch_halo_phase(const ConvP& p, hipStream_t st) {   // p.H / p.W: low-resolution grid; 4 phases per tile
  const int tiles = p.B * (p.H >> 4) * (p.W >> 4) * ((p.N + BN - 1) / BN) * 4;
  hipLaunchKernelGGL((conv3x3_halo_kernel<T, BN, NT, WMv, WNv, NABUF, NBBUF, MINB, true>), dim3(tiles), dim3(NT), 0, st, p);
}

template <typename T, int BN, int NT, int WMv, int WNv, int NABUF, int NBBUF, int MINB, int PIPE = 0>
void launch_halo(const ConvP& p, hipStream_t st) {
  const int tiles = p.B * (p.H >> 4) * (p.W >> 4) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((conv3x3_halo_kernel<T, BN, NT, WMv, WNv, NABUF, NBBUF, MINB, false, PIPE>), dim3(tiles), dim3(NT), 0, st, p);
}

template <typename T>
void dispatch_halo(const ConvP& p, hipStream_t st) {
  // JG_HALO_CFG 0: auto; 1: never the 256-wide tile; 2: 8-wave 128-wide tile; 3: the 256-wide tile whenever N % 256 == 0 (parity tests
  // force the bench's configuration at test-sized grids); 4: 64-wide tile with a 4-deep weight ring
  const int cfg = jg_tune(JG_TUNE_HALO_CFG);
  // 256-wide tiles run one 8-wave workgroup per CU: worth it only when the grid fills whole rounds of 256
  const long b256 = (long)p.B * (p.H >> 4) * (p.W >> 4) * (p.N / 256);
  const bool fill256 = p.N % 256 == 0 && (double)b256 / (double)(((b256 + 255) / 256) * 256) >= 0.85;
  // (tried: <256, 256, 2, 2, 1, 2, 1> = 4 waves x (128 px x 128 ch) with the 256 accumulator registers in AGPRs -- halves the LDS
  //  fragment traffic per MFMA, but one wave per SIMD cannot hide the halo reloads: 1000-1170 vs 1260-1430 TFLOP/s, not kept)
  // JG_HALO_PIPE (1): software-pipelined K loop (mfma_pipe.h) for the configurations with 128-pixel x 64-channel wave tiles; 0 = the
  // compiler-scheduled loop (A/B, tools/halo_pipe_ab.py)
  const int pipe = jg_tune(JG_TUNE_HALO_PIPE);
  if ((fill256 && cfg == 0) || (cfg == 3 && p.N % 256 == 0)) {
    jg_note_kernel("conv3x3_halo_kernel<256-wide,8 waves>");
    if (pipe) launch_halo<T, 256, 512, 2, 4, 2, 2, 1, 1>(p, st); else launch_halo<T, 256, 512, 2, 4, 2, 2, 1>(p, st);
  }
  else if (p.N % 128 == 0 && cfg == 2) { jg_note_kernel("conv3x3_halo_kernel<128-wide,8 waves>"); launch_halo<T, 128, 512, 4, 2, 2, 3, 1>(p, st); }
  else if (p.N % 128 == 0) {   // 4 waves x (128 px x 64 ch), 2 workgroups / CU
    jg_note_kernel("conv3x3_halo_kernel<128-wide,4 waves>");
    if (pipe) launch_halo<T, 128, 256, 2, 2, 1, 2, 2, 1>(p, st); else launch_halo<T, 128, 256, 2, 2, 1, 2, 2>(p, st);
  }
  else if (cfg == 4) { jg_note_kernel("conv3x3_halo_kernel<64-wide>"); launch_halo<T, 64, 256, 4, 1, 1, 4, 2>(p, st); }
  else { jg_note_kernel("conv3x3_halo_kernel<64-wide>"); launch_halo<T, 64, 256, 4, 1, 1, 3, 2>(p, st); }
}

}  // namespace

namespace {

// 16-bit folded weights of the sub-pixel form: out[ph = py * 2 + px][co][a][b][ci] = sum of w32[co][r][s][ci] over r in R(py, a),
// s in R(px, b) with R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}; fp32 sum, one rounding
template <typename T>
__global__ __launch_bounds__(256) void subpixel_fold_kernel(const float* __restrict__ w32, T* __restrict__ out, int Cout, int Cin) {
  const long n = (long)Cout * Cin * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    long t = i / Cin;
    const int tap = (int)(t & 3);
    t >>= 2;
    const int co = (int)(t % Cout), ph = (int)(t / Cout);
    const int py = ph >> 1, px = ph & 1, a = tap >> 1, b = tap & 1;
    const int r0 = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), r1 = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int s0 = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), s1 = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float acc = 0.f;
    for (int r = r0; r <= r1; ++r)
      for (int sx = s0; sx <= s1; ++sx) acc += w32[(((long)co * 3 + r) * 3 + sx) * Cin + ci];
    out[i] = from_f32<T>(acc);
  }
}

// The same four-phase form for a TRANSPOSED convolution of stride 2 (the input gradient of a stride-2 convolution / nn.ConvTranspose2d):
// y[2h + py][2w + px][n] = sum over (a, b) of Wp[py][px][n][a][b] . x[h + a + py - 1][w + b + px - 1], where tap (a, b) of phase (py, px) is tap
// r = pad + 2 - py - 2a, s = pad + 2 - px - 2b of the strided convolution's kernel (zero outside 0 .. R-1: a 3x3 kernel fills 9 of the 16 slots, a
// 4x4 kernel all of them) -- no zero-dilated copy of x, a quarter (4x4) of the dilated form's MACs.  wT = the flipped / transposed working
// weights [N][R][S][C] (wT[n][R-1-r][S-1-s][c] = w[c][r][s][n]), 16-bit; out [4][N][2][2][C]: a rearrangement, no arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void transposed_fold_kernel(const T* __restrict__ wT, T* __restrict__ out, int N, int C, int R, int S, int pad) {
  const long n = (long)N * C * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long t = i / C;
    const int tap = (int)(t & 3);
    t >>= 2;
    const int nn = (int)(t % N), ph = (int)(t / N);
    const int py = ph >> 1, px = ph & 1, a = tap >> 1, b = tap & 1;
    const int r = pad + 2 - py - 2 * a, sx = pad + 2 - px - 2 * b;
    T v = from_f32<T>(0.f);
    if (r >= 0 && r < R && sx >= 0 && sx < S) v = wT[(((long)nn * R + (R - 1 - r)) * S + (S - 1 - sx)) * C + c];
    out[i] = v;
  }
}

}  // namespace

extern "C" int jg_transposed_fold(int dtype, const void* wT, void* out, int N, int C, int R, int S, int pad, jg_stream_t s) {
  if (!wT || !out || N < 1 || C < 1 || R < 1 || S < 1 || pad < 0) return JG_ERR_BAD_ARG;
  const long n = (long)N * C * 16;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transposed_fold_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)s, (const T*)wT, (T*)out, N, C, R, S, pad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_subpixel_fold(int dtype, const float* w32, void* out, int Cout, int Cin, jg_stream_t s) {
  if (!w32 || !out || Cout < 1 || Cin < 1) return JG_ERR_BAD_ARG;
  const long n = (long)Cout * Cin * 16;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((subpixel_fold_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)s, w32, (T*)out, Cout, Cin););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

bool jg_conv_halo_try(int dtype, const ConvP& p0, int nbatch, hipStream_t st) {
  ConvP p = p0;
  p.dbg = jg_tune(JG_TUNE_HALO_DBG);
  if (p.x_up == 2) {
    // sub-pixel form: p.H / p.W arrive as the upsampled (output) size; the kernel runs on the half-resolution grid
    if (nbatch != 1 || p.R != 3 || p.S != 3 || p.pad != 1 || p.stride != 1 || p.out_f32 || p.reflect || p.y_pool) return false;
    if (p.Cin % 64 || p.N % 64 || (p.H & 31) || (p.W & 31) || p.H != p.Ho || p.W != p.Wo || p.ldw != 4L * p.Cin) return false;
    if ((long)p.B * p.H * p.W * p.ldy >= (1L << 33) || (long)4 * p.N * p.ldw >= (1L << 31)) return false;
    if (p.stats && p.stats_mode != 0) return false;
    p.H >>= 1; p.W >>= 1; p.x_up = 0;
    if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31)) return false;
    jg_note_kernel("conv3x3_halo_kernel<subpixel>");
    if (dtype == JG_F16) {
      if (p.N % 128 == 0) launch_halo_phase<f16_t, 128, 256, 2, 2, 1, 2, 2>(p, st); else launch_halo_phase<f16_t, 64, 256, 4, 1, 1, 3, 2>(p, st);
    } else if (dtype == JG_BF16) {
      if (p.N % 128 == 0) launch_halo_phase<bf16_t, 128, 256, 2, 2, 1, 2, 2>(p, st); else launch_halo_phase<bf16_t, 64, 256, 4, 1, 1, 3, 2>(p, st);
    } else return false;
    return true;
  }
  if (nbatch != 1 || p.R != 3 || p.S != 3 || p.pad != 1 || p.stride != 1 || p.out_f32) return false;
  if (p.Cin % 64 || p.N % 64 || (p.H & 15) || (p.W & 15) || p.H != p.Ho || p.W != p.Wo) return false;
  if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return false;
  if (dtype == JG_F16) dispatch_halo<f16_t>(p, st);
  else if (dtype == JG_BF16) dispatch_halo<bf16_t>(p, st);
  else return false;
  return true;
}
