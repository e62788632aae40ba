// Implicit-GEMM convolution / batched "NT" GEMM on gfx950 MFMA (v_mfma_f32_16x16x32_{f16,bf16}).
//
//   y[m][n] = alpha * sum_k A[m][k] * Wt[n][k] + bias[n] + res_scale * res[m][n]
//   A[m][k] = x[b, oh*stride + r - pad, ow*stride + s - pad, ci]   m=(b,oh,ow)  k=(r,s,ci)
//
// Data layout: x NHWC (pixel stride ldx), weights KRSC ([Cout][R][S][Cin], row stride ldw),
// y NHWC.  Both operands are K-contiguous, so every MFMA fragment is one 16-byte read.
//
// Tiling: workgroup = 256 threads = 4 waves; block tile BM x BN x 32; each wave owns a 64x64
// sub-tile = 4x4 MFMA tiles of 16x16 (64 accumulator VGPRs).  Global->register->LDS staging with
// a register prefetch of the next K-step (loads issued before the MFMAs of the current step) and
// two LDS buffers: one barrier per K-step.  LDS rows are 64 B (32 k-values); the 16-byte chunk
// index is XOR-swizzled with bit 3 of the row so that each 16-lane service group of ds_read_b128
// covers all 64 banks (MI355X_MICROARCH.md, LDS table).
//
// The weight fragment is fed as the MFMA "A" operand and the activation fragment as "B", so that
// the 4 accumulator registers of a lane are 4 consecutive output channels of one pixel: the
// epilogue stores 8 bytes per lane per tile instead of four 2-byte stores.
#include "common.h"
#include "conv_params.h"
#include "conv_epilogue.h"
#include <cstdlib>
#include <type_traits>

namespace {


__device__ __forceinline__ int swz64(int row) { return ((row >> 3) & 1) << 1; }

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_nt_kernel(ConvP p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int A_CH = BM * 4 / 256, B_CH = BN * 4 / 256;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(A_CH >= 1 && B_CH >= 1, "tile too small");

  __shared__ uint4 sm[2][(BM + BN) * 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int tilesN = (p.N + BN - 1) / BN;
  const int n0 = (blockIdx.x % tilesN) * BN;
  const int m0 = (blockIdx.x / tilesN) * BM;

  const int z = blockIdx.z;
  const int zb = z / p.nh, zh = z % p.nh;
  const T* __restrict__ x = (const T*)p.x + zb * p.sxb + zh * p.sxh;
  const T* __restrict__ w = (const T*)p.w + zb * p.swb + zh * p.swh;

  // ---- per-thread staging coordinates ---------------------------------------------------
  const int kc = tid & 3;      // 16-byte chunk (8 k-values) inside the 32-wide K-step
  const int srow = tid >> 2;   // 0..63, +64 per extra chunk
  int ih0[A_CH], iw0[A_CH], pb[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int m = m0 + srow + 64 * i;
    const int ow = m % p.Wo;
    const int t = m / p.Wo;
    const int oh = t % p.Ho;
    const int b = t / p.Ho;
    ih0[i] = (m < p.M) ? oh * p.stride - p.pad : -(1 << 28);
    iw0[i] = ow * p.stride - p.pad;
    pb[i] = b * p.H;
  }
  // k-state of this thread's chunk: k = ks*32 + kc*8 -> (r, s, c)
  int c = (kc * 8) % p.Cin;
  int rs0 = (kc * 8) / p.Cin;
  int r = rs0 / p.S, s = rs0 % p.S;
  long kg = kc * 8;

  uint4 ra[A_CH], rb[B_CH];

  auto load_tiles = [&]() {
    const bool kvalid = r < p.R;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int ih = ih0[i] + r, iw = iw0[i] + s;
      const bool ok = kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const long off = ok ? ((long)(pb[i] + ih) * p.W + iw) * p.ldx + c : 0;
      ra[i] = ldg16(x + off, ok);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int n = n0 + srow + 64 * i;
      const bool ok = kvalid && n < p.N;
      rb[i] = ldg16(w + (ok ? (long)n * p.ldw + kg : 0), ok);
    }
  };
  auto advance_k = [&]() {
    kg += 32;
    c += 32;
    while (c >= p.Cin) {
      c -= p.Cin;
      if (++s == p.S) { s = 0; ++r; }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int row = srow + 64 * i;
      sm[buf][row * 4 + (kc ^ swz64(row))] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int row = srow + 64 * i;
      sm[buf][BM * 4 + row * 4 + (kc ^ swz64(row))] = rb[i];
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  auto compute = [&](int buf) {
    uint4 fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * WM + i * 16 + frow;
      fa[i] = sm[buf][row * 4 + (fk ^ swz64(row))];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * WN + j * 16 + frow;
      fb[j] = sm[buf][BM * 4 + row * 4 + (fk ^ swz64(row))];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i] = Mfma<T>::run(fb[j], fa[i], acc[j][i]);
  };

  const int nk = (p.K + 31) / 32;
  load_tiles();
  store_lds(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < nk;
    if (more) {
      advance_k();
      load_tiles();
    }
    compute(cur);
    if (more) store_lds(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------
  char* yb = p.y + (zb * p.syb + zh * p.syh) * (p.out_f32 ? 4 : 2);
  const T* resb = p.res ? (const T*)p.res + zb * p.srb + zh * p.srh : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
    if (n >= p.N) continue;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = p.bias[n + q];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WM + i * 16 + (lane & 15);
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = p.alpha * acc[j][i][q] + bv[q];
      if (resb) {
        const long mr = p.res_up ? jg_res_up_row(p, m) : (long)m;
        const uint2 rv = *reinterpret_cast<const uint2*>(resb + mr * p.ldres + n);
        float rf[4];
        unpack4<T>(rv, rf);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += p.res_scale * rf[q];
      }
      if (p.out_f32) {
        *reinterpret_cast<float4*>((float*)yb + (long)m * p.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        *reinterpret_cast<uint2*>((T*)yb + (long)m * p.ldy + n) = pack4<T>(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Variant 2: direct global->LDS staging (global_load_lds_dwordx4, the gfx950 LDS-DMA path).
// No VGPR round trip and no ds_write pass: on the register-staged kernel above the ds_write_b128
// traffic (~79 B/clk/CU) costs more LDS time than the MFMAs take.  An LDS-DMA instruction writes
// wave-base + lane*16, i.e. LDS stays LINEAR in issue order; the bank swizzle therefore goes on the
// SOURCE side: the lane that lands on physical chunk position c of row r fetches the logical
// k-chunk c ^ swz(r) (cdna_hip_programming.md 5.4 rule 21).  Zero padding / ragged edges: invalid
// lanes fetch from a 16-byte device-global zero page (always a valid address).
// BK = 32 (64-byte rows, swz64) or 64 (128-byte rows, swz128).
__device__ uint4 jg_zero_page = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int swz128r(int row) { return (row >> 1) & 7; }

// LDS-DMA from inline asm (as conv_halo.hip): hipcc neither drains nor counts it, which is what lets the ring form below keep NST - 1
// stages in flight across its one raw s_barrier per K step; every wait on these loads is an explicit wait_vmcnt_nt<N>().
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt_nt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NST = 2: double buffer, one drain + __syncthreads per K step (many co-resident workgroups hide the L2 -> LDS latency between them).
// NST > 2 (round 5): a ring of NST stages with NST - 1 K steps in flight per workgroup and a counted wait -- for the launches whose grid
// does NOT oversubscribe the CUs (token GEMMs of the ViT / SegFormer blocks, 1x1 layers at 32 x 32), where a K step of 8 - 32 MFMAs per
// wave is an order of magnitude shorter than the DMA latency it used to wait out.  Same arithmetic, same accumulation order.
template <typename T, int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool SPLIT = false, int NST = 2>
__global__ __launch_bounds__(256) void conv_nt_glds_kernel(ConvP p) {
  constexpr int CPR = BK / 8;                         // 16-byte chunks per LDS row
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int A_CH = BM * CPR / 256, B_CH = BN * CPR / 256;
  constexpr int RSTEP = 256 / CPR;                    // rows covered by one block-wide staging round
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(A_CH >= 1 && B_CH >= 1, "tile too small");

  __shared__ uint4 sm[NST][(BM + BN) * CPR];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int tilesN = (p.N + BN - 1) / BN;
  // XCD-aware tile order (round 6; conv_halo.hip has had it since round 2): workgroup b runs on XCD b % 8, so with the plain order the tilesN
  // column tiles of one row block -- which read the SAME rows of x -- sat in different L2s and x crossed HBM tilesN times (PMC: 1.49 x the
  // algorithmic bytes on the 128 x 128 tile).  Each XCD now owns a contiguous run of tile ids: neighbours in n (and, for R > 1, in m) share an L2.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  }
  const int n0 = (bid % tilesN) * BN;
  const int m0 = (bid / tilesN) * BM;

  const int z = blockIdx.z;
  const int zb = z / p.nh, zh = z % p.nh;
  const T* __restrict__ x = (const T*)p.x + zb * p.sxb + zh * p.sxh;
  const T* __restrict__ w = (const T*)p.w + zb * p.swb + zh * p.swh;

  const int cpos = tid % CPR;                         // physical chunk position inside the LDS row
  const int srow = tid / CPR;                         // + RSTEP per staging round
  const int sw = (BK == 32) ? swz64(srow) : swz128r(srow);   // invariant under + RSTEP
  const int kc = cpos ^ sw;                           // logical k-chunk this thread fetches
  int ih0[A_CH], iw0[A_CH], pb[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int m = m0 + srow + RSTEP * i;
    const int ow = m % p.Wo;
    const int t = m / p.Wo;
    const int oh = t % p.Ho;
    const int b = t / p.Ho;
    ih0[i] = (m < p.M) ? oh * p.stride - p.pad : -(1 << 28);
    iw0[i] = ow * p.stride - p.pad;
    pb[i] = b * p.H;
  }
  // split-K: blockIdx.y owns K-steps [ks0, ks1)
  const int nk_all = (p.K + BK - 1) / BK;
  const int kper = SPLIT ? (nk_all + p.splitk - 1) / p.splitk : nk_all;
  const int ks0 = SPLIT ? blockIdx.y * kper : 0;
  const int nk = min(nk_all, ks0 + kper) - ks0;       // <= 0: this slice stores zeros
  long kg = (long)ks0 * BK + kc * 8;
  int c = (int)(kg % p.Cin);
  int rs0 = (int)(kg / p.Cin);
  int r = rs0 / p.S, s = rs0 % p.S;
  const T* zp = reinterpret_cast<const T*>(&jg_zero_page);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_loads = [&](int buf) {
    const bool kvalid = r < p.R;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int ih = ih0[i] + r, iw = iw0[i] + s;
      const bool ok = kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const T* src = ok ? x + (((long)(pb[i] + ih) * p.W + iw) * p.ldx + c) : zp;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&sm[buf][256 * i + wave * 64], 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int n = n0 + srow + RSTEP * i;
      const bool ok = kvalid && n < p.N;
      const T* src = ok ? w + ((long)n * p.ldw + kg) : zp;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&sm[buf][BM * CPR + 256 * i + wave * 64], 16, 0, 0);
    }
  };
  auto advance_k = [&]() {
    kg += BK;
    c += BK;
    while (c >= p.Cin) {
      c -= p.Cin;
      if (++s == p.S) { s = 0; ++r; }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  auto compute = [&](int buf) {
#pragma unroll
    for (int sub = 0; sub < BK / 32; ++sub) {
      uint4 fa[TM], fb[TN];
      const int lk = fk + 4 * sub;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 16 + frow;
        fa[i] = sm[buf][row * CPR + (lk ^ ((BK == 32) ? swz64(row) : swz128r(row)))];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WN + j * 16 + frow;
        fb[j] = sm[buf][BM * CPR + row * CPR + (lk ^ ((BK == 32) ? swz64(row) : swz128r(row)))];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = Mfma<T>::run(fb[j], fa[i], acc[j][i]);
    }
  };

  if constexpr (NST > 2) {
    constexpr int LPS = A_CH + B_CH;                  // LDS-DMA instructions per thread and stage (dead stages fetch the zero page: the count stays uniform)
    typedef __attribute__((address_space(3))) char* lds_cptr;
    const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0][0];
    auto issue_ring = [&](int buf, bool live) {
      const bool kvalid = live && r < p.R;
      const unsigned dst = lds0 + (unsigned)(buf * (BM + BN) * CPR + wave * 64) * 16u;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int ih = ih0[i] + r, iw = iw0[i] + s;
        const bool ok = kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        glds16_nt(ok ? x + (((long)(pb[i] + ih) * p.W + iw) * p.ldx + c) : zp, dst + 256 * 16 * i);
      }
#pragma unroll
      for (int i = 0; i < B_CH; ++i) {
        const int n = n0 + srow + RSTEP * i;
        glds16_nt((kvalid && n < p.N) ? w + ((long)n * p.ldw + kg) : zp, dst + (BM * CPR + 256 * i) * 16);
      }
      if (live) advance_k();
    };
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue_ring(st, st < nk);
    int cur = 0, nxt = NST - 1;
    for (int ks = 0; ks < nk; ++ks) {
      wait_vmcnt_nt<(NST - 2) * LPS>();               // stage ks has landed (this thread's share) ...
      __builtin_amdgcn_s_barrier();                   // ... everybody's has, and everybody is done reading stage ks - 1
      issue_ring(nxt, ks + NST - 1 < nk);             // = the buffer of stage ks - 1
      compute(cur);
      cur = cur + 1 == NST ? 0 : cur + 1;
      nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    wait_vmcnt_nt<0>();                               // the epilogue below may reuse the ring as scratch
    __builtin_amdgcn_s_barrier();
  } else {
  if (nk > 0) issue_loads(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      advance_k();
      issue_loads(cur ^ 1);
    }
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  }

  // split-K: this K slice's raw fp32 partial tile goes to the workspace through the plain fp32 store path below (no alpha / bias /
  // residual / statistics: jg_splitk_finalize applies them once to the ordered sum).  SPLIT is a template flag: as a run-time condition
  // (even a wave-uniform one) it cost the unsplit kernel 64 more VGPRs (264: one wave per SIMD instead of two) and 60 % of its speed.
  constexpr bool split = SPLIT;
  const bool of32 = p.out_f32 || split;
  if constexpr (sizeof(sm) >= 4 * 16384 && TN == 4 && TM == 4) {
    if (!of32 && (p.N & 7) == 0) {
      // LDS-transposed epilogue with full-line residual reads / stores (conv_epilogue.h); all waves are past the
      // last barrier of the K loop, each uses a private 16 KB scratch
      ConvP q = p;
      q.y = p.y + (zb * p.syb + zh * p.syh) * 2;
      q.res = p.res ? p.res + (zb * p.srb + zh * p.srh) * 2 : nullptr;
      const long srow = p.stats ? ((long)(m0 / (p.Ho * p.Wo)) * p.nslots + (m0 / BM) % p.nslots) * p.ldstats : 0;
      const int mw = m0 + wm * WM;
      jg_epilogue_lds<T, TM>(
          q, acc, reinterpret_cast<char*>(&sm[0][0]) + wave * 16384, lane, n0 + wn * WN, p.stats ? m0 / (p.Ho * p.Wo) : 0,
          [&](int lp) -> long { return (mw + lp < p.M) ? (long)(mw + lp) : -1L; },
          [&](int, long m) -> long { return p.res_up ? jg_res_up_row(p, m) : m; },
          [&](int, int, int) -> long { return 0L; },   // y_pool is refused before this kernel is reached
          [&](int nch, const float* s1, const float* s2) {
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
              atomicAdd(p.stats + (srow + nch + qq) * 2, s1[qq]);
              atomicAdd(p.stats + (srow + nch + qq) * 2 + 1, s2[qq]);
            }
          });
      return;
    }
  }
  char* yb = split ? (char*)(p.ws + ((long)blockIdx.y * gridDim.z + z) * p.M * p.N) : p.y + (zb * p.syb + zh * p.syh) * (p.out_f32 ? 4 : 2);
  const long ldy = split ? (long)p.N : p.ldy;
  const float alpha = split ? 1.f : p.alpha;
  const float* bias = split ? nullptr : p.bias;
  float* stats = split ? nullptr : p.stats;
  const T* resb = (p.res && !split) ? (const T*)p.res + zb * p.srb + zh * p.srh : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
    if (n >= p.N) continue;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = bias[n + q];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WM + i * 16 + (lane & 15);
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = alpha * acc[j][i][q] + bv[q];
      if (resb) {
        const long mr = p.res_up ? jg_res_up_row(p, m) : (long)m;
        const uint2 rv = *reinterpret_cast<const uint2*>(resb + mr * p.ldres + n);
        float rf[4];
        unpack4<T>(rv, rf);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += p.res_scale * rf[q];
      }
      if (of32) {
        *reinterpret_cast<float4*>((float*)yb + (long)m * ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        *reinterpret_cast<uint2*>((T*)yb + (long)m * ldy + n) = pack4<T>(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1[q] += v[q];
        s2[q] += v[q] * v[q];
      }
    }
    // host guarantees (Ho*Wo) % BM == 0 when stats != nullptr: the whole tile belongs to one image
    if (stats) jg_stats_flush(stats, ((long)(m0 / (p.Ho * p.Wo)) * p.nslots + (m0 / BM) % p.nslots) * p.ldstats, n, s1, s2, lane);
  }
}

// y = alpha * sum over the K slices (in slice order) + bias + res_scale * res, one thread per 4 output channels
template <typename T>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(ConvP p, int nbatch) {
  const long n4 = p.N >> 2;
  const long total = (long)nbatch * p.M * n4;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % n4) * 4;
  const long zm = idx / n4;
  const long m = zm % p.M;
  const int z = (int)(zm / p.M);
  const int zb = z / p.nh, zh = z % p.nh;
  const long slice = (long)nbatch * p.M * p.N;
  const float* src = p.ws + (long)z * p.M * p.N + m * p.N + n;
  float4 a = *reinterpret_cast<const float4*>(src);
  for (int sp = 1; sp < p.splitk; ++sp) {
    const float4 b = *reinterpret_cast<const float4*>(src + sp * slice);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float v[4] = {p.alpha * a.x, p.alpha * a.y, p.alpha * a.z, p.alpha * a.w};
  if (p.bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += p.bias[n + q];
  }
  if (p.res) {
    const T* resb = (const T*)p.res + zb * p.srb + zh * p.srh;
    float rf[4];
    unpack4<T>(*reinterpret_cast<const uint2*>(resb + m * p.ldres + n), rf);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += p.res_scale * rf[q];
  }
  char* yb = p.y + (zb * p.syb + zh * p.syh) * (p.out_f32 ? 4 : 2);
  if (p.out_f32) *reinterpret_cast<float4*>((float*)yb + m * p.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
  else *reinterpret_cast<uint2*>((T*)yb + m * p.ldy + n) = pack4<T>(v[0], v[1], v[2], v[3]);
}

// K slices for a launch of `blocks` workgroups over `nk` K-steps: fill ~256 CUs, keep >= 4 K-steps per slice
static int pick_conv_splitk(long blocks, int nk, long out_elems, long ws_bytes) {
  if (blocks >= 128 || nk < 16) return 1;
  int sk = (int)((256 + blocks - 1) / blocks);
  if (sk > nk / 4) sk = nk / 4;
  if (sk > 32) sk = 32;
  while (sk > 1 && (long)sk * out_elems * 4 > ws_bytes) --sk;
  return sk < 2 ? 1 : sk;
}

template <typename T, int BM, int BN, int BK, int WMv, int WNv>
void launch_glds(ConvP p, int nbatch, hipStream_t st, long ws_bytes = 0) {
  const long blocks = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * nbatch;
  p.splitk = 1;
  if (p.ws && !p.stats && !p.res_up && (p.N & 3) == 0 && jg_tune(JG_TUNE_CONV_SPLITK))
    p.splitk = pick_conv_splitk(blocks, (p.K + BK - 1) / BK, (long)nbatch * p.M * p.N, ws_bytes);
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), p.splitk, nbatch);
  // ring form (JG_CONV_RING: 0 off, 1 auto, 2 wherever it exists): 64 x 64 (4 stages, 64 KB) and 128 x 128 tiles (3 stages, 96 KB), BK 64,
  // unsplit, K loops of >= 8 steps, grids that fit the chip in one round at the ring's occupancy (two / one workgroup per CU) -- beyond
  // that the co-resident workgroups of the double-buffered form hide the latency as well and the ring's LDS footprint costs a round
  // (tools/ring_probe.py: 1536 -> 384 on 16 x 257 tokens 23.2 -> 18.1 us, 512 -> 256 4x4 on 16 x 32 x 32 147 -> 114 us; 384 -> 1536 on
  // the same tokens 16.2 -> 23.9 us if forced)
  constexpr bool has_ring = BK == 64 && BM == BN && (BM == 64 || BM == 128);
  const int ring = jg_tune(JG_TUNE_CONV_RING);
  const bool use_ring = has_ring && ring && p.splitk == 1 && (ring >= 2 || ((p.K + BK - 1) / BK >= 8 && blocks <= (BM == 64 ? 512 : 256)));
  if constexpr (has_ring) {
    if (use_ring) {
      hipLaunchKernelGGL((conv_nt_glds_kernel<T, BM, BN, BK, WMv, WNv, false, BM == 64 ? 4 : 3>), grid, dim3(256), 0, st, p);
      return;
    }
  }
  if (p.splitk > 1) hipLaunchKernelGGL((conv_nt_glds_kernel<T, BM, BN, BK, WMv, WNv, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv_nt_glds_kernel<T, BM, BN, BK, WMv, WNv, false>), grid, dim3(256), 0, st, p);
  if (p.splitk > 1) {
    jg_note_kernel(BM == 64 ? "conv_nt_glds_kernel<64,64,64,2,2>+splitK" : BN == 32 ? "conv_nt_glds_kernel<256,32,64,4,1>+splitK"
                   : BN == 64 ? "conv_nt_glds_kernel<256,64,64,4,1>+splitK" : "conv_nt_glds_kernel<128,128,64,2,2>+splitK");
    const long total = (long)nbatch * p.M * (p.N >> 2);
    hipLaunchKernelGGL((splitk_finalize_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, nbatch);
  }
}

template <typename T>
int launch_conv(const ConvP& p, int nbatch, hipStream_t st, long ws_bytes) {
  const int variant = jg_tune(JG_TUNE_CONV_VARIANT);  // 1: register-staged 32-deep; 2..5: LDS-DMA staged (128x128x64 = 3); 6: + halo-resident 3x3
  if (p.stats && variant < 2) return JG_ERR_UNSUPPORTED;
  // streaming (LDS-free) kernels for the HBM-bound shapes first: 1x1 at >= 64k pixels, the 8-channel 3x3 stem, 64 -> 64 3x3 at >= 1M pixels
  if (variant >= 6 && !p.reflect && jg_conv1x1_try(sizeof(T) == 2 && std::is_same<T, f16_t>::value ? JG_F16 : JG_BF16, p, nbatch, st)) {
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  if (variant >= 6 && jg_conv_p64_try(sizeof(T) == 2 && std::is_same<T, f16_t>::value ? JG_F16 : JG_BF16, p, nbatch, st)) {
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  if (variant >= 6 && jg_conv_halo_try(sizeof(T) == 2 && std::is_same<T, f16_t>::value ? JG_F16 : JG_BF16, p, nbatch, st)) {
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  if (variant >= 6 && jg_conv_kxk_try(sizeof(T) == 2 && std::is_same<T, f16_t>::value ? JG_F16 : JG_BF16, p, nbatch, st)) {
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  if (p.reflect || p.x_up || p.y_pool) return JG_ERR_UNSUPPORTED;   // mirrored borders / upsample-on-read / pooled stores exist only in the halo-resident kernel
  if (variant >= 2) {
    // small launches (SegFormer / EfficientNet linear layers, discriminator tails): the default tiles leave most CUs without a workgroup
    // -- 64 x 64 tiles quadruple the block count (their lower MFMA efficiency does not matter at these sizes); the K loop is cut on top
    // of that when a workspace came along
    const long b128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * nbatch;
    const long b256 = (long)((p.M + 255) / 256) * ((p.N + 63) / 64) * nbatch;
    // (a long K loop is better served by the default tile cut into K slices: 55 vs 113 us on 512 -> 256, 4x4, 16x16 x 16 images)
    const bool can_split = p.ws && !p.stats && !p.res_up && (p.K + 63) / 64 >= 16 && jg_tune(JG_TUNE_CONV_SPLITK);
    // stats_mode 1 (GroupNorm-backward reductions in the epilogue) exists only in the LDS-transposed epilogue of the 4 x 4-fragment wave
    // tiles: those launches keep the default tiles
    const bool alt_ok = (variant == 3 || variant >= 6) && jg_tune(JG_TUNE_CONV_SMALL_TILE) && !(p.stats && p.stats_mode == 1);
    // round 5: when 64 x 64 tiles alone fill the chip (>= 256 of them: a few thousand rows, e.g. the ViT projector's 4112-token GEMMs with
    // N = 384), they are preferred to cutting the K loop of 99 default tiles (1536 -> 384 on 16 x 257 tokens: the split form + its finalize
    // launch took 30 us)
    const long b64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64) * nbatch;
    const bool fill64 = jg_tune(JG_TUNE_CONV_SMALL_TILE) >= 1 && jg_tune(JG_TUNE_CONV_SMALL_TILE) != 2 && b64 >= 256;
    const bool small = alt_ok && (p.N <= 64 ? b256 : b128) < 160 && (!can_split || fill64);
    if (small) {
      jg_note_kernel("conv_nt_glds_kernel<64,64,64,2,2>");
      launch_glds<T, 64, 64, 64, 2, 2>(p, nbatch, st, ws_bytes);
    } else if (p.N <= 32 && alt_ok) {
      jg_note_kernel("conv_nt_glds_kernel<256,32,64,4,1>");      // <= 32 output channels (7x7 content / output heads): no half-empty 64-wide tile
      launch_glds<T, 256, 32, 64, 4, 1>(p, nbatch, st, ws_bytes);
    } else if (p.N <= 64) {
      if (variant == 3 || variant >= 6) launch_glds<T, 256, 64, 64, 4, 1>(p, nbatch, st, ws_bytes);
      else launch_glds<T, 256, 64, 32, 4, 1>(p, nbatch, st);
    } else {
      if (variant == 2) launch_glds<T, 128, 128, 32, 2, 2>(p, nbatch, st);
      else if (variant == 3 || variant >= 6) launch_glds<T, 128, 128, 64, 2, 2>(p, nbatch, st, ws_bytes);
      else if (variant == 4) launch_glds<T, 256, 128, 32, 2, 2>(p, nbatch, st);
      else launch_glds<T, 256, 128, 64, 2, 2>(p, nbatch, st);
    }
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  if (p.N <= 64) {
    constexpr int BM = 256, BN = 64;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), 1, nbatch);
    hipLaunchKernelGGL((conv_nt_kernel<T, BM, BN, 4, 1>), grid, dim3(256), 0, st, p);
  } else {
    constexpr int BM = 128, BN = 128;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), 1, nbatch);
    hipLaunchKernelGGL((conv_nt_kernel<T, BM, BN, 2, 2>), grid, dim3(256), 0, st, p);
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}

}  // namespace

static int conv_args_to_params(const jg_conv_args* a, ConvP& p);

extern "C" int jg_conv2d_nt(int dtype, const jg_conv_args* a, jg_stream_t stream) {
  jg_note_kernel("");      // the dispatch sites that record their instance overwrite it (jg_last_kernel)
  ConvP p;
  const int rc = conv_args_to_params(a, p);
  if (rc != JG_OK) return rc;
  if (p.stats && jg_tune(JG_TUNE_DETERMINISTIC) != 0) {
    // JG_DETERMINISTIC 1: the fused GroupNorm statistics are fp32 atomics from every tile (and ds_add_f32 chains inside a tile), summed in
    // whatever order the tiles retire.  The convolution runs without them and the ordered statistics pass (norm.hip) reads its output:
    // replica 0 of the caller's [B][slots][ldstats][2] rows (the others stay zero; jg_gn_coef_ld sums the replicas in order).
    if (p.stats_mode != 0) return JG_ERR_UNSUPPORTED;      // the GroupNorm-backward reductions of stats_mode 1 have no ordered form
    float* st = p.stats;
    p.stats = nullptr;
    int r2;
    JG_DISPATCH_DTYPE(dtype, r2 = launch_conv<T>(p, a->nbatch, (hipStream_t)stream, a->ws ? (long)a->ws_bytes : 0););
    if (r2 != JG_OK) return r2;
    return jg_gn_stats_ld(dtype, a->y, a->ldy, st, (int64_t)p.nslots * p.ldstats, a->B, a->Ho * a->Wo, a->Cout, stream);
  }
  JG_DISPATCH_DTYPE(dtype, return launch_conv<T>(p, a->nbatch, (hipStream_t)stream, a->ws ? (long)a->ws_bytes : 0););
}

// 1x1 convolution of x WITH the GroupNorm apply pass of x in the same launch: y = conv(x) as jg_conv2d_nt, and additionally
// y_norm[m][c] = act(ab[b][c][0] x[m][c] + ab[b][c][1]) (pixel stride ldyn) -- the two readers of a ResBlock input whose channel count
// changes (`skip_connection(x)` and `act(norm(x))`, unet_generator_attn.py:233-266) share ONE pass over x.  Streaming kernel shapes only
// (1x1, stride 1, Cin % 32 == 0, Cin <= 256, Cout % 64 == 0, >= 65536 pixels in multiples of 16, whole images per 16-pixel tile row):
// JG_ERR_UNSUPPORTED otherwise, nothing launched.
extern "C" int jg_conv1x1_gn_apply(int dtype, const jg_conv_args* a, const float* ab, void* y_norm, int64_t ldyn, int act, jg_stream_t stream) {
  jg_note_kernel("");
  if (!ab || !y_norm || !a || ldyn < a->Cin || (ldyn % 8)) return JG_ERR_BAD_ARG;
  ConvP p;
  const int rc = conv_args_to_params(a, p);
  if (rc != JG_OK) return rc;
  if (((long)a->Ho * a->Wo) % 16 || a->nbatch != 1) return JG_ERR_UNSUPPORTED;
  p.aab = ab; p.ay = (char*)y_norm; p.lday = ldyn; p.aact = act;
  if (!jg_conv1x1_try(dtype, p, a->nbatch, (hipStream_t)stream)) return JG_ERR_UNSUPPORTED;
  JG_CHECK_LAUNCH();
  return JG_OK;
}

// Input gradient of a 1x1 (skip) convolution WITH the GroupNorm-backward apply step of the same tensor in its epilogue:
//   y[m][c] = alpha (dO . W^T)[m][c] + bias + du P + gx Q + R (+ scale1 add1 + scale2 add2),   du = gdy act'(a gx + b)
// with (a, b) = ab[b][c], (P, Q, R) = pqr[b][c] from jg_gn_coef / jg_gn_bwd_coef -- what jg_gn_bwd_apply_ld followed by jg_conv2d_nt(res =
// its result) compute in two passes (autograd fan-in of a ResBlock input that feeds both `in_layers` and `skip_connection`,
// unet_generator_attn.py:233-266).  gx / gdy / add1 / add2: [M][Cout] 16-bit with their own pixel strides.  Streaming-kernel shapes
// only (see jg_conv1x1_gn_apply): JG_ERR_UNSUPPORTED otherwise, nothing launched.
extern "C" int jg_conv1x1_gn_bwd_apply(int dtype, const jg_conv_args* a, const void* gn_x, int64_t ldgx, const void* gn_dy, int64_t ldgdy,
                                       const float* ab, const float* pqr, const void* add1, int64_t ldadd1, float scale1, const void* add2,
                                       int64_t ldadd2, float scale2, int act, jg_stream_t stream) {
  jg_note_kernel("");
  if (!a || !gn_x || !gn_dy || !ab || !pqr || ldgx < a->Cout || ldgdy < a->Cout || (ldgx % 8) || (ldgdy % 8)) return JG_ERR_BAD_ARG;
  if ((add1 && (ldadd1 < a->Cout || ldadd1 % 8)) || (add2 && (ldadd2 < a->Cout || ldadd2 % 8)) || a->res || a->bias) return JG_ERR_BAD_ARG;
  ConvP p;
  const int rc = conv_args_to_params(a, p);
  if (rc != JG_OK) return rc;
  if (((long)a->Ho * a->Wo) % 16 || a->nbatch != 1) return JG_ERR_UNSUPPORTED;
  p.bgx = (const char*)gn_x; p.bldgx = ldgx; p.bgdy = (const char*)gn_dy; p.bldgdy = ldgdy; p.bab = ab; p.bpqr = pqr;
  p.badd1 = (const char*)add1; p.bldadd1 = ldadd1; p.bsc1 = scale1; p.badd2 = (const char*)add2; p.bldadd2 = ldadd2; p.bsc2 = scale2; p.bact = act;
  if (!jg_conv1x1_try(dtype, p, a->nbatch, (hipStream_t)stream)) return JG_ERR_UNSUPPORTED;
  JG_CHECK_LAUNCH();
  return JG_OK;
}

static int conv_args_to_params(const jg_conv_args* a, ConvP& p) {
  if (!a || !a->x || !a->w || !a->y) return JG_ERR_BAD_ARG;
  if (a->Cin % 8 || a->Cout % 4 || a->ldx % 8 || a->ldw % 8 || a->ldy % 4) return JG_ERR_BAD_ARG;
  if (a->res && (a->ldres % 4)) return JG_ERR_BAD_ARG;
  if (a->nbatch < 1 || a->nh < 1 || a->nbatch > 65535) return JG_ERR_BAD_ARG;
  if (a->R < 1 || a->S < 1 || a->stride < 1) return JG_ERR_BAD_ARG;
  const long M = (long)a->B * a->Ho * a->Wo;
  if (M <= 0 || M > (1L << 30)) return JG_ERR_BAD_ARG;
  p.x = (const char*)a->x; p.w = (const char*)a->w; p.y = (char*)a->y; p.bias = a->bias; p.res = (const char*)a->res;
  p.aab = nullptr; p.ay = nullptr; p.lday = 0; p.aact = 0;
  p.bgx = p.bgdy = p.badd1 = p.badd2 = nullptr; p.bab = p.bpqr = nullptr; p.bldgx = p.bldgdy = p.bldadd1 = p.bldadd2 = 0; p.bsc1 = p.bsc2 = 0.f; p.bact = 0;
  p.M = (int)M; p.N = a->Cout; p.K = a->R * a->S * a->Cin;
  p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.R = a->R; p.S = a->S; p.pad = a->pad; p.stride = a->stride;
  p.Ho = a->Ho; p.Wo = a->Wo;
  p.ldx = a->ldx; p.ldw = a->ldw; p.ldy = a->ldy; p.ldres = a->ldres;
  p.nh = a->nh;
  p.sxb = a->sxb; p.sxh = a->sxh; p.swb = a->swb; p.swh = a->swh; p.syb = a->syb; p.syh = a->syh;
  p.srb = a->srb; p.srh = a->srh;
  p.alpha = a->alpha; p.res_scale = a->res_scale; p.out_f32 = a->out_f32;
  p.B = a->B; p.stats = a->stats; p.ldstats = a->ldstats > 0 ? a->ldstats : a->Cout;
  p.nslots = a->stats_slots > 0 ? a->stats_slots : 1;
  p.reflect = a->pad_mode == 1;
  if (a->pad_mode != 0 && a->pad_mode != 1) return JG_ERR_BAD_ARG;
  p.y_pool = a->y_mode == 1;
  if (a->y_mode != 0 && a->y_mode != 1) return JG_ERR_BAD_ARG;
  if (p.y_pool && (a->bias || a->res || a->stats || a->out_f32 || (a->Ho & 1) || (a->Wo & 1) || a->nbatch != 1)) return JG_ERR_BAD_ARG;
  p.x_up = a->x_mode;      // 1: upsample-on-read, 2: sub-pixel form (w = folded weights of jg_subpixel_fold)
  if (a->x_mode < 0 || a->x_mode > 2) return JG_ERR_BAD_ARG;
  if (p.x_up && (p.reflect || (a->H & 1) || (a->W & 1) || a->nbatch != 1)) return JG_ERR_BAD_ARG;
  p.res_up = a->res_mode == 1;
  if (a->res_mode != 0 && a->res_mode != 1) return JG_ERR_BAD_ARG;
  if (p.res_up && (!a->res || a->nbatch != 1 || (a->Ho & 1) || (a->Wo & 1))) return JG_ERR_BAD_ARG;
  p.dbg = 0; p.stats_mode = a->stats_mode; p.gx = (const char*)a->gn_x; p.gldx = a->gn_ldx; p.gab = a->gn_ab; p.gact = a->gn_act;
  if (p.stats && p.stats_mode == 1 && (!p.gx || !p.gab || p.gldx < a->Cout || (a->Cout & 7))) return JG_ERR_BAD_ARG;
  if (p.stats) {
    // fused GroupNorm statistics: single conv, whole tiles inside one image, LDS-DMA kernels only
    const long hw = (long)a->Ho * a->Wo;
    if (a->nbatch != 1 || a->out_f32 || hw % 256 || a->Cout % 64) return JG_ERR_UNSUPPORTED;
  }
  p.ws = (float*)a->ws; p.splitk = 1;
  if (a->ws && (a->ws_bytes < 0 || ((uintptr_t)a->ws & 15))) return JG_ERR_BAD_ARG;
  return JG_OK;
}
