// Weight gradient / batched "TN" GEMM on gfx950 MFMA.
//
//   dw[co][(r,s,ci)] (+)= alpha * sum_p dy[p][co] * x[b, oh*stride+r-pad, ow*stride+s-pad, ci]
//   p = (b,oh,ow) is the reduction ("K") index: both operands are K-STRIDED in memory (NHWC),
//   while an MFMA fragment wants 8 consecutive k per lane.  Each thread therefore loads a
//   4-pixel x 8-channel block (4 x 16 B, channel-contiguous), transposes it in registers and
//   writes 8 x 8 B (4 pixels of one channel) into a [channel][pixel] LDS tile; fragments are
//   then single ds_read_b128 like in gemm_nt.hip.
//
// Tiling: 256 threads = 4 waves (2x2), block tile 128 (co) x 128 (r,s,ci) x 64 pixels per
// K-step, split-K over the pixel range (gridDim.z = nbatch * splitk), fp32 atomics into the
// gradient arena (or plain stores for the attention backward).  LDS rows are 128 B (64 pixels);
// 16-byte chunk index XOR-swizzled with (row >> 1) & 7: conflict-free for both the 16-lane
// ds_read_b128 groups and the 16-lane ds_write_b64 groups.
#include "common.h"
#include "wgrad_params.h"
#include <vector>
#include <cstdlib>

namespace {


__device__ __forceinline__ int swz128(int row) { return (row >> 1) & 7; }

// 4 pixels x 8 channels (ra[i] = 8 channels of pixel i) -> out[j] = 4 pixels of channel j
__device__ __forceinline__ void transpose4x8(const uint4* ra, uint2* out) {
  const uint32_t w[4][4] = {{ra[0].x, ra[0].y, ra[0].z, ra[0].w},
                            {ra[1].x, ra[1].y, ra[1].z, ra[1].w},
                            {ra[2].x, ra[2].y, ra[2].z, ra[2].w},
                            {ra[3].x, ra[3].y, ra[3].z, ra[3].w}};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    out[2 * d].x = (w[0][d] & 0xffffu) | (w[1][d] << 16);
    out[2 * d].y = (w[2][d] & 0xffffu) | (w[3][d] << 16);
    out[2 * d + 1].x = (w[0][d] >> 16) | (w[1][d] & 0xffff0000u);
    out[2 * d + 1].y = (w[2][d] >> 16) | (w[3][d] & 0xffff0000u);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(WgP p) {
  constexpr int BM = 128, BN = 128, BK = 64;
  __shared__ uint4 sm[2][(BM + BN) * 8];  // rows of 128 B = 8 chunks

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int tilesN = (p.Ktot + BN - 1) / BN;
  const int n0 = (blockIdx.x % tilesN) * BN;
  const int m0 = (blockIdx.x / tilesN) * BM;

  const int z = blockIdx.z;
  const int batch = z / p.splitk, split = z % p.splitk;
  const int zb = batch / p.nh, zh = batch % p.nh;
  const T* __restrict__ dy = (const T*)p.dy + zb * p.sdyb + zh * p.sdyh;
  const T* __restrict__ x = (const T*)p.x + zb * p.sxb + zh * p.sxh;

  int per = (p.Mpix + p.splitk - 1) / p.splitk;
  per = (per + BK - 1) / BK * BK;
  const int kbeg = split * per;
  const int kend = min(p.Mpix, kbeg + per);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + BK - 1) / BK;

  // staging coordinates: pixel quad pq (4 consecutive pixels), channel octet co
  const int pq = tid & 15, co = tid >> 4;
  const int cm = m0 + co * 8;                 // dy channel of this thread's octet
  const bool a_ok = cm < p.Cout;
  const int nn = n0 + co * 8;                 // (r,s,ci) column of this thread's octet
  const bool b_ok = nn < p.Ktot;
  const int rs = nn / p.Cin, ci = nn % p.Cin;
  const int fr = rs / p.S, fs = rs % p.S;
  const bool quad_row = (p.Wo & 3) == 0;      // a pixel quad never straddles an output row

  uint4 ra[4], rb[4];
  float bsum[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bsum[q] = 0.f;
  const bool do_bias = p.dbias != nullptr && (blockIdx.x % tilesN) == 0;

  auto load_tiles = [&](int kbase) {
    const int p0 = kbase + pq * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = p0 + i;
      const bool ok = a_ok && pp < kend;
      ra[i] = ldg16(dy + (ok ? (long)pp * p.lddy + cm : 0), ok);
    }
    if (quad_row) {
      const int ow0 = p0 % p.Wo;
      const int t = p0 / p.Wo;
      const int oh = t % p.Ho;
      const int b = t / p.Ho;
      const int ih = oh * p.stride + fr - p.pad;
      const bool row_ok = b_ok && (unsigned)ih < (unsigned)p.H;
      const long rowoff = ((long)(b * p.H + ih) * p.W) * p.ldx + ci;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iw = (ow0 + i) * p.stride + fs - p.pad;
        const bool ok = row_ok && (p0 + i) < kend && (unsigned)iw < (unsigned)p.W;
        rb[i] = ldg16(x + (ok ? rowoff + (long)iw * p.ldx : 0), ok);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = p0 + i;
        const int ow = pp % p.Wo;
        const int t = pp / p.Wo;
        const int oh = t % p.Ho;
        const int b = t / p.Ho;
        const int ih = oh * p.stride + fr - p.pad;
        const int iw = ow * p.stride + fs - p.pad;
        const bool ok = b_ok && pp < kend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        rb[i] = ldg16(x + (ok ? ((long)(b * p.H + ih) * p.W + iw) * p.ldx + ci : 0), ok);
      }
    }
  };
  auto store_lds = [&](int buf) {
    uint2 ta[8], tb[8];
    transpose4x8(ra, ta);
    transpose4x8(rb, tb);
    uint2* s2 = reinterpret_cast<uint2*>(&sm[buf][0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = co * 8 + j;
      const int slot = row * 16 + (((pq >> 1) ^ swz128(row)) << 1) + (pq & 1);
      s2[slot] = ta[j];
      s2[BM * 16 + slot] = tb[j];
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack8<T>(ra[i], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) bsum[q] += f[q];
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  auto compute = [&](int buf) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint4 fa[4], fb[4];
      const int kc = fk + 4 * sub;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + frow;
        fa[i] = sm[buf][row * 8 + (kc ^ swz128(row))];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + frow;
        fb[j] = sm[buf][BM * 8 + row * 8 + (kc ^ swz128(row))];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<T>::run(fa[i], fb[j], acc[i][j]);
    }
  };

  load_tiles(kbeg);
  store_lds(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < nk;
    if (more) load_tiles(kbeg + (ks + 1) * BK);
    compute(cur);
    if (more) store_lds(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: D row = co index = (lane>>4)*4 + reg, col = (r,s,ci) index = lane & 15 ----
  const long zoff = zb * p.sdwb + zh * p.sdwh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + (lane & 15);
    if (n >= p.Ktot) continue;
    const int ors = n / p.Cin, oci = n % p.Cin;
    if (oci >= p.Cin_out) continue;
    const long ocol = (long)ors * p.Cin_out + oci;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + q;
        if (m >= p.Cout_out) continue;
        const float v = p.alpha * acc[i][j][q];
        const long off = zoff + (long)m * p.lddw + ocol;
        if (p.out_mode == JG_OUT_ATOMIC_F32) {
          atomicAdd((float*)p.dw + off, v);
        } else if (p.out_mode == JG_OUT_STORE_F32) {
          ((float*)p.dw)[off] = v;
        } else {
          ((T*)p.dw)[off] = from_f32<T>(v);
        }
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = bsum[q];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      if (pq == 0 && (cm + q) < p.Cout_out) atomicAdd(p.dbias + cm + q, p.dbias_scale * v);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Variant 2: LDS tiles stay in the NHWC order of global memory ([pixel][channel], 256-byte rows,
// plain 16-byte staging writes) and the MFMA fragments are fetched with the gfx950 transposing
// LDS read ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of 4 consecutive
// channels (row = pixel k0 + (i >> 2), channels c0 + 4 (i & 3) ..) and receives channel c0 + i of
// the 4 pixels k0 .. k0+3 (measured lane semantics: tools/probe_tr.hip).  Two reads give the 8
// k-values of a 16x16x32 operand; A and B use the same pixel<->k assignment, which is all the MFMA
// needs.  The 32-byte (16-channel) block index of a row is XOR-swizzled with row bits {0,1,3} so a
// 32-lane service group of the transposing read touches all 64 banks exactly once.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz_tr(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// WAVES_M = 2: block tile 128 (co) x 128, waves 2x2 of 64x64.  WAVES_M = 1: block tile 64 (co) x 128,
// waves 1x4 of 64x32 -- for Cout <= 64 layers, where a 128-row tile would waste half of the MFMAs.
constexpr int TR_TILE = 64 * 16;  // uint4 chunks per operand tile (64 rows x 256 B)

// the body of wgrad_tn_tr_kernel for tile `bx` (column tile fastest) and (batch, split) index `bz` of problem `p`
template <typename T, int WAVES_M, bool DEEP>
__device__ __forceinline__ void wgrad_tn_tr_body(const WgP& p, const int bx, const int bz, uint4 (*sm)[2 * TR_TILE]) {
  constexpr int BM = 64 * WAVES_M, BN = 128, BK = 64;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TN = BN / WAVES_N / 16;  // 16-wide column tiles per wave: 4 or 2
  constexpr int TILE = TR_TILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int tilesN = (p.Ktot + BN - 1) / BN;
  const int n0 = (bx % tilesN) * BN;
  const int m0 = (bx / tilesN) * BM;

  const int z = bz;
  const int batch = z / p.splitk, split = z % p.splitk;
  const int zb = batch / p.nh, zh = batch % p.nh;
  const T* __restrict__ dy = (const T*)p.dy + zb * p.sdyb + zh * p.sdyh;
  const T* __restrict__ x = (const T*)p.x + zb * p.sxb + zh * p.sxh;

  int per = (p.Mpix + p.splitk - 1) / p.splitk;
  per = (per + BK - 1) / BK * BK;
  const int kbeg = split * per;
  const int kend = min(p.Mpix, kbeg + per);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + BK - 1) / BK;

  const int cidx = tid & 15;          // 16-byte chunk (8 channels) of the 128-channel row
  const int prow = (tid >> 4) * 4;    // 4 consecutive pixel rows per thread
  const int cm = m0 + cidx * 8;
  const bool a_ok = cm < p.Cout && cidx * 8 < BM;
  const int nn = n0 + cidx * 8;
  const bool b_ok = nn < p.Ktot;
  const int rs = nn / p.Cin, ci = nn % p.Cin;
  const int fr = rs / p.S, fs = rs % p.S;
  const bool quad_row = (p.Wo & 3) == 0;

  uint4 ra0[4], rb0[4], ra1[4], rb1[4];     // two register stages: the loads of K-steps k + 1 AND k + 2 are in flight under the MFMAs of step k
  float bsum[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bsum[q] = 0.f;
  const bool do_bias = p.dbias != nullptr && (bx % tilesN) == 0;

  auto load_tiles = [&](int kbase, uint4 (&ra)[4], uint4 (&rb)[4]) {
    const int p0 = kbase + prow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = p0 + i;
      const bool ok = a_ok && pp < kend;
      ra[i] = ldg16(dy + (ok ? (long)pp * p.lddy + cm : 0), ok);
    }
    if (quad_row) {
      const int ow0 = p0 % p.Wo;
      const int t = p0 / p.Wo;
      const int oh = t % p.Ho;
      const int b = t / p.Ho;
      const int ih = oh * p.stride + fr - p.pad;
      const bool row_ok = b_ok && (unsigned)ih < (unsigned)p.H;
      const long rowoff = ((long)(b * p.H + ih) * p.W) * p.ldx + ci;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iw = (ow0 + i) * p.stride + fs - p.pad;
        const bool ok = row_ok && (p0 + i) < kend && (unsigned)iw < (unsigned)p.W;
        rb[i] = ldg16(x + (ok ? rowoff + (long)iw * p.ldx : 0), ok);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = p0 + i;
        const int ow = pp % p.Wo;
        const int t = pp / p.Wo;
        const int oh = t % p.Ho;
        const int b = t / p.Ho;
        const int ih = oh * p.stride + fr - p.pad;
        const int iw = ow * p.stride + fs - p.pad;
        const bool ok = b_ok && pp < kend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        rb[i] = ldg16(x + (ok ? ((long)(b * p.H + ih) * p.W + iw) * p.ldx + ci : 0), ok);
      }
    }
  };
  auto store_lds = [&](int buf, const uint4 (&ra)[4], const uint4 (&rb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = prow + i;
      const int pos = row * 16 + ((((cidx >> 1) ^ swz_tr(row)) << 1) | (cidx & 1));
      sm[buf][pos] = ra[i];
      sm[buf][TILE + pos] = rb[i];
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack8<T>(ra[i], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) bsum[q] += f[q];
      }
    }
  };

  f32x4 acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int i16 = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
  auto frag = [&](int buf, int tile, int cb, int sub) -> uint4 {
    const char* base = reinterpret_cast<const char*>(&sm[buf][tile]);
    uint32_t w[4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int row = sub * 32 + g * 8 + rd * 4 + (i16 >> 2);
      const int off = (row * 16 + ((cb ^ swz_tr(row)) << 1)) * 16 + (i16 & 3) * 8;
      const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + off));
      const uint2 u = __builtin_bit_cast(uint2, v);
      w[2 * rd] = u.x;
      w[2 * rd + 1] = u.y;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint4 fa[4], fb[TN];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag(buf, 0, wm * 4 + i, sub);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = frag(buf, TILE, wn * TN + j, sub);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mfma<T>::run(fa[i], fb[j], acc[i][j]);
    }
  };

  // Two K-steps of global loads in flight per thread (round 5; one before): with a single register stage a workgroup had 32 KB in flight
  // for the ~4 us a loaded HBM round trip takes here (2 workgroups per CU: 4.0 TB/s).  The loads are unconditional -- past the end of the
  // slice they are predicated to the base address -- so the compiler's vmcnt waits are exact counts (store_lds of stage k + 1 waits for ITS
  // eight loads and leaves the eight of stage k + 2 in flight).
  if constexpr (DEEP) {
    load_tiles(kbeg, ra0, rb0);
    load_tiles(kbeg + BK, ra1, rb1);
    store_lds(0, ra0, rb0);
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
      load_tiles(kbeg + (ks + 2) * BK, ra0, rb0);
      compute(0);
      store_lds(1, ra1, rb1);
      __syncthreads();
      if (ks + 1 >= nk) break;
      load_tiles(kbeg + (ks + 3) * BK, ra1, rb1);
      compute(1);
      store_lds(0, ra0, rb0);
      __syncthreads();
    }
  } else {
    load_tiles(kbeg, ra0, rb0);
    store_lds(0, ra0, rb0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      const int cur = ks & 1;
      const bool more = ks + 1 < nk;
      if (more) load_tiles(kbeg + (ks + 1) * BK, ra0, rb0);
      compute(cur);
      if (more) store_lds(cur ^ 1, ra0, rb0);
      __syncthreads();
    }
  }

  const long zoff = zb * p.sdwb + zh * p.sdwh;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (TN * 16) + j * 16 + (lane & 15);
    if (n >= p.Ktot) continue;
    const int ors = n / p.Cin, oci = n % p.Cin;
    if (oci >= p.Cin_out) continue;
    const long ocol = (long)ors * p.Cin_out + oci;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + q;
        if (m >= p.Cout_out) continue;
        const float v = p.alpha * acc[i][j][q];
        const long off = zoff + (long)m * p.lddw + ocol;
        if (p.out_mode == JG_OUT_ATOMIC_F32) {
          atomicAdd((float*)p.dw + off, v);
        } else if (p.out_mode == JG_OUT_STORE_F32) {
          ((float*)p.dw)[off] = v;
        } else {
          ((T*)p.dw)[off] = from_f32<T>(v);
        }
      }
    }
  }
  if (do_bias) {
    // threads sharing a channel octet: same (tid & 15) -> lanes l, l+16, l+32, l+48 of every wave;
    // then across the 4 waves through LDS, so that a block issues ONE atomic per channel (the
    // bias gradient has only Cout addresses: per-lane atomics serialise thousands deep).
    float* s_b = reinterpret_cast<float*>(&sm[0][0]);  // all waves are past the last barrier of the K loop
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = bsum[q];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 16) s_b[wave * 128 + cidx * 8 + q] = v;
    }
    __syncthreads();
    if (tid < BM && (m0 + tid) < p.Cout_out)
      atomicAdd(p.dbias + m0 + tid, p.dbias_scale * (s_b[tid] + s_b[128 + tid] + s_b[256 + tid] + s_b[384 + tid]));
  }
}

// XCD-aware (tile, K-slice) order (round 6): workgroup b runs on XCD b % 8, and with the plain order the tiles of ONE K-slice -- which read the same
// pixel rows of dy and x -- sat in eight different L2s (a 512 x 1024 weight on the 256 x 256 tile: dy crossed HBM four times, x twice; PMC 232 MB
// read per launch for 100 MB of operands).  Each XCD now owns a contiguous run of the slice-major ids: all tiles of a slice share an L2.
__device__ __forceinline__ void wgrad_xcd_order(int& bx, int& bz) {
  const int tiles = gridDim.x, nwg = tiles * gridDim.z, lin = blockIdx.x + blockIdx.z * tiles;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = lin & 7, idx = lin >> 3;
  const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  bx = id % tiles;
  bz = id / tiles;
}

template <typename T, int WAVES_M, bool DEEP = false>
__global__ __launch_bounds__(256, 2) void wgrad_tn_tr_kernel(WgP p) {
  __shared__ uint4 sm[2][2 * TR_TILE];
  int bx, bz;
  wgrad_xcd_order(bx, bz);
  wgrad_tn_tr_body<T, WAVES_M, DEEP>(p, bx, bz, sm);
}

// ---------------------------------------------------------------------------------------------
// BIG tile (round 6): 256 (co) x 256 (r, s, ci) per workgroup of 8 waves (2 x 4 of 128 x 64), K steps of 64 pixels.  The 128 x 128 tile reads
// every operand row once per tile COLUMN / ROW it takes part in: a 256 -> 256 point-wise layer (the mobile ResNet blocks: 36 of them per CUT
// step at 131 K pixels) crosses L2 -> LDS twice per operand and runs at 300 TFLOP/s, bound by that traffic.  Here the whole 256 x 256 weight
// gradient is ONE tile: dy and x are read once (268 MB per layer = its HBM time), the fragment reads per MFMA drop from 0.5 to 0.375.  Same
// LDS layout as above with 512-byte pixel rows (32 chunks), one register stage of global loads, two LDS buffers of 64 KB.
constexpr int BIG_CH = 32;                  // 16-byte chunks per 256-channel pixel row
constexpr int BIG_TILE = 64 * BIG_CH;       // chunks per operand tile (64 pixel rows)
template <typename T>
__device__ __forceinline__ void wgrad_tn_big_body(const WgP& p, const int bx, const int bz, uint4 (*sm)[2 * BIG_TILE]) {
  constexpr int BM = 256, BN = 256, BK = 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves of 128 (co) x 64 (columns)

  const int tilesN = (p.Ktot + BN - 1) / BN;
  const int n0 = (bx % tilesN) * BN;
  const int m0 = (bx / tilesN) * BM;
  const int batch = bz / p.splitk, split = bz % p.splitk;
  const int zb = batch / p.nh, zh = batch % p.nh;
  const T* __restrict__ dy = (const T*)p.dy + zb * p.sdyb + zh * p.sdyh;
  const T* __restrict__ x = (const T*)p.x + zb * p.sxb + zh * p.sxh;

  int per = (p.Mpix + p.splitk - 1) / p.splitk;
  per = (per + BK - 1) / BK * BK;
  const int kbeg = split * per;
  const int kend = min(p.Mpix, kbeg + per);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + BK - 1) / BK;

  const int cidx = tid & 31;          // 16-byte chunk (8 channels) of the 256-channel row
  const int prow = (tid >> 5) * 4;    // 4 consecutive pixel rows per thread
  const int cm = m0 + cidx * 8;
  const bool a_ok = cm < p.Cout;
  const int nn = n0 + cidx * 8;
  const bool b_ok = nn < p.Ktot;
  const int rs = nn / p.Cin, ci = nn % p.Cin;
  const int fr = rs / p.S, fs = rs % p.S;

  uint4 ra[4], rb[4];
  float bsum[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bsum[q] = 0.f;
  const bool do_bias = p.dbias != nullptr && (bx % tilesN) == 0;

  auto load_tiles = [&](int kbase) {
    const int p0 = kbase + prow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = p0 + i;
      const bool ok = a_ok && pp < kend;
      ra[i] = ldg16(dy + (ok ? (long)pp * p.lddy + cm : 0), ok);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = p0 + i;
      const int ow = pp % p.Wo;
      const int t = pp / p.Wo;
      const int oh = t % p.Ho;
      const int b = t / p.Ho;
      const int ih = oh * p.stride + fr - p.pad;
      const int iw = ow * p.stride + fs - p.pad;
      const bool ok = b_ok && pp < kend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      rb[i] = ldg16(x + (ok ? ((long)(b * p.H + ih) * p.W + iw) * p.ldx + ci : 0), ok);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = prow + i;
      const int pos = row * BIG_CH + ((((cidx >> 1) ^ swz_tr(row)) << 1) | (cidx & 1));
      sm[buf][pos] = ra[i];
      sm[buf][BIG_TILE + pos] = rb[i];
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack8<T>(ra[i], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) bsum[q] += f[q];
      }
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int i16 = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
  auto frag = [&](int buf, int tile, int cb, int sub) -> uint4 {
    const char* base = reinterpret_cast<const char*>(&sm[buf][tile]);
    uint32_t w[4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int row = sub * 32 + g * 8 + rd * 4 + (i16 >> 2);
      const int off = (row * BIG_CH + ((cb ^ swz_tr(row)) << 1)) * 16 + (i16 & 3) * 8;
      const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + off));
      const uint2 u = __builtin_bit_cast(uint2, v);
      w[2 * rd] = u.x;
      w[2 * rd + 1] = u.y;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint4 fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = frag(buf, BIG_TILE, wn * 4 + j, sub);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint4 fa = frag(buf, 0, wm * 8 + i, sub);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<T>::run(fa, fb[j], acc[i][j]);
      }
    }
  };

  load_tiles(kbeg);
  store_lds(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < nk;
    if (more) load_tiles(kbeg + (ks + 1) * BK);
    compute(cur);
    if (more) store_lds(cur ^ 1);
    __syncthreads();
  }

  const long zoff = zb * p.sdwb + zh * p.sdwh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + (lane & 15);
    if (n >= p.Ktot) continue;
    const int ors = n / p.Cin, oci = n % p.Cin;
    if (oci >= p.Cin_out) continue;
    const long ocol = (long)ors * p.Cin_out + oci;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 128 + i * 16 + (lane >> 4) * 4 + q;
        if (m >= p.Cout_out) continue;
        atomicAdd((float*)p.dw + zoff + (long)m * p.lddw + ocol, p.alpha * acc[i][j][q]);
      }
    }
  }
  if (do_bias) {
    // lanes l, l + 32 of a wave share a channel octet (tid & 31); then across the 8 waves through LDS: one atomic per channel and workgroup
    float* s_b = reinterpret_cast<float*>(&sm[0][0]);      // all waves are past the last barrier of the K loop
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = bsum[q];
      v += __shfl_xor(v, 32);
      if (lane < 32) s_b[wave * 256 + cidx * 8 + q] = v;
    }
    __syncthreads();
    if (tid < BM && (m0 + tid) < p.Cout_out) {
      float v = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) v += s_b[w8 * 256 + tid];
      atomicAdd(p.dbias + m0 + tid, p.dbias_scale * v);
    }
  }
}

// ---- the same tile with LDS-DMA staging (round 6, second form) --------------------------------------------------------------------------
// The register-staged body above spends a K step on four things one after the other -- global loads into registers, ds_write of 64 KB,
// fragment reads, 64 MFMAs per wave -- behind two barriers: 2.9 us per step where the MFMAs alone need 0.85 (tools/wgrad_big_probe.py).
// The LDS tiles ARE the global layout (512-byte pixel rows), so a `global_load_lds_dwordx4` per 16-byte chunk puts them there without
// registers: the XOR swizzle moves to the SOURCE side (the lane that fills slot p of a row fetches chunk p ^ swizzle), the loads of step k + 1
// are in flight under the MFMAs of step k, and the loop has ONE barrier per step (the wait for this wave's loads, then the barrier, tell
// every wave both that buffer `cur` is complete and that nobody reads `cur ^ 1` any more).  The bias gradient comes from the dy fragments
// through an MFMA with a ones operand (wgrad_kxk.hip).
__device__ uint4 jg_wt_zero_page = {0u, 0u, 0u, 0u};
__device__ __forceinline__ void glds16_tn(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// SPREAD: one DMA per 8 MFMAs instead of the burst after the barrier -- measured equal (292.9 vs 292.2 us at 128 K steps per workgroup): off
template <typename T, bool SPREAD = false, int ABL = 0>      // ABL (timing only, wrong results): 1 = no MFMAs, 2 = no fragment reads, 3 = no DMA
__device__ __forceinline__ void wgrad_tn_dma_body(const WgP& p, const int bx, const int bz, uint4 (*sm)[2 * BIG_TILE]) {
  constexpr int BM = 256, BN = 256, BK = 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int tilesN = (p.Ktot + BN - 1) / BN;
  const int n0 = (bx % tilesN) * BN;
  const int m0 = (bx / tilesN) * BM;
  const int batch = bz / p.splitk, split = bz % p.splitk;
  const int zb = batch / p.nh, zh = batch % p.nh;
  const T* __restrict__ dy = (const T*)p.dy + zb * p.sdyb + zh * p.sdyh;
  const T* __restrict__ x = (const T*)p.x + zb * p.sxb + zh * p.sxh;
  const T* zp = reinterpret_cast<const T*>(&jg_wt_zero_page);

  int per = (p.Mpix + p.splitk - 1) / p.splitk;
  per = (per + BK - 1) / BK * BK;
  const int kbeg = split * per;
  const int kend = min(p.Mpix, kbeg + per);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + BK - 1) / BK;

  typedef __attribute__((address_space(3))) char* lds_cptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(char*)&sm[0][0];

  // slot q = rd * 512 + tid of a tile (rd = 0 .. 3): pixel row q / 32, slot p = q % 32 of the row <- source chunk c = p ^ swizzle(row)
  int a_col[4], b_ci[4], b_fr[4], b_fs[4];
  bool a_ok[4], b_ok[4];
#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    const int q = rd * 512 + tid;
    const int row = q >> 5, ps = q & 31;
    const int c = (((ps >> 1) ^ swz_tr(row)) << 1) | (ps & 1);
    a_col[rd] = m0 + c * 8;
    a_ok[rd] = a_col[rd] < p.Cout;
    const int nn = n0 + c * 8;
    b_ok[rd] = nn < p.Ktot;
    const int rs = nn / p.Cin;
    b_ci[rd] = nn % p.Cin;
    b_fr[rd] = rs / p.S;
    b_fs[rd] = rs % p.S;
  }
  // one of the eight DMA instructions of a K step (0 .. 3: dy rows, 4 .. 7: x rows); spread over the MFMA loop (SPREAD) or issued in a burst
  auto issue_one = [&](int kbase, int buf, int idx) {
    const unsigned l0 = lds0 + (unsigned)((buf * 2 * BIG_TILE + wave * 64) * 16);
    const int rd = idx & 3;
    const int pp = kbase + ((rd * 512 + tid) >> 5);
    if (idx < 4) {
      const bool ok = a_ok[rd] && pp < kend;
      glds16_tn(ok ? dy + (long)pp * p.lddy + a_col[rd] : zp, l0 + rd * 512 * 16);
    } else {
      const int ow = pp % p.Wo;
      const int t = pp / p.Wo;
      const int oh = t % p.Ho;
      const int b = t / p.Ho;
      const int ih = oh * p.stride + b_fr[rd] - p.pad;
      const int iw = ow * p.stride + b_fs[rd] - p.pad;
      const bool ok = b_ok[rd] && pp < kend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      glds16_tn(ok ? x + ((long)(b * p.H + ih) * p.W + iw) * p.ldx + b_ci[rd] : zp, l0 + (BIG_TILE + rd * 512) * 16);
    }
  };
  auto issue = [&](int kbase, int buf) {
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) issue_one(kbase, buf, idx);
  };

  f32x4 acc[8][4], accb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = p.dbias != nullptr && (bx % tilesN) == 0 && wn == 0;
  const uint32_t one1 = to_bits<T>(from_f32<T>(1.0f));
  const uint32_t one2 = one1 | (one1 << 16);
  const uint4 ones = make_uint4(one2, one2, one2, one2);

  const int i16 = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
  auto frag = [&](int buf, int tile, int cb, int sub) -> uint4 {
    if constexpr (ABL == 2) return make_uint4((unsigned)(buf + tile), (unsigned)cb, (unsigned)sub, (unsigned)lane);
    const char* base = reinterpret_cast<const char*>(&sm[buf][tile]);
    uint32_t w[4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int row = sub * 32 + g * 8 + rd * 4 + (i16 >> 2);
      const int off = (row * BIG_CH + ((cb ^ swz_tr(row)) << 1)) * 16 + (i16 & 3) * 8;
      const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + off));
      const uint2 u = __builtin_bit_cast(uint2, v);
      w[2 * rd] = u.x;
      w[2 * rd + 1] = u.y;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  };

  if constexpr (ABL != 3) issue(kbeg, 0);
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of buffer `cur` has landed
    __builtin_amdgcn_s_barrier();                           // everybody's has, and everybody is past the MFMAs that read `cur ^ 1`
    const bool more = ks + 1 < nk;
    if (more && !SPREAD && ABL != 3) issue(kbeg + (ks + 1) * BK, cur ^ 1);
    // the dy fragment of row block i + 1 is requested before the four MFMAs of row block i are issued (the compiler otherwise reads, waits,
    // multiplies, reads ... and both waves of a SIMD wait in step)
    uint4 fa_n = frag(cur, 0, wm * 8, 0);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint4 fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = frag(cur, BIG_TILE, wn * 4 + j, sub);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint4 fa = fa_n;
        if (i < 7) fa_n = frag(cur, 0, wm * 8 + i + 1, sub);
        else if (sub == 0) fa_n = frag(cur, 0, wm * 8, 1);
        if (SPREAD && more && (i & 1) == 0) {      // one DMA per 8 MFMAs, fenced: the scheduler otherwise gathers the eight into one burst
          __builtin_amdgcn_sched_barrier(0);
          issue_one(kbeg + (ks + 1) * BK, cur ^ 1, sub * 4 + (i >> 1));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (do_bias) accb[i] = Mfma<T>::run(fa, ones, accb[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (ABL == 1) { acc[i][j][0] += __builtin_bit_cast(float, fa.x ^ fb[j].y); acc[i][j][1] += __builtin_bit_cast(float, fa.z ^ fb[j].w); }
          else acc[i][j] = Mfma<T>::run(fa, fb[j], acc[i][j]);
        }
      }
    }
  }

  const long zoff = zb * p.sdwb + zh * p.sdwh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + (lane & 15);
    if (n >= p.Ktot) continue;
    const int ors = n / p.Cin, oci = n % p.Cin;
    if (oci >= p.Cin_out) continue;
    const long ocol = (long)ors * p.Cin_out + oci;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 128 + i * 16 + (lane >> 4) * 4 + q;
        if (m >= p.Cout_out) continue;
        atomicAdd((float*)p.dw + zoff + (long)m * p.lddw + ocol, p.alpha * acc[i][j][q]);
      }
    }
  }
  if (do_bias && (lane & 15) == 0) {      // the columns of accb are all equal (B = ones): column 0
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 128 + i * 16 + (lane >> 4) * 4 + q;
        if (m < p.Cout_out) atomicAdd(p.dbias + m, p.dbias_scale * accb[i][q]);
      }
  }
}

// (A RING form -- K steps of 32 pixels through four 32 KB stages, three in flight, constant vmcnt(8) -- measured 300 us against 293 at 128 K steps
// per workgroup: the step is not bounded by the round trip of its loads either.  What bounds it is the issue of the DMA instructions
// themselves, ~300 cycles each for the wave that issues them; removed.)
template <typename T, bool DMA = true, int ABL = 0>
__global__ __launch_bounds__(512, 1) void wgrad_tn_big_kernel(WgP p) {
  __shared__ uint4 sm[2][2 * BIG_TILE];
  int bx, bz;
  wgrad_xcd_order(bx, bz);
  if constexpr (DMA) wgrad_tn_dma_body<T, ABL == 4, ABL == 4 ? 0 : ABL>(p, bx, bz, sm);      // ABL 4: the SPREAD form (A/B)
  else wgrad_tn_big_body<T>(p, bx, bz, sm);
}

// GROUPED launch (round 5): up to GROUP_MAX independent small weight-gradient problems in ONE grid.  The SegFormer generator's backward issues
// ~190 of these per step (linear layers / 1x1 convolutions of four stages at 8 ... 64 workgroups each, 10 - 40 us apiece because their
// reduction over 10^4 - 10^5 tokens is a serial K loop per workgroup); side by side they fill the chip and the group costs what its
// slowest member costs.  The descriptors travel BY VALUE in the kernel argument block (16 x 200 B < 4 KB): no device table to fill, and a
// hipGraph capture bakes them into the node.
constexpr int GROUP_MAX = 16;
struct WgGroup {
  WgP p[GROUP_MAX];
  int start[GROUP_MAX + 1];     // first workgroup of problem i in the 1-D grid; start[n] = grid size
  int tiles[GROUP_MAX];         // output tiles of problem i (its workgroups = tiles x splitk)
  int n;
};
template <typename T, int WAVES_M, bool DEEP = false>
__global__ __launch_bounds__(256, 2) void wgrad_tn_tr_group_kernel(const WgGroup g) {
  __shared__ uint4 sm[2][2 * TR_TILE];
  const int b = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k) i += (k < g.n && b >= g.start[k]) ? 1 : 0;
  i = __builtin_amdgcn_readfirstlane(i);
  const int local = b - g.start[i];
  wgrad_tn_tr_body<T, WAVES_M, DEEP>(g.p[i], local % g.tiles[i], local / g.tiles[i], sm);
}

template <typename T, bool DMA = true>
__global__ __launch_bounds__(512, 1) void wgrad_tn_big_group_kernel(const WgGroup g) {
  __shared__ uint4 sm[2][2 * BIG_TILE];
  const int b = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k) i += (k < g.n && b >= g.start[k]) ? 1 : 0;
  i = __builtin_amdgcn_readfirstlane(i);
  const int local = b - g.start[i];
  if constexpr (DMA) wgrad_tn_dma_body<T>(g.p[i], local % g.tiles[i], local / g.tiles[i], sm);
  else wgrad_tn_big_body<T>(g.p[i], local % g.tiles[i], local / g.tiles[i], sm);
}

// the big tile pays where a 128 x 128 tiling would re-read: at least 256 output channels and 256 columns, and a long reduction
static inline bool wg_big_ok(const WgP& p) {
  // (its epilogue is the fp32-atomic accumulation only: the store modes of the batched attention products stay on the 128 x 128 tile)
  return jg_tune(JG_TUNE_WGRAD_BIG) != 0 && p.out_mode == JG_OUT_ATOMIC_F32 && p.Cout >= 256 && p.Ktot >= 256 && p.Mpix >= 8192;
}

}  // namespace

static int make_wgp(const jg_wgrad_args* a, WgP& p) {
  if (!a || !a->dy || !a->x || !a->dw) return JG_ERR_BAD_ARG;
  if (a->Cin % 8 || a->Cout % 8 || a->ldx % 8 || a->lddy % 8) return JG_ERR_BAD_ARG;
  if (a->nbatch < 1 || a->nh < 1 || a->splitk < 1) return JG_ERR_BAD_ARG;
  if ((long)a->nbatch * a->splitk > 65535) return JG_ERR_BAD_ARG;
  if (a->out_mode != JG_OUT_ATOMIC_F32 && a->splitk != 1) return JG_ERR_BAD_ARG;
  if (a->dbias && a->out_mode != JG_OUT_ATOMIC_F32) return JG_ERR_BAD_ARG;
  const long Mpix = (long)a->B * a->Ho * a->Wo;
  if (Mpix <= 0 || Mpix > (1L << 30)) return JG_ERR_BAD_ARG;
  p.dy = (const char*)a->dy; p.x = (const char*)a->x; p.dw = (char*)a->dw; p.dbias = a->dbias;
  p.Mpix = (int)Mpix; p.Cout = a->Cout; p.Ktot = a->R * a->S * a->Cin;
  p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.R = a->R; p.S = a->S; p.pad = a->pad; p.stride = a->stride;
  p.Ho = a->Ho; p.Wo = a->Wo;
  p.Cin_out = a->Cin_out > 0 ? a->Cin_out : a->Cin;
  p.Cout_out = a->Cout_out > 0 ? a->Cout_out : a->Cout;
  p.lddy = a->lddy; p.ldx = a->ldx; p.lddw = a->lddw;
  p.nh = a->nh; p.splitk = a->splitk;
  // JG_DETERMINISTIC 1: no split over the pixels -- every element of dw (and of dbias) is then produced by ONE thread of ONE workgroup, its
  // atomicAdd a single add onto a reproducible value (the halo-resident kernels choose their own split: pick_split there)
  if (jg_tune(JG_TUNE_DETERMINISTIC) != 0 && a->out_mode == JG_OUT_ATOMIC_F32) p.splitk = 1;
  p.sdyb = a->sdyb; p.sdyh = a->sdyh; p.sxb = a->sxb; p.sxh = a->sxh; p.sdwb = a->sdwb; p.sdwh = a->sdwh;
  p.alpha = a->alpha; p.out_mode = a->out_mode; p.B = a->B;
  p.dbias_scale = a->dbias_scale != 0.f ? a->dbias_scale : 1.f;
  if (a->pad_mode != 0 && a->pad_mode != 1) return JG_ERR_BAD_ARG;
  p.reflect = a->pad_mode == 1;
  if (a->x_mode != 0 && a->x_mode != 1) return JG_ERR_BAD_ARG;
  p.x_up = a->x_mode == 1;
  if (p.x_up && (p.reflect || (a->H & 1) || (a->W & 1) || a->nbatch != 1)) return JG_ERR_BAD_ARG;
  return JG_OK;
}

// n independent weight-gradient problems (any convolution geometry of jg_conv2d_wgrad_tn; nbatch 1, atomic accumulation, no mirrored borders /
// upsample-on-read) in grouped launches of at most GROUP_MAX: same arithmetic per problem as jg_conv2d_wgrad_tn (the summation order of the
// atomics differs).
extern "C" int jg_conv2d_wgrad_tn_group(int dtype, const jg_wgrad_args* a, int n, jg_stream_t stream) {
  if (!a || n < 1 || n > 4096) return JG_ERR_BAD_ARG;
  // problems that one of the halo-resident kernels serves (3x3 / 7x7 stride 1 at their channel multiples) are launched by it, singly: those
  // fill the chip by themselves; everything else is the im2col TN kernel and goes into the groups
  // Every descriptor is validated BEFORE anything is launched (a bad descriptor in the middle used to leave part of the group
  // accumulated into the gradient arena).  The grouped kernel is the transposing-read TN kernel of JG_WGRAD_VARIANT >= 2; variant 1 (register
  // transpose) has no grouped form and is refused rather than silently replaced, so that an A/B of the variants measures what it names.
  const int variant = jg_tune(JG_TUNE_WGRAD_VARIANT);
  if (variant < 2) return JG_ERR_UNSUPPORTED;
  std::vector<WgP> ps((size_t)n);
  std::vector<char> special((size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    const int rc = make_wgp(a + i, ps[i]);
    if (rc != JG_OK) return rc;
    if (a[i].nbatch != 1 || a[i].out_mode != JG_OUT_ATOMIC_F32 || ps[i].reflect || ps[i].x_up) return JG_ERR_UNSUPPORTED;
    special[i] = variant >= 4 && (a[i].R > 1 || a[i].S > 1) &&
                 (jg_wgrad_halo_try(dtype, ps[i], 1, (hipStream_t)stream, true) || jg_wgrad_kxk_try(dtype, ps[i], 1, (hipStream_t)stream, true));
  }
  for (int i = 0; i < n; ++i)
    if (special[i] && !jg_wgrad_halo_try(dtype, ps[i], 1, (hipStream_t)stream)) jg_wgrad_kxk_try(dtype, ps[i], 1, (hipStream_t)stream);
  const int group_blocks = jg_tune(JG_TUNE_WGRAD_GROUP_BLOCKS);
  const size_t gpad = (size_t)jg_tune(JG_TUNE_WGRAD_LDS_PAD);      // A/B: unused dynamic LDS = one workgroup per CU instead of two
  for (int wavesm = 1; wavesm <= 3; ++wavesm) {      // 1: 64-row tile, 2: 128-row tile, 3: the 256 x 256 tile
    WgGroup g;
    g.n = 0;
    g.start[0] = 0;
    auto flush = [&]() -> int {
      if (!g.n) return JG_OK;
      // Round 6: the callers' split-K factors are chosen for a launch that has the chip to itself (JG_WGRAD_TARGET_BLOCKS workgroups PER PROBLEM:
      // up to 512 slices of four K-steps each, every one ending in a 32 - 64 KB burst of fp32 atomics).  Side by side the problems of a group
      // fill the chip together: the K-steps of the whole group are dealt out over `group_blocks` workgroups of equal length, a problem is never
      // split further than its caller asked.
      if (group_blocks > 0) {
        long total = 0;
        for (int i = 0; i < g.n; ++i) total += (long)g.tiles[i] * ((g.p[i].Mpix + 63) / 64);
        const long len = std::max<long>(4, (total + group_blocks - 1) / group_blocks);      // K-steps per workgroup
        for (int i = 0; i < g.n; ++i) {
          const long ks = (g.p[i].Mpix + 63) / 64;
          g.p[i].splitk = (int)std::max<long>(1, std::min<long>(g.p[i].splitk, (ks + len - 1) / len));
          g.start[i + 1] = g.start[i] + g.tiles[i] * g.p[i].splitk;
        }
      }
      const dim3 grid(g.start[g.n]);
      const int deep = jg_tune(JG_TUNE_WGRAD_DEEP);      // two register stages of global loads in flight: bit 0 = the 64-row tile, bit 1 = the 128-row tile
      if (wavesm == 3) {
        if (jg_tune(JG_TUNE_WGRAD_BIG) == 2) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_group_kernel<T, false>), grid, dim3(512), 0, (hipStream_t)stream, g);); }
        else { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_group_kernel<T, true>), grid, dim3(512), 0, (hipStream_t)stream, g);); }
      } else if (wavesm == 1) {
        if (deep & 1) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_group_kernel<T, 1, true>), grid, dim3(256), gpad, (hipStream_t)stream, g);); }
        else { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_group_kernel<T, 1>), grid, dim3(256), gpad, (hipStream_t)stream, g);); }
      } else {
        if (deep & 2) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_group_kernel<T, 2, true>), grid, dim3(256), gpad, (hipStream_t)stream, g);); }
        else { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_group_kernel<T, 2>), grid, dim3(256), gpad, (hipStream_t)stream, g);); }
      }
      g.n = 0;
      return JG_OK;
    };
    for (int i = 0; i < n; ++i) {
      const WgP& p = ps[i];
      if (special[i] || (wg_big_ok(p) ? 3 : p.Cout <= 64 ? 1 : 2) != wavesm) continue;
      const int tiles = wavesm == 3 ? ((p.Cout + 255) / 256) * ((p.Ktot + 255) / 256) : ((p.Cout + 64 * wavesm - 1) / (64 * wavesm)) * ((p.Ktot + 127) / 128);
      g.p[g.n] = p;
      g.tiles[g.n] = tiles;
      g.start[g.n + 1] = g.start[g.n] + tiles * p.splitk;
      if (++g.n == GROUP_MAX) {
        const int rc2 = flush();
        if (rc2 != JG_OK) return rc2;
      }
    }
    const int rc3 = flush();
    if (rc3 != JG_OK) return rc3;
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_conv2d_wgrad_tn(int dtype, const jg_wgrad_args* a, jg_stream_t stream) {
  jg_note_kernel("");
  WgP p;
  {
    const int rc = make_wgp(a, p);
    if (rc != JG_OK) return rc;
  }
  const int variant = jg_tune(JG_TUNE_WGRAD_VARIANT);  // 1: register transpose; 2: transposing LDS reads; 3: 2 with 128-row tiles only; 4: + halo-resident 3x3
  if (variant >= 4 && (jg_wgrad_halo_try(dtype, p, a->nbatch, (hipStream_t)stream) || jg_wgrad_kxk_try(dtype, p, a->nbatch, (hipStream_t)stream))) {
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  if (p.reflect || p.x_up) return JG_ERR_UNSUPPORTED;   // mirrored borders / upsample-on-read exist only in the halo-resident kernel
  if (variant >= 2 && wg_big_ok(p)) {
    const int tiles = ((p.Cout + 255) / 256) * ((p.Ktot + 255) / 256);
    // the caller's split-K is sized for 128 x 128 tiles: a quarter of the workgroups per slice here, so keep the slice count
    dim3 grid(tiles, 1, a->nbatch * p.splitk);
    jg_note_kernel("wgrad_tn_big_kernel");
    const int bigmode = jg_tune(JG_TUNE_WGRAD_BIG);      // 11 / 12 / 13: timing-only ablations of the DMA form (tools/wgrad_big_probe.py)
    if (bigmode == 11) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_kernel<T, true, 1>), grid, dim3(512), 0, (hipStream_t)stream, p);); }
    else if (bigmode == 12) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_kernel<T, true, 2>), grid, dim3(512), 0, (hipStream_t)stream, p);); }
    else if (bigmode == 14) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_kernel<T, true, 4>), grid, dim3(512), 0, (hipStream_t)stream, p);); }
    else if (bigmode == 13) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_kernel<T, true, 3>), grid, dim3(512), 0, (hipStream_t)stream, p);); }
    else if (bigmode == 2) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_kernel<T, false>), grid, dim3(512), 0, (hipStream_t)stream, p);); }
    else { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_big_kernel<T, true>), grid, dim3(512), 0, (hipStream_t)stream, p);); }
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  const int tilesN = (p.Ktot + 127) / 128;
  if (variant == 1) {
    dim3 grid(((p.Cout + 127) / 128) * tilesN, 1, a->nbatch * p.splitk);
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, p););
  } else if (p.Cout <= 64 && variant != 3) {
    dim3 grid(tilesN, 1, a->nbatch * p.splitk);
    if (jg_tune(JG_TUNE_WGRAD_DEEP) & 1) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_kernel<T, 1, true>), grid, dim3(256), 0, (hipStream_t)stream, p);); }
    else { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);); }
  } else {
    dim3 grid(((p.Cout + 127) / 128) * tilesN, 1, a->nbatch * p.splitk);
    if (jg_tune(JG_TUNE_WGRAD_DEEP) & 2) { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_kernel<T, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, p);); }
    else { JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((wgrad_tn_tr_kernel<T, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);); }
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
