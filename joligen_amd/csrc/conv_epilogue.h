// Coalesced convolution epilogue through LDS (16-bit outputs).
//
// The MFMA accumulator layout (lane = pixel l & 15 x 4 consecutive channels) turns a direct store
// into 8-byte pieces scattered over 16 pixel rows per instruction: 32-byte segments, store-issue
// bound (cdna_hip_programming.md T21).  Here every wave transposes its 64-pixel x 64-channel slab
// through a private 16 KB fp32 LDS scratch and leaves it row-wise: a lane owns 8 consecutive
// channels (16 bytes) of a pixel, 8 lanes cover a 128-byte line, so the residual read, the output
// store and the bias / statistics bookkeeping are all full-line 16-byte accesses.
//
//   y = alpha * acc + bias + res_scale * res         (fp32, one rounding)
//   statistics (optional): per-channel (sum, sum^2) of y over the wave's pixels, reduced over the
//   lanes that share a channel octet, handed to `flush(channel_octet_base, s1[8], s2[8])`.
//
// 16-byte scratch chunks are XOR-swizzled with (pixel & 15): the ds_write_b128 of a 16-lane group
// (16 pixels, one chunk column) and the two ds_read_b128 of the row-wise pass are conflict-free.
#pragma once
#include "conv_params.h"

// the lane's 8 bias values of the row-wise pass (channel octet lane & 7 of the 64-channel wave tile at nbase): fetched by the
// kernels BEFORE their K loop so that the epilogue does not start with an exposed global-load round trip
__device__ __forceinline__ void jg_epilogue_bias(const ConvP& p, int lane, int nbase, float* bias) {
  const int c8 = lane & 7;
  const bool nok = nbase + c8 * 8 < p.N;
#pragma unroll
  for (int q = 0; q < 8; ++q) bias[q] = (p.bias && nok) ? p.bias[nbase + c8 * 8 + q] : 0.f;
}

// acc[TN][TM]: TN = 4 channel tiles (64 channels), TM = 4 or 8 pixel tiles.  pix(lp) maps the local
// pixel index (0 .. TM*16-1) of this wave to the global pixel row (long, -1 = out of range).
// MFMA row r of a pixel tile <-> pixel column jg_pixperm(r) (PERM kernels): rows {0-3, 12-15} take the even
// columns, rows {4-11} the odd ones.  A ds_read_b128 service group mixes rows {0-3, 12-15} of k-chunk c with
// rows {4-11} of chunk c+1; with this assignment the two sets sit in different 128-byte halves of the bank
// row for EVERY horizontal tap shift, and the 8 same-parity pixels of a set get 8 distinct XOR swizzles:
// the shifted-tap fragment reads of conv_halo.hip become conflict-free (they were 2-way for s = 1, 2).
__device__ __forceinline__ int jg_pixperm(int r) { return r < 4 ? 2 * r : (r < 12 ? 2 * (r - 4) + 1 : 2 * (r - 8)); }

// rrow(lp, m): row of the residual tensor for local pixel lp / output row m (m itself, or the half-resolution row for res_up)
// prow(slab, r2, c2): y_pool (slabs of 4 image rows x 16 columns only): row of the POOLED output tensor of the slab's pooled pixel
//   (r2 in 0..1, c2 in 0..7); the 2x2 sum of alpha * acc is stored there, no bias / residual / statistics
// bias_pre: the lane's 8 bias values when the caller fetched them ahead of its K loop (jg_epilogue_bias), nullptr: fetched here
template <typename T, int TM, bool PERM = false, typename PixFn, typename ResRowFn, typename PoolRowFn, typename FlushFn>
__device__ __forceinline__ void jg_epilogue_lds(const ConvP& p, f32x4 (&acc)[4][TM], char* scratch, int lane, int nbase,
                                                int gimg, PixFn pix, ResRowFn rrow, PoolRowFn prow_fn, FlushFn flush,
                                                const float* bias_pre = nullptr) {
  static_assert(TM % 4 == 0, "slabs of 4 pixel tiles");
  const int l15 = lane & 15, lk = lane >> 4;
  const int c8 = lane & 7, prow = lane >> 3;
  const bool nok = nbase + c8 * 8 < p.N;   // channel octet inside the tensor (N % 8 == 0)
  float bias[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bias[q] = bias_pre ? bias_pre[q] : ((p.bias && nok) ? p.bias[nbase + c8 * 8 + q] : 0.f);
  // vmcnt counts loads AND stores, in order: a wait for a load result inside the store loop below also waits for every earlier
  // store of the wave (one HBM write round trip per iteration).  Pin the loaded values HERE (the empty asm "uses" them, so hipcc
  // places its wait before it): afterwards nothing the loop reads is pending and its stores stream out back to back.
#pragma unroll
  for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(bias[q]));
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s1[q] = s2[q] = 0.f;
  // GroupNorm-backward reduction mode: per-lane (a, b) of its 8 channels; gimg = image index of this tile
  float ga[8], gb[8];
  const bool gnr = p.stats && p.stats_mode == 1;
  if (gnr) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long o = ((long)gimg * p.N + nbase + c8 * 8 + q) * 2;
      ga[q] = nok ? p.gab[o] : 0.f;
      gb[q] = nok ? p.gab[o + 1] : 0.f;
    }
  }
  const T* gx = (const T*)p.gx;
  T* y = (T*)p.y;
  const T* res = (const T*)p.res;
#pragma unroll
  for (int slab = 0; slab < TM / 4; ++slab) {
    // The residual (and, in GroupNorm-backward reduction mode, the norm input) rows of this slab are requested FIRST, all eight per
    // lane at once: issued one by one next to their use they form a chain of eight dependent HBM round trips per wave (measured:
    // the epilogue of a 64 -> 64 layer at 256x256 took 5.6 us per tile without a residual and 11 us with one, i.e. as long as the
    // whole K loop); up front they overlap each other and the LDS transposition below.
    // (one register array serves both: a launch has a residual OR the reduction mode -- with both, the norm input is read in place)
    // The loads are UNCONDITIONAL (out-of-range lanes read row 0 of the same tensor, the value is dropped): a load under a
    // per-lane branch gets an `s_waitcnt vmcnt(0)` at the end of its block, which serialises the eight round trips again.
    uint4 pv[8];
    if (p.dbg & 32) res = nullptr;
    const bool pre_res = res != nullptr, pre_gx = gnr && !res;
    if (!p.y_pool && (pre_res || pre_gx)) {
      const T* pbase = pre_res ? res : gx;
      const long pld = pre_res ? p.ldres : p.gldx;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int pl = it * 8 + prow;
        const long m = nok ? pix(slab * 64 + pl) : -1L;
        const long mm = m >= 0 ? m : 0L;
        const long row = pre_res ? rrow(slab * 64 + pl, mm) : mm;
        pv[it] = *reinterpret_cast<const uint4*>(pbase + row * pld + (nok ? nbase + c8 * 8 : 0));
      }
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int pc = PERM ? jg_pixperm(l15) : l15;
      const int pl = ii * 16 + pc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int chunk = (j * 4 + lk) ^ pc;   // pl & 15 == pc
        float4 v = make_float4(p.alpha * acc[j][slab * 4 + ii][0], p.alpha * acc[j][slab * 4 + ii][1],
                               p.alpha * acc[j][slab * 4 + ii][2], p.alpha * acc[j][slab * 4 + ii][3]);
        *reinterpret_cast<float4*>(scratch + pl * 256 + chunk * 16) = v;
      }
    }
    if (!p.y_pool && (pre_res || pre_gx)) {   // pin the prefetched rows (see the bias comment): the loads had the LDS writes to land
#pragma unroll
      for (int it = 0; it < 8; ++it) asm volatile("" : "+v"(pv[it].x), "+v"(pv[it].y), "+v"(pv[it].z), "+v"(pv[it].w));
    }
    if (p.y_pool) {
      // 16 pooled pixels x 8 channel octets per slab = 2 items per lane; a lane sums its 2x2 window out of the scratch
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int pp = it * 8 + prow, r2 = pp >> 3, c2 = pp & 7;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          const int pl = (2 * r2 + (dd >> 1)) * 16 + 2 * c2 + (dd & 1);
          const float4 a = *reinterpret_cast<const float4*>(scratch + pl * 256 + (((2 * c8) ^ (pl & 15)) << 4));
          const float4 b = *reinterpret_cast<const float4*>(scratch + pl * 256 + (((2 * c8 + 1) ^ (pl & 15)) << 4));
          v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
          v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        if (nok) *reinterpret_cast<uint4*>(y + prow_fn(slab, r2, c2) * p.ldy + nbase + c8 * 8) = pack8<T>(v);
      }
      continue;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int pl = it * 8 + prow;
      const float4 a = *reinterpret_cast<const float4*>(scratch + pl * 256 + (((2 * c8) ^ (pl & 15)) << 4));
      const float4 b = *reinterpret_cast<const float4*>(scratch + pl * 256 + (((2 * c8 + 1) ^ (pl & 15)) << 4));
      float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const long m = nok ? pix(slab * 64 + pl) : -1L;
      if (m >= 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += bias[q];
        if (res) {
          float rf[8];
          const uint4 rval = pv[it];
          unpack8<T>(rval, rf);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] += p.res_scale * rf[q];
        }
        if (!(p.dbg & 8)) *reinterpret_cast<uint4*>(y + m * p.ldy + nbase + c8 * 8) = pack8<T>(v);
        if (gnr) {
          float xf[8];
          uint4 xval = pv[it];
          if (!pre_gx) xval = *reinterpret_cast<const uint4*>(gx + m * p.gldx + nbase + c8 * 8);
          unpack8<T>(xval, xf);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float du = v[q];
            du *= act_grad_rt(ga[q] * xf[q] + gb[q], p.gact);
            s1[q] += du;
            s2[q] += du * xf[q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            s1[q] += v[q];
            s2[q] += v[q] * v[q];
          }
        }
      }
    }
  }
  if (p.stats) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        s1[q] += __shfl_xor(s1[q], o);
        s2[q] += __shfl_xor(s2[q], o);
      }
    }
    if (prow == 0 && nok && !(p.dbg & 16)) flush(nbase + c8 * 8, s1, s2);
  }
}
