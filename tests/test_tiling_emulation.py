"""CPU: lane-level emulation of the two MFMA GEMM kernels' index math (csrc/gemm_nt.hip,
csrc/gemm_tn.hip): who stages which 16-byte chunk into which (swizzled) LDS slot, which slot each
lane reads as its MFMA fragment, and where each accumulator register is stored.  The MFMA itself
is modelled with the documented gfx950 layout of v_mfma_f32_16x16x32_{f16,bf16}
(cdna_hip_programming.md section 3): operand lane l holds row/col (l & 15), k-group (l >> 4), 8
consecutive k; D[row = (l >> 4) * 4 + reg][col = l & 15].

This is host-logic test infrastructure (no GPU): it transcribes the kernels' integer arithmetic
line by line so that a layout mistake is caught before burning GPU minutes; it also checks the LDS
swizzles are conflict-free under the service-group model of MI355X_MICROARCH.md (LDS table).
"""
import numpy as np
import pytest


def mfma_16x16x32(a_frag, b_frag, acc):
    """a_frag/b_frag: [64][8] per-lane operands; acc: [64][4]."""
    A = np.zeros((16, 32))
    Bm = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, (l >> 4) * 8:(l >> 4) * 8 + 8] = a_frag[l]
        Bm[(l >> 4) * 8:(l >> 4) * 8 + 8, l & 15] = b_frag[l]
    D = A @ Bm
    out = acc.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[(l >> 4) * 4 + r, l & 15]
    return out


def swz64(row):
    return ((row >> 3) & 1) << 1


def swz128(row):
    return (row >> 1) & 7


def emulate_conv_nt(x, w, BM, BN, WAVES_M, WAVES_N, pad, stride):
    B, H, W, Cin = x.shape
    N, R, S, _ = w.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    M, K = B * Ho * Wo, R * S * Cin
    y = np.zeros((M, N))
    WM, WN = BM // WAVES_M, BN // WAVES_N
    TM, TN = WM // 16, WN // 16
    A_CH, B_CH = BM * 4 // 256, BN * 4 // 256
    wf = w.reshape(N, K)
    tilesN = (N + BN - 1) // BN
    nblocks = ((M + BM - 1) // BM) * tilesN
    nk = (K + 31) // 32
    for bid in range(nblocks):
        n0, m0 = (bid % tilesN) * BN, (bid // tilesN) * BM
        acc = np.zeros((4, TN, TM, 64, 4))
        # per-thread state
        st = []
        for tid in range(256):
            kc, srow = tid & 3, tid >> 2
            rows = []
            for i in range(A_CH):
                m = m0 + srow + 64 * i
                ow, t = m % Wo, m // Wo
                oh, b = t % Ho, t // Ho
                ih0 = oh * stride - pad if m < M else -(1 << 28)
                rows.append((ih0, ow * stride - pad, b * H))
            c, rs0 = (kc * 8) % Cin, (kc * 8) // Cin
            st.append(dict(kc=kc, srow=srow, rows=rows, c=c, r=rs0 // S, s=rs0 % S, kg=kc * 8))
        for ks in range(nk):
            sm = np.zeros(((BM + BN) * 4, 8))
            for tid in range(256):
                t = st[tid]
                if ks > 0:  # advance_k
                    t["kg"] += 32
                    t["c"] += 32
                    while t["c"] >= Cin:
                        t["c"] -= Cin
                        t["s"] += 1
                        if t["s"] == S:
                            t["s"] = 0
                            t["r"] += 1
                kvalid = t["r"] < R
                for i in range(A_CH):
                    ih0, iw0, pb = t["rows"][i]
                    ih, iw = ih0 + t["r"], iw0 + t["s"]
                    ok = kvalid and 0 <= ih < H and 0 <= iw < W
                    v = x.reshape(-1, Cin)[(pb + ih) * W + iw, t["c"]:t["c"] + 8] if ok else np.zeros(8)
                    row = t["srow"] + 64 * i
                    sm[row * 4 + (t["kc"] ^ swz64(row))] = v
                for i in range(B_CH):
                    n = n0 + t["srow"] + 64 * i
                    ok = kvalid and n < N
                    v = wf[n, t["kg"]:t["kg"] + 8] if ok else np.zeros(8)
                    row = t["srow"] + 64 * i
                    sm[BM * 4 + row * 4 + (t["kc"] ^ swz64(row))] = v
            for wave in range(4):
                wm, wn = wave // WAVES_N, wave % WAVES_N
                fa = np.zeros((TM, 64, 8))
                fb = np.zeros((TN, 64, 8))
                for lane in range(64):
                    frow, fk = lane & 15, lane >> 4
                    for i in range(TM):
                        row = wm * WM + i * 16 + frow
                        fa[i, lane] = sm[row * 4 + (fk ^ swz64(row))]
                    for j in range(TN):
                        row = wn * WN + j * 16 + frow
                        fb[j, lane] = sm[BM * 4 + row * 4 + (fk ^ swz64(row))]
                for j in range(TN):
                    for i in range(TM):
                        acc[wave, j, i] = mfma_16x16x32(fb[j], fa[i], acc[wave, j, i])
        for wave in range(4):
            wm, wn = wave // WAVES_N, wave % WAVES_N
            for lane in range(64):
                for j in range(TN):
                    n = n0 + wn * WN + j * 16 + (lane >> 4) * 4
                    if n >= N:
                        continue
                    for i in range(TM):
                        m = m0 + wm * WM + i * 16 + (lane & 15)
                        if m >= M:
                            continue
                        y[m, n:n + 4] = acc[wave, j, i, lane]
    return y.reshape(B, Ho, Wo, N)


def ref_conv(x, w, pad, stride):
    B, H, W, Cin = x.shape
    N, R, S, _ = w.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    xp = np.zeros((B, H + 2 * pad, W + 2 * pad, Cin))
    xp[:, pad:pad + H, pad:pad + W] = x
    y = np.zeros((B, Ho, Wo, N))
    for r in range(R):
        for s in range(S):
            patch = xp[:, r:r + Ho * stride:stride, s:s + Wo * stride:stride]
            y += patch @ w[:, r, s].T
    return y


@pytest.mark.parametrize("cfg", [(128, 128, 2, 2), (256, 64, 4, 1)])
@pytest.mark.parametrize("shape", [(1, 6, 7, 16, 24, 3, 1, 1), (1, 5, 5, 8, 8, 3, 1, 1), (2, 4, 4, 40, 72, 1, 0, 1),
                                   (1, 8, 8, 16, 16, 3, 1, 2)])
def test_conv_nt_index_math(cfg, shape):
    B, H, W, Cin, Cout, k, pad, stride = shape
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, H, W, Cin))
    w = rng.standard_normal((Cout, k, k, Cin))
    y = emulate_conv_nt(x, w, *cfg, pad, stride)
    np.testing.assert_allclose(y, ref_conv(x, w, pad, stride), rtol=1e-9, atol=1e-9)


def test_dgrad_as_forward_conv_with_flipped_transposed_weights():
    """dx = conv(dy, w16T) with w16T[ci][r][s][co] = w[co][R-1-r][S-1-s][ci], pad' = R-1-pad
    (joligen_amd/ops.py conv2d_dgrad + csrc/optim.hip refresh_weights_kernel)."""
    rng = np.random.default_rng(1)
    B, H, W, Cin, Cout, R, pad = 1, 5, 6, 8, 16, 3, 1
    w = rng.standard_normal((Cout, R, R, Cin))
    dy = rng.standard_normal((B, H, W, Cout))
    # autograd-free reference: dx[b,y,x,ci] = sum dy[b,y-r+pad,x-s+pad,co] w[co,r,s,ci]
    dx = np.zeros((B, H, W, Cin))
    for y in range(H):
        for x_ in range(W):
            for r in range(R):
                for s in range(R):
                    oy, ox = y - r + pad, x_ - s + pad
                    if 0 <= oy < H and 0 <= ox < W:
                        dx[:, y, x_] += dy[:, oy, ox] @ w[:, r, s]
    RS = R * R
    wT = np.zeros((Cin, RS, Cout))
    wflat = w.reshape(Cout, RS, Cin)
    for rs in range(RS):
        wT[:, rs, :] = wflat[:, RS - 1 - rs, :].T
    got = ref_conv(dy, wT.reshape(Cin, R, R, Cout), R - 1 - pad, 1)
    np.testing.assert_allclose(got, dx, rtol=1e-9, atol=1e-9)


def transpose4x8(ra):
    """ra: [4 pixels][8 channels] -> out[j] = 4 pixels of channel j (csrc/gemm_tn.hip)."""
    return [np.array([ra[0][j], ra[1][j], ra[2][j], ra[3][j]]) for j in range(8)]


def emulate_wgrad_tn(dy, x, R, S, pad, stride, splitk):
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    Mpix, Ktot = B * Ho * Wo, R * S * Cin
    BM = BN = 128
    BK = 64
    dw = np.zeros((Cout, Ktot))
    dbias = np.zeros(Cout)
    dyf, xf = dy.reshape(Mpix, Cout), x.reshape(-1, Cin)
    tilesN = (Ktot + BN - 1) // BN
    nblocks = ((Cout + 127) // 128) * tilesN
    for split in range(splitk):
        per = (Mpix + splitk - 1) // splitk
        per = (per + BK - 1) // BK * BK
        kbeg, kend = split * per, min(Mpix, split * per + per)
        if kbeg >= kend:
            continue
        nk = (kend - kbeg + BK - 1) // BK
        for bid in range(nblocks):
            n0, m0 = (bid % tilesN) * BN, (bid // tilesN) * BM
            acc = np.zeros((4, 4, 4, 64, 4))
            bsum = np.zeros((256, 8))
            for ks in range(nk):
                kbase = kbeg + ks * BK
                sm = np.zeros(((BM + BN) * 16, 4))  # uint2 slots: 4 values each
                for tid in range(256):
                    pq, co = tid & 15, tid >> 4
                    cm, nn = m0 + co * 8, n0 + co * 8
                    a_ok, b_ok = cm < Cout, nn < Ktot
                    rs, ci = nn // Cin, nn % Cin
                    fr, fs = rs // S, rs % S
                    p0 = kbase + pq * 4
                    ra, rb = [], []
                    for i in range(4):
                        pp = p0 + i
                        ra.append(dyf[pp, cm:cm + 8] if (a_ok and pp < kend) else np.zeros(8))
                        ow, t = pp % Wo, pp // Wo
                        oh, b = t % Ho, t // Ho
                        ih, iw = oh * stride + fr - pad, ow * stride + fs - pad
                        ok = b_ok and pp < kend and 0 <= ih < H and 0 <= iw < W
                        rb.append(xf[(b * H + ih) * W + iw, ci:ci + 8] if ok else np.zeros(8))
                    ta, tb = transpose4x8(ra), transpose4x8(rb)
                    for j in range(8):
                        row = co * 8 + j
                        slot = row * 16 + (((pq >> 1) ^ swz128(row)) << 1) + (pq & 1)
                        sm[slot] = ta[j]
                        sm[BM * 16 + slot] = tb[j]
                    if bid % tilesN == 0:
                        bsum[tid] += sum(ra)
                sm16 = sm.reshape(-1, 8)  # view as 16-byte chunks (uint4)
                for wave in range(4):
                    wm, wn = wave >> 1, wave & 1
                    for sub in range(2):
                        fa = np.zeros((4, 64, 8))
                        fb = np.zeros((4, 64, 8))
                        for lane in range(64):
                            frow, kc = lane & 15, (lane >> 4) + 4 * sub
                            for i in range(4):
                                row = wm * 64 + i * 16 + frow
                                fa[i, lane] = sm16[row * 8 + (kc ^ swz128(row))]
                                row = wn * 64 + i * 16 + frow
                                fb[i, lane] = sm16[BM * 8 + row * 8 + (kc ^ swz128(row))]
                        for i in range(4):
                            for j in range(4):
                                acc[wave, i, j] = mfma_16x16x32(fa[i], fb[j], acc[wave, i, j])
            for wave in range(4):
                wm, wn = wave >> 1, wave & 1
                for lane in range(64):
                    for j in range(4):
                        n = n0 + wn * 64 + j * 16 + (lane & 15)
                        if n >= Ktot:
                            continue
                        for i in range(4):
                            for q in range(4):
                                m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + q
                                if m < Cout:
                                    dw[m, n] += acc[wave, i, j, lane, q]
            if bid % tilesN == 0:
                for tid in range(256):
                    pq, co = tid & 15, tid >> 4
                    if pq == 0:
                        tot = bsum[tid - 0:tid + 16].sum(axis=0)  # shfl_xor reduction over the 16 lanes sharing co
                        for q in range(8):
                            if m0 + co * 8 + q < Cout:
                                dbias[m0 + co * 8 + q] += tot[q]
    return dw.reshape(Cout, R, S, Cin), dbias


@pytest.mark.parametrize("shape", [(1, 6, 8, 16, 24, 3, 1, 1, 2), (2, 5, 7, 8, 8, 3, 1, 1, 1), (1, 12, 12, 8, 16, 1, 0, 1, 3)])
def test_wgrad_tn_index_math(shape):
    B, H, W, Cin, Cout, k, pad, stride, splitk = shape
    rng = np.random.default_rng(2)
    x = rng.standard_normal((B, H, W, Cin))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = rng.standard_normal((B, Ho, Wo, Cout))
    dw, db = emulate_wgrad_tn(dy, x, k, k, pad, stride, splitk)
    xp = np.zeros((B, H + 2 * pad, W + 2 * pad, Cin))
    xp[:, pad:pad + H, pad:pad + W] = x
    ref = np.zeros((Cout, k, k, Cin))
    for r in range(k):
        for s in range(k):
            patch = xp[:, r:r + Ho * stride:stride, s:s + Wo * stride:stride]
            ref[:, r, s] = np.einsum("bhwo,bhwi->oi", dy, patch)
    np.testing.assert_allclose(dw, ref, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(db, dy.sum(axis=(0, 1, 2)), rtol=1e-9, atol=1e-9)


# ---- LDS bank-conflict model (MI355X_MICROARCH.md, LDS table) --------------------------------
B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


def _conflict_free_b128(addr_of_lane):
    for grp in B128_GROUPS:
        slots = [(addr_of_lane(l) // 16) % 16 for l in grp]  # 64 banks x 4 B = 16 slots of 16 B
        if len(set(slots)) != len(slots):
            return False
    return True


def test_fragment_reads_are_bank_conflict_free():
    for base in (0, 16, 48, 112):
        # gemm_nt: 64-byte rows, chunk = fk ^ swz64(row)
        assert _conflict_free_b128(lambda l: ((base + (l & 15)) * 4 + ((l >> 4) ^ swz64(base + (l & 15)))) * 16)
        # gemm_tn: 128-byte rows, chunk = (fk + 4 sub) ^ swz128(row)
        for sub in (0, 1):
            assert _conflict_free_b128(
                lambda l: ((base + (l & 15)) * 8 + (((l >> 4) + 4 * sub) ^ swz128(base + (l & 15)))) * 16)
    # the unswizzled layouts are NOT conflict free (the test has teeth)
    assert not _conflict_free_b128(lambda l: ((l & 15) * 4 + (l >> 4)) * 16)


def test_staging_writes_are_bank_conflict_free():
    # gemm_nt ds_write_b128: 8 contiguous lanes per group, 32 banks (128 B)
    for g in range(0, 64, 8):
        slots = []
        for tid in range(g, g + 8):
            row, kc = tid >> 2, tid & 3
            slots.append(((row * 4 + (kc ^ swz64(row))) * 16 // 16) % 8)
        assert len(set(slots)) == 8
    # gemm_tn ds_write_b64: 16 contiguous lanes per group, 32 banks (128 B) -> 16 slots of 8 B
    for j in range(8):
        for g in range(0, 64, 16):
            slots = []
            for tid in range(g, g + 16):
                pq, co = tid & 15, tid >> 4
                row = co * 8 + j
                slots.append((row * 16 + (((pq >> 1) ^ swz128(row)) << 1) + (pq & 1)) % 16)
            assert len(set(slots)) == 16


# ---- wgrad variant 2: transposing LDS reads (csrc/gemm_tn.hip wgrad_tn_tr_kernel) ---------------
def ds_read_b64_tr_b16(lds, byte_addr):
    """Measured gfx950 semantics (tools/probe_tr.hip): in each 16-lane group, lane i supplies the
    address of 4 consecutive 16-bit elements M[i][0..3]; it receives out[j] = M[4 j + i // 4][i % 4]."""
    out = np.zeros((64, 4))
    for grp in range(4):
        M = np.array([lds[byte_addr[grp * 16 + i] // 2: byte_addr[grp * 16 + i] // 2 + 4] for i in range(16)])
        for i in range(16):
            for j in range(4):
                out[grp * 16 + i, j] = M[4 * j + i // 4, i % 4]
    return out


def swz_tr(row):
    return (row & 3) | (((row >> 3) & 1) << 2)


def emulate_wgrad_tr(dy, x, R, S, pad, stride, splitk):
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    Mpix, Ktot = B * Ho * Wo, R * S * Cin
    BM = BN = 128
    BK = 64
    TILE = BK * 16
    dw = np.zeros((Cout, Ktot))
    dyf, xf = dy.reshape(Mpix, Cout), x.reshape(-1, Cin)
    tilesN = (Ktot + BN - 1) // BN
    nblocks = ((Cout + 127) // 128) * tilesN
    for split in range(splitk):
        per = (Mpix + splitk - 1) // splitk
        per = (per + BK - 1) // BK * BK
        kbeg, kend = split * per, min(Mpix, split * per + per)
        if kbeg >= kend:
            continue
        nk = (kend - kbeg + BK - 1) // BK
        for bid in range(nblocks):
            n0, m0 = (bid % tilesN) * BN, (bid // tilesN) * BM
            acc = np.zeros((4, 4, 4, 64, 4))
            for ks in range(nk):
                kbase = kbeg + ks * BK
                sm = np.zeros((2 * TILE, 8))
                for tid in range(256):
                    cidx, prow = tid & 15, (tid >> 4) * 4
                    cm, nn = m0 + cidx * 8, n0 + cidx * 8
                    a_ok, b_ok = cm < Cout, nn < Ktot
                    rs, ci = nn // Cin, nn % Cin
                    fr, fs = rs // S, rs % S
                    for i in range(4):
                        pp = kbase + prow + i
                        ra = dyf[pp, cm:cm + 8] if (a_ok and pp < kend) else np.zeros(8)
                        ow, t = pp % Wo, pp // Wo
                        oh, b = t % Ho, t // Ho
                        ih, iw = oh * stride + fr - pad, ow * stride + fs - pad
                        ok = b_ok and pp < kend and 0 <= ih < H and 0 <= iw < W
                        rb = xf[(b * H + ih) * W + iw, ci:ci + 8] if ok else np.zeros(8)
                        row = prow + i
                        pos = row * 16 + ((((cidx >> 1) ^ swz_tr(row)) << 1) | (cidx & 1))
                        sm[pos] = ra
                        sm[TILE + pos] = rb
                lds = sm.reshape(-1)  # element-addressed view (2 bytes per element)
                for wave in range(4):
                    wm, wn = wave >> 1, wave & 1

                    def frag(tile, cb, sub):
                        halves = []
                        for rd in range(2):
                            addr = []
                            for lane in range(64):
                                i16, g = lane & 15, lane >> 4
                                row = sub * 32 + g * 8 + rd * 4 + (i16 >> 2)
                                addr.append(tile * 16 + (row * 16 + ((cb ^ swz_tr(row)) << 1)) * 16 + (i16 & 3) * 8)
                            halves.append(ds_read_b64_tr_b16(lds, addr))
                        return np.concatenate(halves, axis=1)  # [64][8]

                    for sub in range(2):
                        fa = [frag(0, wm * 4 + i, sub) for i in range(4)]
                        fb = [frag(TILE, wn * 4 + j, sub) for j in range(4)]
                        for i in range(4):
                            for j in range(4):
                                acc[wave, i, j] = mfma_16x16x32(fa[i], fb[j], acc[wave, i, j])
            for wave in range(4):
                wm, wn = wave >> 1, wave & 1
                for lane in range(64):
                    for j in range(4):
                        n = n0 + wn * 64 + j * 16 + (lane & 15)
                        if n >= Ktot:
                            continue
                        for i in range(4):
                            for q in range(4):
                                m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + q
                                if m < Cout:
                                    dw[m, n] += acc[wave, i, j, lane, q]
    return dw.reshape(Cout, R, S, Cin)


@pytest.mark.parametrize("shape", [(1, 6, 8, 16, 24, 3, 1, 1, 2), (2, 5, 7, 8, 8, 3, 1, 1, 1)])
def test_wgrad_tr_variant_index_math(shape):
    B, H, W, Cin, Cout, k, pad, stride, splitk = shape
    rng = np.random.default_rng(4)
    x = rng.standard_normal((B, H, W, Cin))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = rng.standard_normal((B, Ho, Wo, Cout))
    dw = emulate_wgrad_tr(dy, x, k, k, pad, stride, splitk)
    xp = np.zeros((B, H + 2 * pad, W + 2 * pad, Cin))
    xp[:, pad:pad + H, pad:pad + W] = x
    ref = np.zeros((Cout, k, k, Cin))
    for r in range(k):
        for s in range(k):
            patch = xp[:, r:r + Ho * stride:stride, s:s + Wo * stride:stride]
            ref[:, r, s] = np.einsum("bhwo,bhwi->oi", dy, patch)
    np.testing.assert_allclose(dw, ref, rtol=1e-9, atol=1e-9)


def test_tr_reads_and_staging_writes_bank_conflict_free():
    # ds_read_b64_tr_b16: two 32-lane groups, 64 banks; 8-byte slots must be distinct mod 256 B
    for sub in (0, 1):
        for rd in (0, 1):
            for cb in range(8):
                for half in (0, 1):
                    slots = []
                    for lane in range(half * 32, half * 32 + 32):
                        i16, g = lane & 15, lane >> 4
                        row = sub * 32 + g * 8 + rd * 4 + (i16 >> 2)
                        off = (row * 16 + ((cb ^ swz_tr(row)) << 1)) * 16 + (i16 & 3) * 8
                        slots.append((off // 8) % 32)
                    assert len(set(slots)) == 32
    # ds_write_b128 staging: groups of 8 contiguous lanes, 32 banks (128 B)
    for g0 in range(0, 256, 8):
        for i in range(4):
            slots = []
            for tid in range(g0, g0 + 8):
                cidx, row = tid & 15, (tid >> 4) * 4 + i
                pos = row * 16 + ((((cidx >> 1) ^ swz_tr(row)) << 1) | (cidx & 1))
                slots.append(pos % 8)
            assert len(set(slots)) == 8


def test_subpixel_identity_of_conv_after_nearest_upsample():
    """Groundwork for the next kernel: conv3x3(pad 1) of a nearest-x2-upsampled map equals four 2x2-tap convolutions
    of the LOW-resolution map (one per output phase) with summed weights -- 16 instead of 36 tap-MACs per low-resolution pixel.
        phase 0 along an axis: taps {-1: w0, 0: w1 + w2};   phase 1: taps {0: w0 + w1, +1: w2}"""
    import torch
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(0)
    L = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(L, scale_factor=2, mode="nearest"), w, padding=1)
    comb = {0: ((-1, [0]), (0, [1, 2])), 1: ((0, [0, 1]), (1, [2]))}       # phase -> ((offset, summed original taps), ...)
    Lp = F.pad(L, (1, 1, 1, 1))
    out = torch.zeros_like(ref)
    H, W = L.shape[2:]
    for py in (0, 1):
        for px in (0, 1):
            acc = 0
            for oy, ry in comb[py]:
                for ox, rx in comb[px]:
                    wk = w[:, :, ry][:, :, :, rx].sum((2, 3))                 # [Cout, Cin] combined weight of this tap
                    acc = acc + torch.einsum("oc,bchw->bohw", wk, Lp[:, :, 1 + oy:1 + oy + H, 1 + ox:1 + ox + W])
            out[:, :, py::2, px::2] = acc
    torch.testing.assert_close(out, ref, rtol=1e-12, atol=1e-12)


def test_subpixel_backward_identities():
    """The two adjoints the four-phase kernels need: with Wp[py][px][a][b] = sum of the original taps folded into
    phase tap (a, b),
      * input gradient at LOW resolution = sum over the 4 phases of the transposed 2x2-tap convolution of the parity sub-image
        dO[:, :, py::2, px::2] (so the gradient of the upsampled map is never formed, 16 instead of 36 tap-MACs per pixel),
      * dW[r][s] = sum of dWp[py][px][a][b] over the phase taps that (r, s) was folded into, where dWp is the plain weight gradient
        of the phase convolution (parity sub-image of dO x shifted low-resolution input)."""
    import torch
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(1)
    L = torch.randn(2, 5, 6, 8, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    dO = torch.randn(2, 4, 12, 16, generator=g, dtype=torch.float64)
    F.conv2d(F.interpolate(L, scale_factor=2, mode="nearest"), w, padding=1).backward(dO)
    comb = {0: ((-1, [0]), (0, [1, 2])), 1: ((0, [0, 1]), (1, [2]))}
    H, W = L.shape[2:]
    Ld = L.detach()
    Lp = F.pad(Ld, (1, 1, 1, 1))
    dL = torch.zeros(2, 5, H + 2, W + 2, dtype=torch.float64)       # padded accumulator, the border is dropped
    dW = torch.zeros_like(w)
    for py in (0, 1):
        for px in (0, 1):
            sub = dO[:, :, py::2, px::2]                              # [B, Cout, H, W]: the phase's output pixels
            for oy, ry in comb[py]:
                for ox, rx in comb[px]:
                    wk = w.detach()[:, :, ry][:, :, :, rx].sum((2, 3))                  # folded tap [Cout, Cin]
                    dL[:, :, 1 + oy:1 + oy + H, 1 + ox:1 + ox + W] += torch.einsum("oc,bohw->bchw", wk, sub)
                    dwk = torch.einsum("bohw,bchw->oc", sub, Lp[:, :, 1 + oy:1 + oy + H, 1 + ox:1 + ox + W])
                    for r in ry:                                      # unfold: every original tap of the fold receives dWp
                        for s in rx:
                            dW[:, :, r, s] += dwk
    torch.testing.assert_close(dL[:, :, 1:-1, 1:-1], L.grad, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(dW, w.grad, rtol=1e-12, atol=1e-12)


def test_subpixel_input_gradient_is_a_4x4_stride2_convolution():
    """The gather form of the sub-pixel input gradient: the gradient w.r.t. the LOW-resolution map of
    conv3x3(Upsample(L)) is ONE 4x4, stride-2, pad-1 convolution of dO whose taps along an axis are [w2, w1 + w2, w0 + w1, w0]
    (rows 2i-1 .. 2i+2 of dO feed low-resolution row i): the four phase planes of dO accumulate into the same output tile, which is
    what the halo kernel's K loop over (phase, tap) pairs would execute -- 16 tap-MACs per low-resolution pixel instead of 36."""
    import torch
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(2)
    L = torch.randn(2, 5, 6, 8, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    dO = torch.randn(2, 4, 12, 16, generator=g, dtype=torch.float64)
    F.conv2d(F.interpolate(L, scale_factor=2, mode="nearest"), w, padding=1).backward(dO)
    fold = [[2], [1, 2], [0, 1], [0]]                       # 4x4 tap u (row 2i-1+u of dO) <- summed original taps
    K = torch.zeros(5, 4, 4, 4, dtype=torch.float64)        # [Cin (output of this conv)][Cout (its input)][u][v]
    for u in range(4):
        for v in range(4):
            K[:, :, u, v] = w[:, :, fold[u]][:, :, :, fold[v]].sum((2, 3)).t()
    dL = F.conv2d(dO, K, stride=2, padding=1)
    torch.testing.assert_close(dL, L.grad, rtol=1e-12, atol=1e-12)

