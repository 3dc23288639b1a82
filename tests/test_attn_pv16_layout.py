"""Lane-level emulation of the 16x16x32 second products of csrc/attn.hip (pack_xy, tr16_addr / trfrag16, mma16, store_rows16).

The kernel's index formulas are restated here in numpy and run through an emulation of the hardware semantics they rely on
(v_mfma_f32_32x32x16 accumulator layout, v_mfma_f32_16x16x32 operand / accumulator layout, ds_read_b64_tr_b16, v_permlane16_swap - the
first three measured by probe/probe.hip, the last by probe/perm16.hip); the result must equal the plain matrix product.  A CPU check of the
derivation - the kernels themselves are checked against fp32 attention on the GPU (tests/test_kernels_gpu.py)."""
import numpy as np

ROWB = 192          # bytes per tile row: 12 chunks of 16 B (9 data + 3 pad)
DH = 72


def soff(r, c):     # attn.hip soff(): byte offset of chunk c of row r
    return r * ROWB + ((c ^ ((r >> 2) & 3)) << 4)


def build_tile(X):  # X [64][72] -> LDS image as an array of 16-bit elements (stored as float for the emulation); pads zero
    img = np.zeros(64 * ROWB // 2)
    for r in range(64):
        for c in range(12):
            for e in range(8):
                d = c * 8 + e
                img[(soff(r, c) >> 1) + e] = X[r, d] if d < X.shape[1] else 0.0
    return img


def tr_read(img, addr):            # addr[64] byte addresses -> [64][4]
    out = np.zeros((64, 4))
    for G in range(4):
        E = np.array([[img[(addr[16 * G + s] >> 1) + e] for e in range(4)] for s in range(16)])
        for t in range(16):
            for j in range(4):
                out[16 * G + t, j] = E[4 * j + (t >> 2), t & 3]
    return out


def tr16_addr(lane):               # attn.hip tr16_addr(): tb[e][par]
    gg, tt = lane >> 4, lane & 15
    x = (tt & 3) >> 1
    tb = [[0, 0], [0, 0]]
    for e in range(2):
        row = 8 * (gg & 1) + 4 * (gg >> 1) + (tt >> 2) + 16 * e
        sw = (row >> 2) & 3
        for par in range(2):
            tb[e][par] = row * ROWB + (((2 * par + x) ^ sw) << 4) + (tt & 1) * 8
    return tb


def trfrag16(img, t, sub):         # -> [64 lanes][8]
    off = sub * 32 * ROWB + (t >> 1) * 64
    a0 = [tr16_addr(l)[0][t & 1] + off for l in range(64)]
    a1 = [tr16_addr(l)[1][t & 1] + off for l in range(64)]
    return np.concatenate([tr_read(img, a0), tr_read(img, a1)], axis=1)


def permlane16_swap(a, b):         # odd 16-lane rows of a <-> even rows of b; returns (new a, new b)
    a, b = a.copy(), b.copy()
    for p in range(2):
        lo, hi = slice(32 * p, 32 * p + 16), slice(32 * p + 16, 32 * p + 32)
        tmp = a[hi].copy()
        a[hi] = b[lo]
        b[lo] = tmp
    return a, b


def pack_xy(v):                    # v [64 lanes][16 accumulator values] -> X, Y [64][8]
    a = np.concatenate([v[:, 0:4], v[:, 8:12]], axis=1)
    b = np.concatenate([v[:, 4:8], v[:, 12:16]], axis=1)
    X, Y = np.zeros_like(a), np.zeros_like(b)
    for w in range(4):             # dword w = elements 2w, 2w + 1
        xa, yb = permlane16_swap(a[:, 2 * w:2 * w + 2], b[:, 2 * w:2 * w + 2])
        X[:, 2 * w:2 * w + 2], Y[:, 2 * w:2 * w + 2] = xa, yb
    return X, Y


def mfma16(a, b, c):               # a, b [64][8], c [64][4]
    A, B = np.zeros((16, 32)), np.zeros((32, 16))
    for l in range(64):
        for j in range(8):
            A[l & 15, 8 * (l >> 4) + j] = a[l, j]
            B[8 * (l >> 4) + j, l & 15] = b[l, j]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for g in range(4):
            out[l, g] += D[4 * (l >> 4) + g, l & 15]
    return out


def acc32_layout(M):               # M [32 rows][32 cols] -> the 32x32x16 accumulator registers [64 lanes][16]
    v = np.zeros((64, 16))
    for l in range(64):
        for g in range(16):
            v[l, g] = M[(g & 3) + 8 * (g >> 2) + 4 * (l >> 5), l & 31]
    return v


def test_second_product_layout():
    rng = np.random.default_rng(0)
    V = rng.standard_normal((64, DH))                  # row-major [row][d] tile (V, K, Q or dO)
    P = rng.standard_normal((64, 32))                  # P^T[row][column]: 64 rows (keys) x the wave's 32 columns (queries)
    img = build_tile(V)
    acc = np.zeros((5, 2, 64, 4))
    for sub in range(2):
        X, Y = pack_xy(acc32_layout(P[32 * sub:32 * sub + 32]))
        for t in range(5):
            af = trfrag16(img, t, sub)
            acc[t, 0] = mfma16(af, X, acc[t, 0])
            acc[t, 1] = mfma16(af, Y, acc[t, 1])
    # store_rows16(): lane (R, c) holds out[column c + 16 half][d = 16 t + 4 R + g]
    out = np.zeros((32, 80))
    for t in range(5):
        for half in range(2):
            for l in range(64):
                for g in range(4):
                    out[(l & 15) + 16 * half, 16 * t + 4 * (l >> 4) + g] = acc[t, half, l, g]
    ref = P.T @ V                                      # [32 columns][72]
    assert np.allclose(out[:, :DH], ref, atol=1e-12)
    assert np.allclose(out[:, DH:], 0.0)               # pad columns of the tile are zero


def test_tr16_reads_are_bank_conflict_free():
    # ds_read_b64_tr_b16 is served in two 32-lane groups; bank = (addr / 4) % 64, 8 bytes = 2 banks per lane: each group must cover 64 banks once
    for t in range(5):
        for e in range(2):
            for grp in range(2):
                banks = []
                for l in range(32 * grp, 32 * grp + 32):
                    a = tr16_addr(l)[e][t & 1] + (t >> 1) * 64
                    banks += [(a // 4) % 64, (a // 4 + 1) % 64]
                assert sorted(banks) == list(range(64)), (t, e, grp)


def block_coords(blk, T, nx, H):    # attn.hip block_coords(): flat workgroup index -> (row block, head, batch)
    xq, xr, xcd = T >> 3, T & 7, blk & 7
    v = xcd * xq + min(xcd, xr) + (blk >> 3)
    return v % nx, (v // nx) % H, v // nx // H


def test_block_order_is_a_bijection_and_keeps_heads_on_one_xcd():
    for nx, H, B in [(32, 16, 16), (16, 16, 16), (3, 16, 16), (2, 3, 2), (1, 3, 2), (1, 2, 1), (64, 16, 4), (5, 7, 3)]:
        T = nx * H * B
        seen = {}
        for blk in range(T):
            c = block_coords(blk, T, nx, H)
            assert c not in seen and c[0] < nx and c[1] < H and c[2] < B
            seen[c] = blk & 7                       # the XCD the block runs on (workgroup i -> XCD i % 8)
        assert len(seen) == T
        if T % 8 == 0 and (T // 8) % nx == 0:       # whole heads per XCD: every block of a (batch, head) on the same XCD
            for h in range(H):
                for b in range(B):
                    assert len({seen[(x, h, b)] for x in range(nx)}) == 1
