"""CPU check of the LDS layouts behind the 16 x 16 x 32 main loop of csrc/gemm.hip (gemm_pers_kernel, M16): the index formulas of the kernel are
restated here and checked for (a) bank-conflict freedom of every fragment read per hardware lane group, (b) agreement between where the LDS-DMA
puts a source chunk (the swizzle lives on the SOURCE address, the LDS side is lane-linear) and where the fragment reads look for it.  The
kernels themselves are checked against fp32 GEMMs on the GPU (tests/test_kernels_gpu.py)."""

# ds_read_b128 is served in four 16-lane groups (MI355X_MICROARCH.md, LDS table); a group is conflict-free when its lanes hit 16 distinct 16-byte slots mod 256 B
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def kc_swz16(row):                 # gemm.hip kc_swz<true>
    return ((row >> 3) & 1) * 3


def frag16_kc_addr(rbase, lane):   # gemm.hip frag16_kc: 64-byte rows (k-unit 32), chunk = lane >> 4
    row = rbase + (lane & 15)
    return row * 64 + (((lane >> 4) ^ kc_swz16(row)) << 4)


def test_kc_fragment_reads_conflict_free_and_match_dma():
    for rbase in range(0, 512, 16):
        for grp in B128_GROUPS:
            slots = [(frag16_kc_addr(rbase, l) // 16) % 16 for l in grp]
            assert len(set(slots)) == 16, (rbase, grp)
    # DMA placement (piece_ptr_rt<true, M16>): LDS chunk q of the image holds source chunk (q & 3) ^ swz(row) of row q >> 2
    for q in range(512 * 4):
        row, c = q >> 2, (q & 3) ^ kc_swz16(q >> 2)
        lane = (row & 15) + 16 * c                 # the lane that wants k-chunk c of this row
        assert frag16_kc_addr(row & ~15, lane) == q * 16


def ks_addr(rbase, lane, rows_log2, e):            # gemm.hip frag16<false>: the two transpose reads (e = 0, 1) of a k-strided image
    gg, tt = lane >> 4, lane & 15
    kr = 8 * gg + (tt >> 2) + 4 * e
    col = rbase + (tt & 3) * 4
    blk, inblk = col >> 5, ((col & 31) * 2) ^ ((gg & 1) << 5)
    return kr, col, (kr << (rows_log2 + 1)) + ((blk ^ (kr & 3)) << 6) + inblk


def test_k_strided_fragment_reads_conflict_free_and_match_dma():
    for rows_log2 in (7, 8, 9):                    # half items, full tiles, paired tiles
        sh = rows_log2 - 3
        pos = {}
        for q in range(32 << sh):                  # piece_ptr_rt<false, M16>: LDS chunk q <- source chunk c of k-row kr
            kr, cl = q >> sh, q & ((1 << sh) - 1)
            c = (((cl >> 2) ^ (kr & 3)) << 2) | ((cl & 3) ^ (((kr >> 3) & 1) << 1))
            pos[(kr, c)] = q * 16
        for rbase in range(0, 1 << rows_log2, 16):
            for e in (0, 1):
                for half in (0, 1):                # ds_read_b64_tr_b16: two 32-lane groups, 8 bytes = 2 banks per lane, 64 banks
                    banks = []
                    for lane in range(32 * half, 32 * half + 32):
                        kr, col, a = ks_addr(rbase, lane, rows_log2, e)
                        banks += [(a // 4) % 64, (a // 4 + 1) % 64]
                        assert a == pos[(kr, col // 8)] + (col % 8) * 2
                    assert sorted(banks) == list(range(64)), (rows_log2, rbase, e, half)
