"""Host-side model of every address the fused GroupNorm tail (csrc/gn_tail.h) and the epilogues of the two F(4x4,3x3) kernels
(csrc/conv_wino4.hip, csrc/conv_wino4h.hip) form, checked against the true extents of the buffers for the parametrisations of
tests/test_ops_gpu.py::test_conv_groupnorm_finalize_fused_equals_separate_launch -- and the arithmetic of the r03 abort: the
partials row address rebuilt from two readfirstlane halves with the low half still a signed int."""
import numpy as np
import pytest

from instancediff_amd import _lib

# (B, Cin, Cout, H, W, groups, film, kernel): kernel 4 = half-patch items (8x32), 3 = 16x32 items
CASES = [
    (16, 64, 64, 64, 64, 8, True, 4), (16, 64, 64, 64, 64, 8, True, 3), (3, 64, 64, 128, 128, 8, False, 3), (2, 80, 128, 32, 64, 8, True, 4),
    (5, 128, 256, 32, 32, 8, True, 4), (1, 16, 32, 28, 28, 8, False, 4), (1, 16, 32, 28, 28, 8, False, 3), (2, 64, 80, 20, 40, 8, True, 3),
]


def partial_rows_written(B, Cout, H, W, kernel, ntiles, tiles_x):
    """(row index into stats[B*ntiles][Cout][2], first channel, channel count) of every gn_store_partial of a launch"""
    TH = 8 if kernel == 4 else 16
    ncob = (Cout + 63) // 64
    rows = []
    for b in range(B):
        for py in range((H + TH - 1) // TH):
            for px in range(tiles_x):
                for cob in range(ncob):
                    y0, x0, co0 = py * TH, px * 32, cob * 64
                    for cb in range(4):
                        if co0 + cb * 16 >= Cout:      # the uniform skip of a 16-channel block beyond a partial Cout
                            continue
                        for tblk in range(1 if kernel == 4 else 2):
                            ty0 = y0 + 8 * tblk
                            if ty0 >= H:               # conv_wino4.hip: want_stats = ... && ty0 < a.Hout
                                continue
                            cell = (ty0 >> 3) * tiles_x + (x0 >> 5)
                            rows.append((b * ntiles + cell, co0 + cb * 16, 16))
    return rows


@pytest.mark.parametrize("B,Cin,Cout,H,W,G,film,kernel", CASES)
def test_epilogue_and_tail_addresses_stay_inside_their_buffers(B, Cin, Cout, H, W, G, film, kernel):
    lib = _lib.load()
    ntiles = lib.idiff_conv2d_num_tiles(H, W)
    tiles_x = (W + 31) // 32
    assert ntiles == tiles_x * ((H + 7) // 8)
    # --- producers: every (row, channel) of stats[B][ntiles][Cout][2] is written exactly once, none outside
    cover = np.zeros((B * ntiles, Cout), dtype=np.int32)
    for row, c0, n in partial_rows_written(B, Cout, H, W, kernel, ntiles, tiles_x):
        assert 0 <= row < B * ntiles and c0 + n <= Cout
        cover[row, c0:c0 + n] += 1
    assert (cover == 1).all()
    # --- the tail's reads of one (sample, group) pair: byte offsets relative to the pair's base, against the resource's extent
    cpg = Cout // G
    tstep = 256 // cpg
    extent = ((ntiles - 1) * Cout + cpg) * 8
    stats_bytes = B * ntiles * Cout * 8
    for b in (0, B - 1):
        for g in (0, G - 1):
            base = (b * ntiles * Cout + g * cpg) * 8
            seen = set()
            for tid in range(256):
                cl, tph = tid % cpg, tid // cpg
                if tph >= tstep:
                    continue
                for tt in range(tph, ntiles, tstep):
                    off = (tt * Cout + cl) * 8
                    assert 0 <= off and off + 8 <= extent and base + off + 8 <= stats_bytes
                    seen.add((tt, cl))
            assert len(seen) == ntiles * cpg     # every partial of the group, once
            # --- the tail's writes and FiLM reads
            film_ld = 2 * Cout + 8
            for i in range(cpg):
                c = g * cpg + i
                assert b * Cout + c < B * Cout                       # out_a / out_b [B][Cout]
                if film:
                    assert b * film_ld + Cout + c < (B - 1) * film_ld + 2 * Cout   # last row of a row-strided [B][2 Cout] view
            assert (b * G + g) * 2 + 1 < B * G * 2                   # mean_rstd [B][groups][2]


def test_pointer_rebuilt_from_a_signed_low_half_breaks_when_bit_31_is_set():
    """readfirstlane returns int.  `(u64)hi << 32 | lo` with lo an int converts lo to 64 bits by SIGN extension: the r03 form.  The
    resource's base is then (value & 0xffff_ffff_ffff) -- 0xffff.... instead of the row's address."""
    def rebuild_r03(addr):
        lo = np.int32(np.uint32(addr & 0xffffffff))                 # the builtin's int result
        hi = np.int32(np.uint32(addr >> 32))
        v = (np.uint64(np.uint32(hi)) << np.uint64(32)) | np.uint64(np.int64(lo))   # usual arithmetic conversions: int -> u64
        return int(v) & 0xffffffffffff                                # 48-bit base of the buffer resource
    def rebuild_fixed(addr):
        lo, hi = np.uint32(addr & 0xffffffff), np.uint32(addr >> 32)
        return int((np.uint64(hi) << np.uint64(32)) | np.uint64(lo)) & 0xffffffffffff
    low_half_clear, low_half_set = 0x7f2a_1234_5600, 0x7f2a_9234_5600
    assert rebuild_r03(low_half_clear) == low_half_clear
    assert rebuild_r03(low_half_set) == 0xffff_9234_5600 != low_half_set       # upper 16 bits of the base overwritten
    assert rebuild_fixed(low_half_clear) == low_half_clear and rebuild_fixed(low_half_set) == low_half_set
