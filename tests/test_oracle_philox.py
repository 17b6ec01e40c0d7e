"""Known-answer tests for the Philox4x32-10 oracle (Random123 kat_vectors: philox4x32 10 rounds)."""
import numpy as np

from oracle import philox_ref


def test_philox_kat():
    # Random123 known answers: counter, key -> output
    kat = [
        ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
         (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, exp in kat:
        out = philox_ref.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(v) for v in out[0]) == exp


def test_randn_moments():
    z = philox_ref.randn(1 << 16, seed=1234, offset=7)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    assert np.isfinite(z).all()
