"""Pins the RNG layer of the oracle: Philox4x32-10 against the Random123 known-answer
vectors, the MicroURNG word order, and the uniform->normal/gamma transforms against
the real libstdc++ distributions the reference calls (c++/mvnormal.cpp:42,68)."""
import json
import os

import numpy as np

from oracle.oracle import PinLibstdcxx
from tests.util import GOLDEN


def test_philox_kat(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "philox_kat.json")))["philox4x32_10"]
    assert len(kat) == 3
    for v in kat:
        out = oracle.philox([int(x, 16) for x in v["ctr"]], [int(x, 16) for x in v["key"]])
        assert [format(int(x), "08x") for x in out] == v["out"]


def test_microurng_word_order(oracle):
    # stream c hands out block {c,0,0,n} key {42,0} as w3,w2,w1,w0 (Random123 MicroURNG.hpp)
    for c in (0, 5, 0xFFFFFFFF):
        words = oracle.words(c, 12)
        for n in range(3):
            blk = oracle.philox([c, 0, 0, n], [42, 0])
            assert list(words[4 * n:4 * n + 4]) == [blk[3], blk[2], blk[1], blk[0]]


def test_randn_matches_libstdcxx_bit_for_bit(oracle):
    pin = PinLibstdcxx()
    for c in (0, 1, 32, 2 ** 32 - 1, 31337, 64 * 943 * 20):
        assert np.array_equal(oracle.randn(c, 2000), pin.randn(c, 2000))


def test_gamma_matches_libstdcxx_bit_for_bit(oracle):
    pin = PinLibstdcxx()
    # alpha = 0.5*(df-i) as in WishartUnitChol (c++/mvnormal.cpp:68); includes alpha < 1 and large alpha
    alphas = np.concatenate([0.5 * np.arange(1, 200), [0.25, 0.5, 0.75, 487.5, 5e6 + 0.5]])
    for c in (0, 3, 19):
        g1, z1 = oracle.gamma_stream(c, alphas)
        g2, z2 = pin.gamma_stream(c, alphas)
        assert np.array_equal(g1, g2) and np.array_equal(z1, z2)


def test_randn_moments(oracle):
    z = oracle.randn(7, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    # fourth moment of a standard normal is 3
    assert abs((z ** 4).mean() - 3.0) < 0.1
