"""The counter-based randomness restated in the oracle: known answers and distributional checks."""
import numpy as np


def test_philox_known_answers(O):
    # Random123 kat_vectors, philox4x32-10
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_shock_matrix_is_standard_normal(O):
    Z = O.gen_Z(1234, 2, 10000)
    assert Z.shape == (2, 10000) and np.isfinite(Z).all()
    assert abs(Z.mean()) < 0.03 and abs(Z.std() - 1) < 0.03
    assert abs(np.corrcoef(Z)[0, 1]) < 0.05
    assert np.array_equal(Z, O.gen_Z(1234, 2, 10000))      # a function of (seed, k, s) only
    assert not np.array_equal(Z, O.gen_Z(1235, 2, 10000))
    assert np.array_equal(O.gen_Z(1234, 1, 500), Z[:1, :500])


def test_exchange_pairs_are_a_sample_without_replacement(O):
    # sample(props, N, replace=false), AlgoBGP.jl:653-656
    for Ng in (2, 3, 4, 7, 64, 1000, 4096):
        K = Ng - 1 if Ng < 3 else Ng
        p = O.gen_pairs(12, 5, Ng)
        assert p.shape == (K, 2)
        assert (p[:, 0] < p[:, 1]).all() and p.min() >= 0 and p.max() < Ng
        lin = p[:, 1].astype(np.int64) * (p[:, 1] - 1) // 2 + p[:, 0]
        assert len(np.unique(lin)) == K  # distinct pairs
    # N=3: all three pairs, in an order that changes with the iteration
    orders = {tuple(map(tuple, O.gen_pairs(12, t, 3))) for t in range(2, 40)}
    assert all(sorted(o) == [(0, 1), (0, 2), (1, 2)] for o in orders) and len(orders) > 1


def test_exchange_pairs_cover_all_pairs_uniformly(O):
    Ng, M = 12, 66
    counts = np.zeros(M)
    for t in range(2, 2002):
        p = O.gen_pairs(99, t, Ng)
        counts[p[:, 1] * (p[:, 1] - 1) // 2 + p[:, 0]] += 1
    expected = 2000 * Ng / M
    assert counts.min() > 0.8 * expected and counts.max() < 1.2 * expected
