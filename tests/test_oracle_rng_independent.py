"""The keyed random numbers against a generator that shares nothing with them.

The GPU library and the CPU checker draw the same keyed hash (rng.h, restated in oracle.cpp), so every GPU-vs-checker
test is blind to a defect of that hash: both sides would be wrong together.  Here the checker renders the two media
scenes - whose path trees have the most nodes per pixel, and the most key collisions - once with the keyed numbers and
twice with a std::mt19937_64 per rendering thread that is simply drawn from in call order (the reference's own scheme,
src/randomnumbers.cpp:30-41, without its race).  If the hash biased the estimator, the keyed image would sit further
from the sequential ones than they sit from each other.

Tolerances (stated here): relRMSE(keyed, sequential) <= 1.5 x relRMSE(sequential A, sequential B), and the image means
of the keyed render within 1 % of the mean of the two sequential renders, per channel.
Measured when the test was written (1024 spp): fog 0.0211 / 0.0200 against 0.0194, means within 0.1 %; milk 0.0297 / 0.0302
against 0.0323, means within 0.2 %.
"""
import numpy as np
import pytest

import helpers
import oracle_api
from pathtracer_amd import scenes


def _check(sc, spp, threads=8):
    o = oracle_api.OracleScene(sc)
    try:
        keyed = o.render(spp, seed=11, threads=threads)
        seq_a = o.render(spp, seed=21, threads=threads, sequential=True)
        seq_b = o.render(spp, seed=22, threads=threads, sequential=True)
    finally:
        o.close()
    noise = helpers.rel_rmse(seq_a, seq_b)
    ra, rb = helpers.rel_rmse(keyed, seq_a), helpers.rel_rmse(keyed, seq_b)
    print(f"seq-vs-seq {noise:.4f}; keyed-vs-seq {ra:.4f} {rb:.4f}; means keyed {keyed.mean((0, 1))} "
          f"seq {0.5 * (seq_a + seq_b).mean((0, 1))}")
    assert noise > 0.0
    assert ra <= 1.5 * noise and rb <= 1.5 * noise
    np.testing.assert_allclose(keyed.mean((0, 1)), 0.5 * (seq_a.mean((0, 1)) + seq_b.mean((0, 1))), rtol=1e-2)
    # the two sequential renders are different estimates (the mode is really on), and not wildly so
    assert not np.array_equal(seq_a, seq_b)


def test_sequential_mode_depends_on_seed_only_through_the_generator():
    """Same seed and thread count: the sequential mode reproduces itself; the keyed mode does not depend on threads."""
    sc = scenes.cornell(res=(12, 12), spheres=False)
    o = oracle_api.OracleScene(sc)
    a = o.render(8, seed=3, threads=2, sequential=True)
    b = o.render(8, seed=3, threads=2, sequential=True)
    c = o.render(8, seed=4, threads=2, sequential=True)
    k1 = o.render(8, seed=3, threads=1)
    k4 = o.render(8, seed=3, threads=4)
    o.close()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.array_equal(k1, k4) and not np.array_equal(k1, a)


def test_fog_keyed_numbers_against_an_independent_generator():
    # 1.4e3 closest-hit queries per path: 12 x 8 pixels x 1024 spp is 1.4e8 queries per render
    _check(scenes.fog(res=(12, 8)), 1024)


def test_milk_keyed_numbers_against_an_independent_generator():
    _check(scenes.dragon("milk", res=(24, 14), n_along=120, n_around=40), 1024)
