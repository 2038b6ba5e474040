"""The probe stream (np.random.randint(2, size=n) of utils.py:213-216,255-258):
host MT19937 + GF(2) jump-ahead against NumPy (CPU), and the device generator
(k_mt_jump / k_mt_generate through sw_probes_generate) against NumPy bit for bit (GPU).
Integer work: every comparison is exact equality."""
import numpy as np
import pytest

from deflatedmlmc_schwinger_amd import matrix
from deflatedmlmc_schwinger_amd.engine import Engine, ProbeStream

Z4_CODES = np.array([1, 2, -1, -2], dtype=np.int8)


# ---------------------------------------------------------------------------------------------
# host (not gpu)
# ---------------------------------------------------------------------------------------------
def test_host_stream_equals_numpy_legacy_randint():
    np.random.seed(123456)
    ref = np.random.randint(2, size=(3, 1000))
    g = ProbeStream(123456)
    assert np.array_equal(g.rademacher(3, 1000), (2 * ref - 1).astype(np.int8))
    # SURVEY F10 known answers: first raw outputs of MT19937(123456)
    assert ProbeStream(123456).raw(4).tolist() == [545331265, 2211535594, 4152021490, 3857419313]
    np.random.seed(5)
    q = np.random.randint(4, size=777)
    assert np.array_equal(ProbeStream(5).z4(1, 777)[0], Z4_CODES[q])


@pytest.mark.parametrize("ndraws", [1, 5, 623, 624, 625, 1247, 1248, 19937, 100003, 8388608,
                                    3 * 8388608 + 17])
def test_jump_equals_sequential_walk(ndraws):
    a, b = ProbeStream(123456), ProbeStream(123456)
    a.raw(11)
    b.raw(11)                      # start inside a state block
    a.skip(ndraws)
    b.jump(ndraws)
    assert np.array_equal(a.raw(1500), b.raw(1500))


def test_jump_from_fresh_seed_and_numpy_roundtrip():
    # a freshly seeded state has not produced its first block yet (pos == 624)
    np.random.seed(99)
    ref = np.random.randint(2, size=50000 + 300)
    g = ProbeStream(99)
    g.jump(50000)
    assert np.array_equal(g.rademacher(1, 300)[0], 2 * ref[50000:] - 1)
    # NumPy state in, jump, NumPy state out: the global stream continues where NumPy would be
    np.random.seed(7)
    np.random.randint(2, size=1234)
    g = ProbeStream.from_numpy_state()
    expect = np.random.randint(1 << 30, size=4000 + 5)
    g.jump(4000)
    np.random.set_state(g.numpy_state())
    assert np.array_equal(np.random.randint(1 << 30, size=5), expect[4000:])


def test_jump_composes_at_production_distances():
    """size-independent property of the GF(2) jump-ahead, at stream distances no sequential walk could
    check (a rank's block of round r sits at (r * world + rank) * batch * n draws: 1e12 and beyond):
    jump(a) followed by jump(b) lands exactly where jump(a + b) does, from inside a state block."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=12, deadline=None)
    @given(st.integers(min_value=0, max_value=1 << 44), st.integers(min_value=0, max_value=1 << 44),
           st.integers(min_value=0, max_value=1300))
    def check(a, b, start):
        x, y = ProbeStream(123456), ProbeStream(123456)
        x.raw(start)
        y.raw(start)
        x.jump(a)
        x.jump(b)
        y.jump(a + b)
        assert np.array_equal(x.raw(700), y.raw(700))

    check()


def test_window_is_the_raw_word_sequence():
    g = ProbeStream(2024)
    g.raw(700)                       # position 700: window spans two state blocks
    w = g.window()
    nxt = g.copy()
    key = np.asarray(nxt.numpy_state()[1])
    pos = nxt.numpy_state()[2]
    assert np.array_equal(w[:624 - pos], key[pos:])
    # tempering the window's words gives the next outputs
    y = w.astype(np.uint32).copy()
    y ^= (y >> 11)
    y ^= (y << 7) & np.uint32(0x9d2c5680)
    y ^= (y << 15) & np.uint32(0xefc60000)
    y ^= (y >> 18)
    assert np.array_equal(y, g.raw(624))


# ---------------------------------------------------------------------------------------------
# device (gpu)
# ---------------------------------------------------------------------------------------------
def _engine_with_lattice(L):
    U1, U2 = matrix.synthetic_links(L, 0.3, 11)
    eng = Engine(0)
    eng.hier_begin(0, 1)
    eng.set_lattice(0, L, 0.1, U1, U2)
    eng.hier_end(0)
    return eng


@pytest.mark.gpu
def test_device_probes_bit_exact_small_lattice():
    """n = 512: probes straddle the 624-word state blocks everywhere; 300 probes = three
    generator segments (two jump polynomials)."""
    eng = _engine_with_lattice(16)
    n = 512
    np.random.seed(123456)
    ref = (2 * np.random.randint(2, size=(300, n)) - 1).astype(np.int8)
    eng.stream_set(ProbeStream(123456).window())
    eng.probes_generate(0, 0, 300, 0)
    assert np.array_equal(eng.probes_fetch(0), ref)
    # a rank offset: probes 37..36+50 of the same stream, position not a multiple of 624
    eng.probes_generate(1, 0, 50, 37 * n)
    assert np.array_equal(eng.probes_fetch(1), ref[37:87])
    # moving backwards restarts from the stored window
    eng.probes_generate(2, 0, 3, 0)
    assert np.array_equal(eng.probes_fetch(2), ref[:3])
    # position exactly on and next to a state-block boundary
    for pos in (624, 623, 625, 5 * 624):
        np.random.seed(123456)
        np.random.randint(2, size=pos)
        exp = (2 * np.random.randint(2, size=(2, n)) - 1).astype(np.int8)
        eng.probes_generate(3, 0, 2, pos)
        assert np.array_equal(eng.probes_fetch(3), exp), pos
    eng.close()


@pytest.mark.gpu
def test_device_probes_bit_exact_schwinger128_batch_and_rank_offset():
    """The bench's configuration: 256 probes of n = 32768 (128 segments) at stream position 0 and
    at the position of (rank 3, stream 1) of an 8-rank, 2-stream round."""
    eng = _engine_with_lattice(128)
    n, nb = 32768, 256
    eng.stream_set(ProbeStream(123456).window())
    eng.probes_generate(0, 0, nb, 0)
    np.random.seed(123456)
    ref = (2 * np.random.randint(2, size=(nb, n)) - 1).astype(np.int8)
    got = eng.probes_fetch(0)
    assert np.array_equal(got, ref)
    assert int(got[0].astype(np.int64).sum()) == 144          # probe 0 entry sum
    first = (3 * 2 + 1) * nb
    eng.probes_generate(1, 0, 5, first * n)
    host = ProbeStream(123456)
    host.skip(first * n)
    assert np.array_equal(eng.probes_fetch(1), host.rademacher(5, n))
    # the next round of that (rank, stream): a second jump by the same distance
    nxt = first + 8 * 2 * nb
    eng.probes_generate(1, 0, 5, nxt * n)
    host = ProbeStream(123456)
    host.skip(nxt * n)
    assert np.array_equal(eng.probes_fetch(1), host.rademacher(5, n))
    eng.close()


@pytest.mark.gpu
def test_device_probes_z4_and_numpy_midstream():
    eng = _engine_with_lattice(16)
    n = 512
    np.random.seed(31)
    np.random.randint(2, size=1000)                          # somewhere inside the stream
    eng.stream_set(ProbeStream.from_numpy_state().window())
    ref = Z4_CODES[np.random.randint(4, size=(9, n))]
    eng.probes_generate(0, 0, 9, 0, kind="z4")
    assert np.array_equal(eng.probes_fetch(0), ref)
    eng.close()
