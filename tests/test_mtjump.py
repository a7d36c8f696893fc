"""CPU: the jump-ahead polynomials of numpy's legacy MT19937 (csrc/gsage_mtjump.hip, host side) against numpy itself.
The device side -- many workgroups consuming ONE np.random stream -- is tests/test_gpu_round5.py."""
import ctypes

import numpy as np
import pytest

from conftest import pkg


def _table():
    L = pkg()._native.lib()
    words = int(L.gsage_mt_jump_table_words())
    tab = np.zeros(words, dtype=np.uint64)
    assert L.gsage_mt_jump_table(tab.ctypes.data, words) == 0
    return L, tab.reshape(128, 312)


def _jump(L, state, poly):
    out = np.zeros(624, dtype=np.uint32)
    assert L.gsage_mt_jump_host(state.ctypes.data, np.ascontiguousarray(poly).ctypes.data, out.ctypes.data) == 0
    return out


def _same_state(a, b):
    """624-word windows that describe the same generator state: word 0 contributes its top bit only"""
    return np.array_equal(a[1:], b[1:]) and (int(a[0]) >> 31) == (int(b[0]) >> 31)


def _aligned(seed, burn):
    rs = np.random.RandomState(seed)
    rs.randint(0, 2 ** 31, size=burn)
    pos = rs.get_state()[2]
    rs.randint(0, 2 ** 32, size=624 - pos, dtype=np.uint32)           # finish the block: position 624
    assert rs.get_state()[2] == 624
    return rs, np.asarray(rs.get_state()[1], dtype=np.uint32).copy()


@pytest.mark.parametrize("seed,entry,refills", [(123, 1, 64), (7, 5, 5 * 64), (42, 63, 63 * 64), (9, 64 + 1, 64 * 64),
                                                (5, 64 + 3, 3 * 64 * 64)])
def test_jump_polynomials_equal_numpy_stepped_that_far(seed, entry, refills):
    """table[b] = x^(b * 64 refills), table[64 + a] = x^(a * 64 * 64 refills) (mod MT19937's characteristic polynomial):
    applied to a state they give the state numpy reaches by drawing that many words."""
    L, tab = _table()
    rs, key0 = _aligned(seed, 1000 + seed)
    rs.randint(0, 2 ** 32, size=refills * 624, dtype=np.uint32)
    key1 = np.asarray(rs.get_state()[1], dtype=np.uint32)
    assert rs.get_state()[2] == 624
    assert _same_state(_jump(L, key0, tab[entry]), key1)


def test_two_level_jump_and_identity():
    L, tab = _table()
    rs, key0 = _aligned(1, 77)
    assert int(tab[0][0]) == 1 and not tab[0][1:].any() and int(tab[64][0]) == 1 and not tab[64][1:].any()
    assert _same_state(_jump(L, key0, tab[0]), key0)
    rs.randint(0, 2 ** 32, size=(2 * 64 + 7) * 64 * 624, dtype=np.uint32)
    key = np.asarray(rs.get_state()[1], dtype=np.uint32)
    # the garbage in word 0's low bits after the first jump must not reach the second jump's result
    assert _same_state(_jump(L, _jump(L, key0, tab[64 + 2]), tab[7]), key)
