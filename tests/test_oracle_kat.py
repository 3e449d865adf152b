"""Known-answer tests pinning the primitives the oracle borrows from published algorithms.

The reference ships no tests or golden vectors for this path: these vectors are the
published ones of the algorithms themselves (eSTREAM/RFC 7539 ChaCha keystreams, Random123 Philox
kat_vectors), so they pin the generators, not rustray's use of them ("parity unpinned" there).
"""
import ctypes as C

import numpy as np


def _chacha(oracle, rounds):
    key = np.zeros(8, np.uint32)
    out = np.zeros(16, np.uint32)
    oracle.lib().rro_chacha_block(key.ctypes.data_as(C.c_void_p), C.c_uint64(0), rounds, out.ctypes.data_as(C.c_void_p))
    return out.tobytes().hex()


def test_chacha20_zero_key_block0(oracle):
    # RFC 7539 A.1 test vector #1 (all-zero key and nonce, counter 0)
    assert _chacha(oracle, 20) == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                                   "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")


def test_chacha12_zero_key_block0(oracle):
    # eSTREAM ChaCha12, 256-bit key, TC1 (all zero): the core of rand 0.8's StdRng
    assert _chacha(oracle, 12) == ("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
                                   "0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")


def test_chacha8_zero_key_block0(oracle):
    assert _chacha(oracle, 8) == ("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
                                  "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")


def _philox(oracle, ctr, key):
    c, k, o = np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), np.zeros(4, np.uint32)
    oracle.lib().rro_philox(c.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    return [int(v) for v in o]


def test_philox4x32_10_random123_vectors(oracle):
    assert _philox(oracle, [0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox(oracle, [0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox(oracle, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_sample_table_shape_and_determinism(oracle):
    # reference src/raytracing.rs:290-313: cell_size = 1 or next_power_of_two(samples + 2) / 2
    for samples, cell in ((1, 1), (2, 2), (16, 16), (64, 64), (128, 128), (5, 4), (6, 4), (7, 8)):
        xy, cs = oracle.sample_table(samples)
        assert cs == cell
        assert xy.shape == (samples, 2)
        assert int(xy.max(initial=0)) < cell
        assert len({(int(a), int(b)) for a, b in xy}) == samples  # a truncated permutation: no repeats
        xy2, _ = oracle.sample_table(samples)
        assert (xy == xy2).all()
    xy, _ = oracle.sample_table(1)
    assert xy.tolist() == [[0, 0]]


def test_sample_table_golden(oracle):
    # regression pin of the restated StdRng::seed_from_u64(0) + shuffle (self-generated, see tests/golden/README.md)
    xy, _ = oracle.sample_table(16)
    assert xy[:6].tolist() == [[8, 15], [1, 11], [4, 1], [7, 0], [15, 9], [4, 9]]


def test_product_sample_table_matches_oracle(oracle):
    # the product's own ChaCha12 / shuffle (rustray_amd/csrc/rr_api.hip) against the oracle's, no GPU needed
    from rustray_amd import capi
    for samples in (1, 2, 3, 16, 64, 128, 512):
        a, ca = capi.sample_table(samples)
        b, cb = oracle.sample_table(samples)
        assert ca == cb and (a == b).all()
