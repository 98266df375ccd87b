"""CPU tests: pin the oracle (FIPS-197, harness fixtures, exact integer products, decrypt-level
correctness of every homomorphic stage).  Sizes are chosen so the whole file runs in about a minute;
n = 256 instead of 768 only shortens the blind rotation, every other parameter is AES_TIGHT."""
import ctypes as C
import json
import os
import numpy as np
import pytest
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "harness_fixtures.json")))
AES_KEY = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0")     # sha256("None")[:16], SURVEY 4
AES_IV = bytes.fromhex("64b92e64992534e495b9da182409427c")


def gf_mul(a, b):
    r = 0
    for _ in range(8):
        if b & 1:
            r ^= a
        a = ((a << 1) ^ 0x1B) & 0xFF if a & 0x80 else a << 1
        b >>= 1
    return r


def inv_sbox_table():
    inv = [0] * 256
    for x in range(256):
        y = 0
        if x:
            y = next(v for v in range(1, 256) if gf_mul(x, v) == 1)
        r, s = y, y
        for _ in range(4):
            r = ((r << 1) | (r >> 7)) & 0xFF
            s ^= r
        inv[s ^ 0x63] = x
    return inv


def ctr_keystream(rk, iv, i):
    c = (int.from_bytes(iv, "big") + i) % (1 << 128)
    return O.aes_encrypt_block(rk, c.to_bytes(16, "big"))


def pack_u16(db):
    return b"".join(int(v).to_bytes(2, "big") for v in db)


@pytest.fixture(scope="module")
def small():
    p = O.aes_tight(n=256)
    keys = O.Keys(p, seed=21)
    ctx = O.Ctx(keys, threads=8)
    yield p, keys, ctx
    ctx.close()


def test_aes_fips197_appendix_c1():
    rk = O.aes_expand(bytes(range(16)))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    ct = O.aes_encrypt_block(rk, pt)
    assert ct.hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"
    assert O.aes_decrypt_block(rk, ct) == pt
    # FIPS-197 appendix A.1 last round key
    rk2 = O.aes_expand(bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")).reshape(11, 16)
    assert bytes(rk2[10]).hex() == "d014f9a8c9ee2589e13f0cc8b6630ca6"


@pytest.mark.parametrize("fx", FIX, ids=lambda f: f"size{f['size']}-seed{f['dataset_seed']}")
def test_aes_matches_harness_fixture(fx):
    """db.hex of the fixture (made with openssl from the reference's dataset) decrypts to db.txt"""
    key, iv = bytes.fromhex(fx["aes_key"]), bytes.fromhex(fx["aes_iv"])
    assert key == AES_KEY and iv == AES_IV
    rk = O.aes_expand(key)
    ct = bytes.fromhex(fx["db_hex"])
    if fx["size"] == 0:
        pt = b"".join(O.aes_decrypt_block(rk, ct[i:i + 16]) for i in range(0, len(ct), 16))
    else:
        pt = b"".join(bytes(a ^ b for a, b in zip(ctr_keystream(rk, iv, i // 16), ct[i:i + 16])) for i in range(0, len(ct), 16))
    assert pt == pack_u16(fx["db"])
    assert max(fx["db"]) == fx["max_value"]


@pytest.mark.parametrize("N,bits", [(1024, 22), (1024, 12), (1024, 1), (256, 3)])
def test_fft_product_close_to_exact(N, bits):
    L = O.lib()
    rng = np.random.default_rng(N + bits)
    small = rng.integers(-(1 << bits), (1 << bits) + 1, N).astype(np.int64)
    big = rng.integers(0, 1 << 64, N, dtype=np.uint64)
    ex = np.zeros(N, np.uint64)
    L.txo_poly_mul_exact(C.c_uint32(N), O.p_i64(small), O.p_u64(big), O.p_u64(ex))
    D = np.zeros(N); G = np.zeros(N)
    L.txo_fft_forward_int(C.c_uint32(N), O.p_i64(small), O.p_f64(D))
    L.txo_fft_forward_torus_scaled(C.c_uint32(N), O.p_u64(big), O.p_f64(G))
    P = (D.view(np.complex128) * G.view(np.complex128)).view(np.float64).copy()
    out = np.zeros(N, np.uint64)
    L.txo_fft_inverse_add(C.c_uint32(N), O.p_f64(P), O.p_u64(out))
    err = np.abs(O.signed(out - ex)).max()
    # product magnitude ~ 2^(bits + 63 + log2(N)/2); f64 keeps 53 bits
    assert err < 2.0 ** (bits + 64 + np.log2(N) / 2 - 49)


def test_fft_exact_on_small_operands():
    """when everything fits in 53 bits the transform is exact"""
    L = O.lib()
    N = 1024
    rng = np.random.default_rng(3)
    small = rng.integers(-2, 3, N).astype(np.int64)
    big = rng.integers(0, 1 << 20, N, dtype=np.uint64)
    ex = np.zeros(N, np.uint64)
    L.txo_poly_mul_exact(C.c_uint32(N), O.p_i64(small), O.p_u64(big), O.p_u64(ex))
    D = np.zeros(N); G = np.zeros(N)
    L.txo_fft_forward_int(C.c_uint32(N), O.p_i64(small), O.p_f64(D))
    L.txo_fft_forward_torus_scaled(C.c_uint32(N), O.p_u64(big), O.p_f64(G))
    P = (D.view(np.complex128) * G.view(np.complex128)).view(np.float64).copy()
    out = np.zeros(N, np.uint64)
    L.txo_fft_inverse_add(C.c_uint32(N), O.p_f64(P), O.p_u64(out))
    assert np.array_equal(out, ex)


@pytest.mark.parametrize("B,l", [(23, 1), (4, 3), (13, 3), (17, 2), (2, 7)])
def test_signed_decomposition_recomposes(B, l):
    L = O.lib()
    rng = np.random.default_rng(B * 10 + l)
    xs = list(rng.integers(0, 1 << 64, 2000, dtype=np.uint64)) + [0, (1 << 64) - 1, 1 << 63, (1 << 63) - 1, 1 << (63 - B)]
    d = (C.c_int64 * 16)()
    nonrep = 64 - B * l
    for odd in (0, 1):
        for x in xs:
            x = int(x)
            L.txo_decompose_idx(C.c_uint64(x), C.c_uint32(B), C.c_uint32(l), C.c_uint32(odd), d)
            digs = [d[i] for i in range(l)]
            assert all(-(1 << (B - 1)) <= v <= (1 << (B - 1)) for v in digs)
            # digits[i] is level l - i; recomposition equals the closest representable value mod 2^64
            rec = sum(v << (64 - B * (l - i)) for i, v in enumerate(digs)) % (1 << 64)
            closest = (((x >> (nonrep - 1)) + 1) >> 1 << nonrep) % (1 << 64)
            assert rec == closest
            assert min((x - closest) % (1 << 64), (closest - x) % (1 << 64)) <= 1 << (nonrep - 1)
            if odd == 0:                       # even indices: exactly tfhe-rs' rule
                L.txo_decompose(C.c_uint64(x), C.c_uint32(B), C.c_uint32(l), d)
                assert digs == [d[i] for i in range(l)]


def test_top_digit_is_balanced_across_coefficient_parity():
    """the one deviation from tfhe-rs: the top-level tie B/2 alternates sign with the coefficient index,
    so the top digit has (nearly) zero mean over a polynomial (tfhe-rs: about +0.4 for B = 4)"""
    L = O.lib()
    rng = np.random.default_rng(4)
    d = (C.c_int64 * 16)()
    tops = {0: [], 1: []}
    for j, x in enumerate(rng.integers(0, 1 << 64, 20000, dtype=np.uint64)):
        L.txo_decompose_idx(C.c_uint64(int(x)), 2, 7, j & 1, d)
        tops[j & 1].append(d[6])
    # a tie has probability about 1/4 and moves the digit by B = 4: the two parities differ by about 1
    assert np.mean(tops[0]) > 0.3 and np.mean(tops[1]) < -0.3 and 0.6 < np.mean(tops[0]) - np.mean(tops[1]) < 1.2
    assert abs(np.mean(tops[0] + tops[1])) < 0.15
    assert set(tops[0]) == {-1, 0, 1, 2} and set(tops[1]) == {-2, -1, 0, 1}


def test_modswitch_is_multiple_of_lut_count_and_rounds():
    L = O.lib()
    for x in [0, 1 << 63, (1 << 64) - 1, 1 << 55, (1 << 55) - 1, 3 << 54]:
        m = L.txo_modswitch(C.c_uint64(x), 10, 3)
        assert m % 8 == 0 and 0 <= m <= 2048
        assert m == (((x >> 55) + 1) >> 1) << 3


def test_keyswitch_blind_rotate_trace_levels(small):
    p, keys, ctx = small
    bits = np.array([1, 0, 0, 1], np.uint8)
    ct = O.lwe_encrypt_bits(p, keys.glwe_sk, bits, p.glwe_std, seed=5)
    sm = ctx.lwe_keyswitch(ct)
    ph = O.lwe_phase(keys.lwe_sk, sm)
    assert np.abs(O.signed(ph - (bits.astype(np.uint64) << np.uint64(63)))).max() < 2 ** 59
    acc = ctx.blind_rotate(sm)
    glev = ctx.extract_trace(acc)
    for b in range(4):
        for lv in range(7):
            php = O.glwe_phase(p, keys.glwe_sk, glev[b, lv])
            exp = np.uint64(int(bits[b]) << (64 - 2 * (lv + 1)))
            assert abs(int(O.signed(php[:1] - exp)[0])) < 2 ** 52       # constant coefficient = bit * 2^(64-2 level)
            assert np.abs(O.signed(php[1:])).max() < 2 ** 40            # the trace killed the rest


def test_trace_multiplies_constant_by_N(small):
    p, keys, ctx = small
    rng = np.random.default_rng(9)
    g = np.zeros((1, 3, 1024), np.uint64)
    g[0, 2] = rng.integers(0, 1 << 54, 1024, dtype=np.uint64)    # trivial GLWE, values < 2^54
    out = ctx.trace(g)
    ph = O.glwe_phase(p, keys.glwe_sk, out[0])
    assert abs(int(O.signed(ph[:1] - (g[0, 2, 0] << np.uint64(10)))[0])) < 2 ** 40
    assert np.abs(O.signed(ph[1:])).max() < 2 ** 40


def test_cbs_and_lut_tree_give_keyed_sbox_multiples(small):
    p, keys, ctx = small
    x = 0xA7
    bits = np.array([(x >> i) & 1 for i in range(8)], np.uint8)
    ct = O.lwe_encrypt_bits(p, keys.glwe_sk, bits, p.glwe_std, seed=6)
    gf = ctx.circuit_bootstrap(ct)
    luts = keys.round_luts(AES_KEY)
    gl = ctx.gl
    out = ctx.lut_eval(gf, luts[3 * 4 * 16 * 2 * gl:], 4, 2 * gl, 16 * 2 * gl)      # set 3 = round 6
    rk = O.aes_expand(AES_KEY).reshape(11, 16)
    inv_sbox_x = inv_sbox_table()[x]
    for mi, m in enumerate((9, 11, 13, 14)):
        dec = O.lwe_decrypt_bits(keys.glwe_sk, out[mi])
        val = sum(int(dec[i]) << i for i in range(8))
        assert val == gf_mul(inv_sbox_x ^ int(rk[6][0]), m)
        ph = O.lwe_phase(keys.glwe_sk, out[mi])
        assert np.abs(O.signed(ph - (dec.astype(np.uint64) << np.uint64(63)))).max() < 2 ** 61


def test_transcipher_toy_fixture_ecb(small):
    """BASELINE config 1: toy instance, one block, decrypts bit-exact to the harness dataset"""
    p, keys, ctx = small
    fx = [f for f in FIX if f["size"] == 0 and f["dataset_seed"] == 123][0]
    luts = keys.round_luts(AES_KEY)
    out = ctx.transcipher(luts, bytes.fromhex(fx["db_hex"]))
    bits = O.lwe_decrypt_bits(keys.glwe_sk, out)
    assert O.bits_msb_to_bytes(bits) == pack_u16(fx["db"])
    # client_postprocess_aes_decryption.rs:18-37: every 16 bits -> u16, first bit = MSB
    vals = [int("".join(str(int(b)) for b in bits[i:i + 16]), 2) for i in range(0, 128, 16)]
    assert vals == fx["db"]


def test_transcipher_ctr_first_block_of_small_fixture(small):
    p, keys, ctx = small
    fx = [f for f in FIX if f["size"] == 1 and f["dataset_seed"] == 7][0]
    ctr_luts = keys.ctr_luts(AES_KEY)
    ct = bytes.fromhex(fx["db_hex"])[:16]
    out = ctx.transcipher_ctr(ctr_luts, AES_IV, ct)
    assert O.bits_msb_to_bytes(O.lwe_decrypt_bits(keys.glwe_sk, out)) == pack_u16(fx["db"][:8])


def test_max_tournament_and_reference_order_agree(small):
    p, keys, ctx = small
    vals = [0x1234, 0xF00D, 0xF00D, 0x0FFF]          # includes an equal pair (SURVEY 3.2 defect)
    bits = np.array([(v >> (15 - i)) & 1 for v in vals for i in range(16)], np.uint8)
    ct = O.lwe_encrypt_bits(p, keys.glwe_sk, bits, p.glwe_std, seed=8)
    for tournament in (True, False):
        out = ctx.max_u16(ct, tournament=tournament)
        dec = O.lwe_decrypt_bits(keys.glwe_sk, out)
        assert int("".join(str(int(b)) for b in dec), 2) == max(vals)


def test_inner_product_bitsliced_circuit(small):
    """mini-workload 1 (cleartext_impl.py:65-70): sum of pairwise products mod 2^16"""
    p, keys, ctx = small
    vals = [0x1234, 0xFFFF, 0x0003, 0x8001]
    bits = np.array([(v >> (15 - i)) & 1 for v in vals for i in range(16)], np.uint8)
    ct = O.lwe_encrypt_bits(p, keys.glwe_sk, bits, p.glwe_std, seed=8)
    out = ctx.inner_product_u16(ct)
    dec = O.lwe_decrypt_bits(keys.glwe_sk, out)
    half = len(vals) // 2
    exp = sum((x * y) % 65536 for x, y in zip(vals[:half], vals[half:])) % 65536
    assert int("".join(str(int(b)) for b in dec), 2) == exp
    ph = O.lwe_phase(keys.glwe_sk, out)
    assert np.abs(O.signed(ph - (dec.astype(np.uint64) << np.uint64(63)))).max() < 2 ** 60
