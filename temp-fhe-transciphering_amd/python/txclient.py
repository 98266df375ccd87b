"""ctypes binding of libtxclient.so (client side: key generation, keyed LUTs, decryption)."""
import ctypes as C
import os
import numpy as np
from txhip import TxParams, PKG_DIR, u64p, u8p

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(PKG_DIR, "libtxclient.so"))
        for name in ("txc_bsk_words", "txc_ksk_words", "txc_auto_words", "txc_ss_words", "txc_lut_words", "txc_ctr_lut_words"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.POINTER(TxParams)]
        _LIB = L
    return _LIB


def _p64(a): return a.ctypes.data_as(u64p)
def _p8(a): return a.ctypes.data_as(u8p)


ERR_ENTROPY = -2                     # txclient.h TXC_ERR_ENTROPY
LUTS_REFERENCE_TRIVIAL = 1           # txclient.h TXC_LUTS_REFERENCE_TRIVIAL


class ClientError(RuntimeError):
    pass


def _chk(rc, what):
    if rc == ERR_ENTROPY:
        raise ClientError(what + ": the OS returned no entropy for an unseeded generator")
    if rc:
        raise ClientError(what + " failed (%d)" % rc)


class ClientKeys:
    """What client_key_generation writes: secret keys + the four evaluation keys (standard domain)."""

    def __init__(self, p, seed):
        L = lib()
        self.p = p
        self.lwe_sk = np.zeros(p.n, np.uint64)
        self.glwe_sk = np.zeros(p.k * p.N, np.uint64)
        self.bsk = np.zeros(L.txc_bsk_words(C.byref(p)), np.uint64)
        self.ksk = np.zeros(L.txc_ksk_words(C.byref(p)), np.uint64)
        self.auto = np.zeros(L.txc_auto_words(C.byref(p)), np.uint64)
        self.ss = np.zeros(L.txc_ss_words(C.byref(p)), np.uint64)
        rc = L.txc_keygen(C.byref(p), C.c_uint64(seed), _p64(self.lwe_sk), _p64(self.glwe_sk), _p64(self.bsk),
                          _p64(self.ksk), _p64(self.auto), _p64(self.ss))
        _chk(rc, "txc_keygen")

    def round_luts(self, aes_key, seed, reference_trivial=False):
        """every table GLWE-encrypted; reference_trivial=True gives the reference's form (rounds 8..0 as
        trivial GLWEs, data_struct.rs:146-151), which discloses the AES key -- format comparisons only"""
        luts = np.zeros(lib().txc_lut_words(C.byref(self.p)), np.uint64)
        k = np.frombuffer(bytes(aes_key), np.uint8).copy()
        _chk(lib().txc_make_round_luts_ex(C.byref(self.p), C.c_uint64(seed), _p8(k), _p64(self.glwe_sk),
                                          C.c_uint(LUTS_REFERENCE_TRIVIAL if reference_trivial else 0), _p64(luts)),
             "txc_make_round_luts")
        return luts

    def ctr_luts(self, aes_key, seed):
        luts = np.zeros(lib().txc_ctr_lut_words(C.byref(self.p)), np.uint64)
        k = np.frombuffer(bytes(aes_key), np.uint8).copy()
        _chk(lib().txc_make_ctr_luts(C.byref(self.p), C.c_uint64(seed), _p8(k), _p64(self.glwe_sk), _p64(luts)),
             "txc_make_ctr_luts")
        return luts

    def decrypt_bits(self, lwe):
        lwe = np.ascontiguousarray(lwe, np.uint64)
        cnt = lwe.size // (self.p.k * self.p.N + 1)
        bits = np.zeros(cnt, np.uint8)
        lib().txc_decrypt_bits(C.byref(self.p), _p64(self.glwe_sk), _p64(lwe), C.c_size_t(cnt), _p8(bits))
        return bits

    def encrypt_bits(self, bits, seed):
        bits = np.ascontiguousarray(bits, np.uint8)
        out = np.zeros(bits.size * (self.p.k * self.p.N + 1), np.uint64)
        _chk(lib().txc_encrypt_bits(C.byref(self.p), C.c_uint64(seed), _p64(self.glwe_sk), _p8(bits), C.c_size_t(bits.size),
                                    _p64(out)), "txc_encrypt_bits")
        return out.reshape(bits.size, -1)


def bits_to_u16(bits):
    bits = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros(bits.size // 16, np.uint16)
    lib().txc_bits_to_u16(_p8(bits), C.c_size_t(bits.size), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def aes_ecb_encrypt(key, data):
    d = np.frombuffer(bytes(data), np.uint8).copy(); o = np.zeros_like(d)
    k = np.frombuffer(bytes(key), np.uint8).copy()
    lib().txc_aes128_ecb_encrypt(_p8(k), _p8(d), C.c_size_t(d.size // 16), _p8(o))
    return bytes(o)


def aes_ecb_decrypt(key, data):
    d = np.frombuffer(bytes(data), np.uint8).copy(); o = np.zeros_like(d)
    k = np.frombuffer(bytes(key), np.uint8).copy()
    lib().txc_aes128_ecb_decrypt(_p8(k), _p8(d), C.c_size_t(d.size // 16), _p8(o))
    return bytes(o)


def aes_ctr_crypt(key, iv, data):
    d = np.frombuffer(bytes(data), np.uint8).copy(); o = np.zeros_like(d)
    k = np.frombuffer(bytes(key), np.uint8).copy(); v = np.frombuffer(bytes(iv), np.uint8).copy()
    lib().txc_aes128_ctr_crypt(_p8(k), _p8(v), _p8(d), C.c_size_t(d.size), _p8(o))
    return bytes(o)


# ---- the stage executables' own file format (host/stages.cpp): 16-byte header {magic "TXU64", word count} + little-endian u64 ----
_MAGIC = 0x0000003436555854


def read_words(path):
    """the u64 words of a TXU64 file, memory-mapped read-only; the header's count must match the file size"""
    size = os.path.getsize(path)
    if size < 16:
        raise ClientError("bad header in " + path)
    hdr = np.fromfile(path, dtype="<u8", count=2)
    if int(hdr[0]) != _MAGIC:
        raise ClientError("bad header in " + path)
    if (size - 16) % 8 or (size - 16) // 8 != int(hdr[1]):
        raise ClientError("word count in the header of " + path + " does not match the file size")
    if int(hdr[1]) == 0:
        return np.zeros(0, np.uint64)
    return np.memmap(path, dtype="<u8", mode="r", offset=16, shape=(int(hdr[1]),))


def write_words(path, words):
    words = np.ascontiguousarray(words, dtype="<u8").reshape(-1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        np.array([_MAGIC, words.size], dtype="<u8").tofile(f)
        words.tofile(f)


class StageKeys:
    """What a SERVER finds under io/<size>/ after the client stages ran (client_key_generation, client_encode_encrypt):
    public_keys/{bsk,ksk,auto_keys,ss_key}.bin and ciphertexts_upload/trans_key.bin, checked against the parameter set.
    The secret key is loaded only if it is there (a test harness on one machine; a real server never has it)."""

    def __init__(self, p, io_dir):
        L = lib()
        self.p = p
        pk = os.path.join(io_dir, "public_keys")
        self.bsk, self.ksk = read_words(os.path.join(pk, "bsk.bin")), read_words(os.path.join(pk, "ksk.bin"))
        self.auto, self.ss = read_words(os.path.join(pk, "auto_keys.bin")), read_words(os.path.join(pk, "ss_key.bin"))
        want = (L.txc_bsk_words(C.byref(p)), L.txc_ksk_words(C.byref(p)), L.txc_auto_words(C.byref(p)), L.txc_ss_words(C.byref(p)))
        if (self.bsk.size, self.ksk.size, self.auto.size, self.ss.size) != want:
            raise ClientError("public_keys/*.bin do not match the AES_TIGHT parameter set")
        self.trans_key = read_words(os.path.join(io_dir, "ciphertexts_upload", "trans_key.bin"))
        sk = os.path.join(io_dir, "secret_keys", "glwe_sk.bin")
        self.glwe_sk = np.array(read_words(sk)) if os.path.exists(sk) else None

    def decrypt_bits(self, lwe):
        if self.glwe_sk is None:
            raise ClientError("no secret key under io/: a server cannot decrypt")
        lwe = np.ascontiguousarray(lwe, np.uint64)
        cnt = lwe.size // (self.p.k * self.p.N + 1)
        bits = np.zeros(cnt, np.uint8)
        lib().txc_decrypt_bits(C.byref(self.p), _p64(self.glwe_sk), _p64(lwe), C.c_size_t(cnt), _p8(bits))
        return bits
