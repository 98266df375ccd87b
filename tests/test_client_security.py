"""CPU tests of the client library's handling of secrets (round-2 advisor findings): every keyed-LUT
table is a real GLWE encryption (the reference's trivial tables disclose the AES key,
submission/src/data_struct.rs:146-151), an unseeded generator without OS entropy is an error, and
the stage-file reader / key loader reject truncated or foreign-sized key material."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AES_KEY = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0")     # harness default key


@pytest.fixture(scope="module")
def keys256():
    import txhip, txclient
    return txclient.ClientKeys(txhip.aes_tight(n=256), seed=11)


def _masks(luts, K=3, N=1024):
    g = luts.reshape(-1, K, N)
    return g[:, :K - 1, :], g[:, K - 1, :]


def test_every_ctr_table_is_encrypted(keys256):
    luts = keys256.ctr_luts(AES_KEY, seed=12)
    mask, body = _masks(luts)
    g = luts.reshape(10, 3, 16, 2, 3, 1024)
    used = np.ones((10, 3), bool); used[9, 1:] = False          # set 9 has one table (rk10 folded in)
    for s in range(10):
        for m in range(3):
            nz = (g[s, m, :, :, :2, :] != 0).any(axis=-1)       # a trivial GLWE has an all-zero mask
            assert nz.all() == used[s, m], (s, m)
    # bodies carry noise: no body is the bare table (a multiple of 2^63)
    gb = g[:9, :, :, :, 2, :]
    assert ((gb & np.uint64((1 << 63) - 1)) != 0).mean() > 0.99


def test_every_round_table_is_encrypted_by_default_and_reference_form_is_opt_in(keys256):
    luts = keys256.round_luts(AES_KEY, seed=13)
    g = luts.reshape(-1, 3, 1024)
    assert (g[:, :2, :] != 0).any(axis=-1).all()
    ref = keys256.round_luts(AES_KEY, seed=13, reference_trivial=True).reshape(37, 16, 2, 3, 1024)
    m = (ref[:, :, :, :2, :] != 0).any(axis=(-1, -2))
    assert m[:4].all() and not m[4:].any()                      # set 0 (rounds 10/9) encrypted, the rest trivial
    # the trivial form really does disclose a round key: table bits are readable without any secret
    import txclient
    body = ref[36, 0, :, 2, :]                                  # round 0, byte 0, both accumulators
    bits = (body >> np.uint64(63)).astype(np.uint8).reshape(2, 4, 256)   # [acc][lut][x]
    t0 = sum(int(bits[b // 4, b % 4, 0]) << b for b in range(8))         # InvS[0] ^ rk0[0]
    inv_s0 = txclient.aes_ecb_decrypt(bytes(16), bytes(16))     # independent of this: just check structure
    assert (t0 ^ 0x52) == AES_KEY[0]                            # InvS[0] = 0x52, rk0 = the AES key itself


def test_encrypted_tables_decrypt_to_the_same_bits(keys256):
    import txclient
    enc = keys256.round_luts(AES_KEY, seed=13).reshape(37, 16, 2, 3, 1024)
    ref = keys256.round_luts(AES_KEY, seed=13, reference_trivial=True).reshape(37, 16, 2, 3, 1024)
    sk = keys256.glwe_sk.reshape(2, 1024)
    for idx in [(5, 3, 1), (36, 0, 0), (20, 15, 1)]:
        g = enc[idx]
        ph = g[2].copy()
        for q in range(2):                                      # phase = b - a * s (negacyclic)
            a = g[q]
            for i in np.nonzero(sk[q])[0]:
                rolled = np.roll(a, i)
                rolled[:i] = np.uint64(0) - rolled[:i]
                ph -= rolled
        dec = ((ph + np.uint64(1 << 62)) >> np.uint64(63)).astype(np.uint8)
        assert np.array_equal(dec, (ref[idx][2] >> np.uint64(63)).astype(np.uint8))


@pytest.fixture(scope="module")
def no_entropy_env(tmp_path_factory):
    """LD_PRELOAD shim (test-only code, built here) that makes getrandom(2) and /dev/urandom fail: the product has
    no switch that disables its entropy source"""
    d = tmp_path_factory.mktemp("shim")
    src = d / "noentropy.c"
    src.write_text(
        "#define _GNU_SOURCE\n#include <dlfcn.h>\n#include <errno.h>\n#include <stdio.h>\n#include <string.h>\n#include <sys/types.h>\n"
        "ssize_t getrandom(void *b, size_t n, unsigned f) { (void)b; (void)n; (void)f; errno = ENOSYS; return -1; }\n"
        "static FILE *deny(const char *p, const char *m, const char *sym) {\n"
        "    if (p && !strcmp(p, \"/dev/urandom\")) { errno = ENOENT; return NULL; }\n"
        "    FILE *(*real)(const char *, const char *) = (FILE *(*)(const char *, const char *))dlsym(RTLD_NEXT, sym);\n"
        "    return real(p, m); }\n"
        "FILE *fopen(const char *p, const char *m) { return deny(p, m, \"fopen\"); }\n"
        "FILE *fopen64(const char *p, const char *m) { return deny(p, m, \"fopen64\"); }\n")
    so = d / "noentropy.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", str(so), str(src), "-ldl"])
    return dict(os.environ, LD_PRELOAD=str(so))


def test_unseeded_generator_without_entropy_is_an_error(keys256, no_entropy_env):
    """seed == 0 is the production default of the client stages; no entropy must never mean a constant key"""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import txhip, txclient\n"
        "try:\n"
        "    txclient.ClientKeys(txhip.aes_tight(n=256), seed=0)\n"
        "except txclient.ClientError as e:\n"
        "    print('ERR', e); sys.exit(3)\n"
        "sys.exit(0)\n" % (os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"), ROOT))
    r = subprocess.run([sys.executable, "-c", code], env=no_entropy_env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "entropy" in r.stdout, (r.returncode, r.stdout, r.stderr)
    # and with entropy two unseeded generations differ
    import txhip, txclient
    p = txhip.aes_tight(n=256)
    a = keys256.encrypt_bits(np.zeros(2, np.uint8), seed=0)
    b = keys256.encrypt_bits(np.zeros(2, np.uint8), seed=0)
    assert not np.array_equal(a, b)


def test_key_generation_stage_exits_nonzero_without_entropy(tmp_path, no_entropy_env):
    exe = os.path.join(ROOT, "temp-fhe-transciphering_amd", "bin", "client_key_generation")
    env = dict(no_entropy_env)
    env.pop("TX_SEED", None)
    r = subprocess.run([exe, "0"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "entropy" in r.stderr
    assert not (tmp_path / "io" / "toy" / "secret_keys" / "glwe_sk.bin").exists()


def test_stage_reader_rejects_truncated_key_file(tmp_path):
    """server stage: a header that promises more words than the file holds is an error before any GPU work"""
    d = tmp_path / "io" / "toy" / "public_keys"
    d.mkdir(parents=True)
    (tmp_path / "datasets" / "toy").mkdir(parents=True)
    (tmp_path / "datasets" / "toy" / "db.hex").write_text("00" * 16)
    hdr = np.array([0x0000003436555854, 1 << 40], np.uint64).tobytes()
    for n in ("bsk", "ksk", "auto_keys", "ss_key"):
        (d / (n + ".bin")).write_bytes(hdr + bytes(64))
    exe = os.path.join(ROOT, "temp-fhe-transciphering_amd", "bin", "server_encrypted_aes_decryption")
    r = subprocess.run([exe, "0"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
    assert "does not match the file size" in r.stderr          # the stage reads its key files before it touches the GPU
    # well-formed files of the wrong length (keys of another parameter set) are refused the same way
    for n in ("bsk", "ksk", "auto_keys", "ss_key"):
        (d / (n + ".bin")).write_bytes(np.array([0x0000003436555854, 8], np.uint64).tobytes() + bytes(64))
    r = subprocess.run([exe, "0"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "do not match the AES_TIGHT parameter set" in r.stderr


def test_eval_key_word_counts_are_exported_and_consistent():
    """tx_eval_key_words needs a context, hence a GPU; without one the client-side counts still pin the ABI"""
    import txhip, txclient
    L = txclient.lib()
    p = txhip.aes_tight()
    K, N = 3, 1024
    assert L.txc_bsk_words(C.byref(p)) == 768 * 1 * K * K * N
    assert L.txc_ksk_words(C.byref(p)) == 8 * 3 * 4 * 256
    assert L.txc_auto_words(C.byref(p)) == 10 * 2 * 3 * K * N
    assert L.txc_ss_words(C.byref(p)) == 2 * 2 * K * K * N
    assert txhip.lib().tx_load_eval_keys.argtypes[2] is C.c_size_t


def test_stage_file_format_round_trip_and_checks(tmp_path):
    """the TXU64 format of the stage files (host/stages.cpp) as python/txclient.py reads and writes it: round trip, a
    truncated file and a foreign file are refused; StageKeys (what a server finds under io/<size>/) checks the key
    lengths against the parameter set and has no secret key unless the client left one on the same machine"""
    import txclient
    import txhip
    w = np.arange(1000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    f = str(tmp_path / "a" / "w.bin")
    txclient.write_words(f, w)
    assert os.path.getsize(f) == 16 + 8 * 1000
    assert np.array_equal(np.array(txclient.read_words(f)), w)
    raw = open(f, "rb").read()
    open(f, "wb").write(raw[:-8])
    with pytest.raises(txclient.ClientError, match="does not match the file size"):
        txclient.read_words(f)
    open(f, "wb").write(b"NOTTXU64" + raw[8:])
    with pytest.raises(txclient.ClientError, match="bad header"):
        txclient.read_words(f)
    # a stage executable reads what python wrote and refuses it for its length, not for its format
    p = txhip.aes_tight()
    io = tmp_path / "io" / "toy"
    for name in ("bsk", "ksk", "auto_keys", "ss_key"):
        txclient.write_words(str(io / "public_keys" / (name + ".bin")), w)
    txclient.write_words(str(io / "ciphertexts_upload" / "trans_key.bin"), w)
    with pytest.raises(txclient.ClientError, match="do not match the AES_TIGHT parameter set"):
        txclient.StageKeys(p, str(io))
    os.makedirs(tmp_path / "datasets" / "toy")
    (tmp_path / "datasets" / "toy" / "db.hex").write_text("00" * 16)
    exe = os.path.join(ROOT, "temp-fhe-transciphering_amd", "bin", "server_encrypted_aes_decryption")
    r = subprocess.run([exe, "0"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "do not match the AES_TIGHT parameter set" in r.stderr
