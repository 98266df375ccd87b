"""CPU tests of the drop-in boundary: the shared library loads without a GPU, exports every
symbol declared in include/tfhe_mi355x.h, validates parameters, and fails loudly (no fallback)."""
import ctypes as C
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tfhe_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import txhip
    L = txhip.lib()
    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/tfhe_mi355x.h but not exported"
    assert set(txhip.EXPORTS) == set(syms)


def test_client_library_exports():
    import txclient
    L = txclient.lib()
    text = open(os.path.join(ROOT, "temp-fhe-transciphering_amd", "host", "txclient.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for s in set(re.findall(r"\b(txc_[a-z0-9_]+)\s*\(", text)):
        assert hasattr(L, s)


def test_params_match_aes_tight():
    """tx_params_aes_tight mirrors cbs_lib/src/aes_instances.rs:76-97"""
    import txhip
    p = txhip.aes_tight()
    assert (p.n, p.N, p.k) == (768, 1024, 2)
    assert (p.pbs_base_log, p.pbs_level) == (23, 1)
    assert (p.ks_base_log, p.ks_level, p.ks_N) == (4, 3, 256)
    assert (p.auto_base_log, p.auto_level, p.auto_split) == (13, 3, 41)
    assert (p.ss_base_log, p.ss_level) == (17, 2)
    assert (p.cbs_base_log, p.cbs_level, p.log_lut_count) == (2, 7, 3)
    assert abs(p.lwe_std - 7.0204746294012e-06) < 1e-18 and abs(p.glwe_std - 2.9403601535432533e-16) < 1e-28


def test_create_rejects_foreign_parameters():
    import txhip
    p = txhip.aes_tight()
    p.cbs_level = 6
    with pytest.raises(txhip.TxError, match="AES_TIGHT"):
        txhip.Engine(p, 0)
    p = txhip.aes_tight(n=300)
    with pytest.raises(txhip.TxError, match="multiple of 256"):
        txhip.Engine(p, 0)


def test_no_cpu_fallback_without_gpu():
    """on a box without a HIP device the product path must fail loudly, never route elsewhere"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import txhip
    with pytest.raises(txhip.TxError, match="no CPU fallback"):
        txhip.Engine(txhip.aes_tight(), 0)


def test_product_sources_do_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "temp-fhe-transciphering_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".hpp", ".h", ".cpp", ".py", "Makefile")):
                t = open(os.path.join(d, f)).read()
                assert "txoracle" not in t and "oracle_lib" not in t and "libtxoracle" not in t, f


def test_client_aes_and_bit_packing():
    import txclient
    key = bytes(range(16))
    assert txclient.aes_ecb_encrypt(key, bytes.fromhex("00112233445566778899aabbccddeeff")).hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"
    bits = np.array([1] + [0] * 15 + [0] * 15 + [1], np.uint8)
    assert txclient.bits_to_u16(bits).tolist() == [0x8000, 1]


def test_header_is_plain_c99(tmp_path):
    """the drop-in boundary is a C ABI: the header must compile as strict C99 (no C++ or torch types) and link
    against the shared library from a C translation unit"""
    import subprocess
    src = tmp_path / "abi_c.c"
    src.write_text('#include "tfhe_mi355x.h"\n'
                   'int main(void) { tx_params p; tx_params_aes_tight(&p); return !(p.N == 1024 && p.k == 2 && tx_last_error() != 0); }\n')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "temp-fhe-transciphering_amd")
    exe = tmp_path / "abi_c"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe),
                        "-L", libdir, "-ltfhe_mi355x", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)       # no GPU call: parameter table only
    assert r.returncode == 0, r.stdout + r.stderr


def test_install_layout_for_the_reference_harness(tmp_path):
    """`make install PREFIX=<root>/submission/target/release`: the ten stage names the
    reference's harness runs (harness/run_submission.py:68-115) exist there, load their libraries from that directory
    and honour the argv / exit-code contract (usage error -> 1, the two stub stages -> 0) without a GPU"""
    import subprocess
    prefix = tmp_path / "submission" / "target" / "release"
    host = os.path.join(ROOT, "temp-fhe-transciphering_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host, "install", "PREFIX=" + str(prefix)])
    stages = ["client_preprocess", "client_key_generation", "client_encode_encrypt", "server_preprocess_dataset",
              "server_encrypted_aes_decryption", "server_encrypted_compute", "client_decrypt_decode_aes_decryption",
              "client_postprocess_aes_decryption", "client_decrypt_decode", "client_postprocess"]
    for s in stages:
        assert os.access(prefix / s, os.X_OK), s
    ldd = subprocess.run(["ldd", str(prefix / "server_encrypted_compute")], capture_output=True, text=True).stdout
    for lib in ("libtxclient.so", "libtfhe_mi355x.so"):
        line = [l for l in ldd.splitlines() if lib in l][0]
        assert str(prefix) in line, line                     # not the build tree: the installed copy
    assert subprocess.run([str(prefix / "client_preprocess"), "0"], cwd=tmp_path).returncode == 0
    assert subprocess.run([str(prefix / "server_preprocess_dataset"), "2"], cwd=tmp_path).returncode == 0
    r = subprocess.run([str(prefix / "server_encrypted_compute")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr
    r = subprocess.run([str(prefix / "client_postprocess"), "7"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "size must be" in r.stderr


def test_timing_struct_layout_matches_the_python_binding(tmp_path):
    """tx_timing grew in round 3 (per-kernel blind-rotation totals): size and field offsets of the C struct, as a C
    compiler lays them out, are what txhip.TxTiming (ctypes) assumes; same for tx_params"""
    import subprocess
    import txhip
    fields_t = [n for n, _ in txhip.TxTiming._fields_]
    fields_p = [n for n, _ in txhip.TxParams._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tfhe_mi355x.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(tx_timing));\n' +
                   "".join('  printf("%%zu\\n", offsetof(tx_timing, %s));\n' % f for f in fields_t) +
                   '  printf("%zu\\n", sizeof(tx_params));\n' +
                   "".join('  printf("%%zu\\n", offsetof(tx_params, %s));\n' % f for f in fields_p) +
                   '  return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    want = [C.sizeof(txhip.TxTiming)] + [getattr(txhip.TxTiming, f).offset for f in fields_t] + \
           [C.sizeof(txhip.TxParams)] + [getattr(txhip.TxParams, f).offset for f in fields_p]
    assert got == want


def test_ctr_shared_columns_are_exactly_the_equal_columns_of_cleartext_aes():
    """host logic of the CTR schedule's shared second round (tx_ctr_shared_columns, no GPU): two (block, column) pairs are
    put in one group exactly when the cleartext state entering the second round -- AddRoundKey(0), SubBytes, ShiftRows,
    MixColumns of the counter blocks, computed here with the client library's AES tables -- agrees in that column for
    EVERY key tried (the grouping may not depend on the key); consecutive counters, a carry into byte 14, a wrap around
    2^128 and a first_block offset"""
    import txhip
    import txclient
    import oracle_lib as O

    def gm(a, b):
        x = 0
        for _ in range(8):
            if b & 1: x ^= a
            a = ((a << 1) ^ 0x1B) & 0xFF if a & 0x80 else a << 1
            b >>= 1
        return x

    rng = np.random.default_rng(3)
    cases = [(bytes.fromhex("64b92e64992534e495b9da182409427c"), 0, 64), (bytes(14) + b"\x07\xfe", 0, 5),
             (bytes([0xFF] * 15) + b"\xfd", 0, 6), (bytes.fromhex("64b92e64992534e495b9da182409427c"), 120, 300)]
    for iv, first, nb in cases:
        rep, col = txhip.ctr_shared_columns(iv, first, nb)
        assert rep.shape[0] == nb * 4 and col.shape[0] == rep.max() + 1
        cols_by_key = []
        for trial in range(3):
            key = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
            rk = O.aes_expand(key).reshape(11, 16)
            state_cols = np.zeros((nb, 4, 4), np.uint8)
            for b in range(nb):
                ctr = ((int.from_bytes(iv, "big") + first + b) % (1 << 128)).to_bytes(16, "big")
                # the first round of cleartext AES on the counter block (S-box computed below from the field inverse)
                s = bytes(a ^ k for a, k in zip(ctr, rk[0]))
                sb = bytes(SBOX[v] for v in s)
                sh = bytes(sb[4 * ((c + r) % 4) + r] for c in range(4) for r in range(4))
                for c in range(4):
                    colv = sh[4 * c:4 * c + 4]
                    state_cols[b, c] = [gm(colv[r], 2) ^ gm(colv[(r + 1) % 4], 3) ^ colv[(r + 2) % 4] ^ colv[(r + 3) % 4] for r in range(4)]
            cols_by_key.append(state_cols)
        for b in range(nb):
            for c in range(4):
                b2, c2 = divmod(int(col[rep[b * 4 + c]]), 4)
                assert c2 == c
                for st in cols_by_key:
                    assert np.array_equal(st[b, c], st[b2, c])              # grouped => equal for every key
        # and distinct groups of one column index differ for at least one of the keys (no sharing is missed)
        for c in range(4):
            groups = sorted(set(int(rep[b * 4 + c]) for b in range(nb)))
            reps = [divmod(int(col[g]), 4)[0] for g in groups]
            for i in range(len(reps)):
                for j in range(i + 1, min(len(reps), i + 40)):
                    assert any(not np.array_equal(st[reps[i], c], st[reps[j], c]) for st in cols_by_key)
    rep, col = txhip.ctr_shared_columns(bytes.fromhex("64b92e64992534e495b9da182409427c"), 0, 64)
    assert col.shape[0] == 67                                                     # the medium instance: 64 + 3 distinct columns


def _aes_sbox():
    def gm(a, b):
        x = 0
        for _ in range(8):
            if b & 1: x ^= a
            a = ((a << 1) ^ 0x1B) & 0xFF if a & 0x80 else a << 1
            b >>= 1
        return x
    s = [0] * 256
    for x in range(256):
        y = next(v for v in range(1, 256) if gm(x, v) == 1) if x else 0
        r, t = y, y
        for _ in range(4):
            r = ((r << 1) | (r >> 7)) & 0xFF
            t ^= r
        s[x] = t ^ 0x63
    return s


SBOX = _aes_sbox()


def test_blind_rotation_launch_plan():
    """host logic of the blind rotation's dispatch (tx_plan_blind_rotate, no GPU), for a 256-CU chip: the sizes the round
    schedules produce -- one block, the 8- and 16-block shares of the medium instance, the medium instance itself and its shared
    CTR round -- every unit in exactly one launch, in order"""
    import txhip
    P = txhip.plan_blind_rotate
    assert P(0) == []
    assert P(128) == [(6, 128)] and P(768) == [(6, 768)]                     # up to three units per CU: three waves per unit
    assert P(769) == [(8, 769)] and P(1024) == [(8, 1024)]                   # three to four: four units per twelve-wave workgroup, soft barriers
    assert P(1025) == [(5, 1025)] and P(2048) == [(5, 2048)] and P(8192) == [(5, 8192)]
    assert P(2144) == [(5, 2048), (6, 96)]                                   # medium CTR, shared round: a whole pass + the ragged rest
    assert P(2048 + 1000) == [(5, 2048), (8, 1000)]
    assert P(2048 + 1025) == [(5, 3073)]                                     # the rest is worth a pass of its own
    assert P(4 * 2048 + 352) == [(5, 8192), (6, 352)]
    for count in list(range(1, 5000, 37)) + [16384, 16384 + 500]:
        plan = P(count)
        assert sum(n for _, n in plan) == count and all(n > 0 for _, n in plan) and len(plan) <= 2
    assert P(300, num_cus=64) == [(5, 300)] and P(190, num_cus=64) == [(6, 190)]


def test_round4_entry_points_reject_bad_arguments_without_a_gpu():
    """the entry points added in round 4 fail with a message (never crash, never touch a device) on null contexts / pointers"""
    import txhip
    L = txhip.lib()
    assert L.tx_timing_size() == C.sizeof(txhip.TxTiming) == 120        # round 5: + timing_failed
    assert not L.tx_device_alloc(None, 1024) and b"null context" in L.tx_last_error()
    assert L.tx_device_free(None, None) < 0 and L.tx_device_upload(None, None, None, 8) < 0 and L.tx_device_download(None, None, None, 8) < 0
    assert L.tx_max_u16_io(None, None, 4, None, 1, 1) < 0 and b"tx_max_u16" in L.tx_last_error()
    assert L.tx_inner_product_u16_io(None, None, 4, None, 1, 1) < 0
    L.tx_stage_circuit_bootstrap.restype = C.c_int
    assert L.tx_stage_circuit_bootstrap(None, None, 1, None) < 0
    # the CTR column grouping refuses a block count whose indices do not fit its int arrays (before it touches them)
    iv = (C.c_uint8 * 16)()
    assert L.tx_ctr_shared_columns(iv, 0, (1 << 29) + 1, None, None) < 0
    one = (C.c_int * 4)(); two = (C.c_int * 4)()
    assert L.tx_ctr_shared_columns(iv, 0, (1 << 29) + 1, one, two) < 0 and b"does not fit" in L.tx_last_error()


def test_round5_entry_points_without_a_gpu():
    """the entry points added in round 5 (pipelined host output, column-sharing switch, page-locked memory, device count) fail
    with a message on null arguments and never crash; the chunk size is the documented constant"""
    import txhip
    L = txhip.lib()
    assert L.tx_chunk_blocks() == 128
    cb = txhip.CHUNK_FN(lambda u, a, b: None)
    assert L.tx_transcipher_chunks(None, None, 1, None, None, 0, cb, None) < 0 and b"null argument" in L.tx_last_error()
    assert L.tx_set_ctr_column_sharing(None, 0) < 0 and b"null context" in L.tx_last_error()
    assert L.tx_host_free(None) == 0
    assert L.tx_set_decomposer_tie_rule(None, 0) < 0 and b"null context" in L.tx_last_error()
    assert L.tx_reserve(None, 4) < 0 and L.tx_host_register(None, 8) < 0 and L.tx_host_unregister(None) == 0
    n = L.tx_device_count()
    import torch
    if not torch.cuda.is_available():
        assert n <= 0 or n == torch.cuda.device_count()


def test_product_library_has_no_experiment_surface():
    """the shipped library holds neither the superseded kernels nor the TX_* environment knobs; the
    experiment build (same sources, -DTXD_EXPERIMENTS) holds both and exports the same C ABI"""
    import subprocess
    import txhip
    pkg = os.path.join(ROOT, "temp-fhe-transciphering_amd")
    prod = open(os.path.join(pkg, "libtfhe_mi355x.so"), "rb").read()
    exp = open(os.path.join(pkg, "libtfhe_mi355x_exp.so"), "rb").read()
    for knob in (b"TX_BR_VARIANT", b"TX_TRACE_VARIANT", b"TX_SS_VARIANT", b"TX_LUT_VARIANT", b"TX_BR5_WGS", b"TX_TRSS_WGS_PER_CU", b"TX_CTR_SHARE"):
        assert knob not in prod, knob
        assert knob in exp, knob
    assert b"getenv" not in prod                       # the product library reads no environment at all
    for kernel in (b"k_blind_rotate4", b"k_blind_rotate7", b"k_lut_eval4", b"15k_extract_trace", b"15k_scheme_switch", b"7k_trace"):
        assert kernel not in prod, kernel
        assert kernel in exp, kernel
    for kernel in (b"k_blind_rotate5", b"k_blind_rotate6", b"k_blind_rotate8", b"k_trace_scheme_switch4", b"k_lut_eval", b"k_extract_trace4"):
        assert kernel in prod, kernel
    Lx = txhip.lib(experiments=True)
    for s in declared_symbols():
        assert hasattr(Lx, s), s


def test_stage_device_list_parsing(tmp_path):
    """TX_DEVICES of the server stages (host/stages.cpp device_list): a malformed list is a usage error (exit 1, message) before
    any GPU work; the key files are checked before that"""
    import subprocess
    exe = os.path.join(ROOT, "temp-fhe-transciphering_amd", "bin", "server_encrypted_aes_decryption")
    r = subprocess.run([exe, "0"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "cannot read datasets/toy/db.hex" in r.stderr
    os.makedirs(tmp_path / "datasets" / "toy")
    (tmp_path / "datasets" / "toy" / "db.hex").write_text("00" * 16)
    r = subprocess.run([exe, "0"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "public_keys/ksk.bin" in r.stderr


def test_server_stage_fails_loudly_on_a_bad_device_list_and_without_a_gpu(tmp_path):
    """the in-process sharding of the server stages (host/stages.cpp): with real key files from the CLIENT stages (CPU only), a
    malformed TX_DEVICES is exit status 1 with a message before any GPU work, and on a box without a HIP device the stage exits 1
    with the no-fallback message -- never a crash, never a silent CPU path"""
    import subprocess
    import torch
    binp = os.path.join(ROOT, "temp-fhe-transciphering_amd", "bin")
    os.makedirs(tmp_path / "datasets" / "toy")
    (tmp_path / "datasets" / "toy" / "aes_key.hex").write_text("dc937b59892604f5a86ac96936cd7ff0")
    (tmp_path / "datasets" / "toy" / "db.hex").write_text("e907e226bb17b0e43f99940eab33ed6a")
    env = dict(os.environ, TX_SEED="77")
    for stage in ("client_preprocess", "client_key_generation", "client_encode_encrypt", "server_preprocess_dataset"):
        r = subprocess.run([os.path.join(binp, stage), "0"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, stage + r.stderr
    exe = os.path.join(binp, "server_encrypted_aes_decryption")
    for bad, msg in (("0,,1", "not a comma-separated list"), ("a", "not a comma-separated list"), ("0,", "ends with a comma"), ("", "TX_DEVICES is empty")):
        r = subprocess.run([exe, "0"], cwd=tmp_path, capture_output=True, text=True, env=dict(env, TX_DEVICES=bad), timeout=120)
        assert r.returncode == 1 and msg in r.stderr, (bad, r.returncode, r.stderr)
    if not torch.cuda.is_available():
        for dv in (None, "0", "0,1"):
            e2 = dict(env) if dv is None else dict(env, TX_DEVICES=dv)
            e2.pop("TX_DEVICE", None)
            r = subprocess.run([exe, "0"], cwd=tmp_path, capture_output=True, text=True, env=e2, timeout=120)
            assert r.returncode == 1 and ("no HIP device" in r.stderr or "no such HIP device" in r.stderr), (dv, r.returncode, r.stderr)
            assert not os.path.exists(tmp_path / "io" / "toy" / "ciphertext_aes_download" / "result.bin")      # no half-written result
        r = subprocess.run([os.path.join(binp, "server_encrypted_compute"), "0"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 1
