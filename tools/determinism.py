"""exploratory: the same ciphertexts through the product path REPS times; every output must equal the first bit for bit (the kernels'
priorities, fair tables and launch-size dispatch may change timing, never results):  tools/determinism.py [REPS] [BLOCKS ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np, txhip, txclient
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sizes = [int(x) for x in sys.argv[2:]] or [64, 8, 3]
key = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0"); iv = bytes.fromhex("64b92e64992534e495b9da182409427c")
p = txhip.aes_tight()
ck = txclient.ClientKeys(p, seed=4242)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(ck.bsk, ck.ksk, ck.auto, ck.ss)
eng.load_round_luts(ck.round_luts(key, seed=7)); eng.load_ctr_luts(ck.ctr_luts(key, seed=8))
bad = 0
for nb in sizes:
    pt = np.random.default_rng(nb).integers(0, 256, 16 * nb, dtype=np.uint8).tobytes()
    for mode, ct, kw in (("ecb", txclient.aes_ecb_encrypt(key, pt), {}), ("ctr", txclient.aes_ctr_crypt(key, iv, pt), {"iv": iv})):
        first = np.array(eng.transcipher(ct, **kw), copy=True)
        same = sum(bool(np.array_equal(first, eng.transcipher(ct, **kw))) for _ in range(reps))
        want = np.frombuffer(pt, np.uint8).view(">u2").astype(np.uint16)
        ok = np.array_equal(txclient.bits_to_u16(ck.decrypt_bits(first)), want)
        print("blocks", nb, mode, "identical repeats", same, "of", reps, "| decrypts exactly:", ok, "|", eng.last_blind_rotate_kernel(), flush=True)
        bad += (same != reps) + (not ok)
print("failures:", bad)
