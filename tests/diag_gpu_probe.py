"""diagnostic (uses the oracle's client-side helpers, hence under tests/; not collected by pytest): full transcipher on the GPU, timing per kernel family, decrypt check"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np
import oracle_lib as O, txhip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
nblocks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8]
p = O.aes_tight(n=n)
t = time.time(); keys = O.Keys(p, seed=5); print("keygen", time.time() - t, flush=True)
aes_key = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0"); rk = O.aes_expand(aes_key)
luts = keys.round_luts(aes_key)
eng = txhip.Engine(txhip.aes_tight(n=n), 0)
t = time.time(); eng.load_eval_keys(keys.bsk, keys.ksk, keys.auto, keys.ss); eng.load_round_luts(luts); print("load", time.time() - t, flush=True)
rng = np.random.default_rng(1)
for nb in nblocks:
    pt = rng.integers(0, 256, 16 * nb, dtype=np.uint8).tobytes()
    ct = b"".join(O.aes_encrypt_block(rk, pt[16 * i:16 * i + 16]) for i in range(nb))
    for rep in range(2):
        t = time.time(); out = eng.transcipher(ct); dt = time.time() - t
        tm = eng.timing()
        print(f"nb={nb} wall {dt:.3f}s blocks/s {nb/dt:.2f}", {k: round(v, 2) if isinstance(v, float) else v for k, v in tm.items()}, flush=True)
    got = O.bits_msb_to_bytes(O.lwe_decrypt_bits(keys.glwe_sk, out))
    print("decrypt ok:", got == pt, flush=True)
