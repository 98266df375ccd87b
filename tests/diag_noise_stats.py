"""diagnostic (uses the oracle, hence under tests/; not collected by pytest): distribution of the mod-switched rotation error entering the blind rotation"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, txhip, txclient
import oracle_lib as O
key = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0"); iv = bytes.fromhex("64b92e64992534e495b9da182409427c")
p = txhip.aes_tight()
ck = txclient.ClientKeys(p, seed=1002)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(ck.bsk, ck.ksk, ck.auto, ck.ss)
eng.load_round_luts(ck.round_luts(key, seed=7))
rng = np.random.default_rng(2)
ct = rng.integers(0, 256, 16 * 64, dtype=np.uint8).tobytes()
sk_small = ck.lwe_sk.astype(np.int64)
for rd in (1, 3, 8):
    st = eng.transcipher_partial(ct, rd)
    bits = ck.decrypt_bits(st)
    ph = O.lwe_phase(ck.glwe_sk, st)
    e = O.signed(ph - (bits.astype(np.uint64) << np.uint64(63))).astype(np.float64)
    sm = eng.lwe_keyswitch(st)
    at = (((sm >> np.uint64(55)) + np.uint64(1)) >> np.uint64(1)).astype(np.int64) << 3        # modswitch
    pp = (at[:, -1] - (at[:, :-1] * sk_small).sum(axis=1)) % 2048
    err = ((pp - bits.astype(np.int64) * 1024 + 1024) % 2048) - 1024
    s = err.std()
    print(f"round {rd}: state noise std 2^{np.log2(e.std()):.2f} max 2^{np.log2(np.abs(e).max()):.2f}; rotation error std {s:.1f} max {np.abs(err).max()} of 512 ({512/s:.2f} sigma), kurtosis {((err-err.mean())**4).mean()/s**4:.2f}, n={err.size}, >384: {(np.abs(err)>384).sum()} >448: {(np.abs(err)>448).sum()}", flush=True)
