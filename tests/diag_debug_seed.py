"""diagnostic (uses the oracle, hence under tests/; not collected by pytest): rerun one failing block of a soak seed on GPU and oracle"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, txhip, txclient, txdist
import oracle_lib as O
key = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0"); iv = bytes.fromhex("64b92e64992534e495b9da182409427c")
p = txhip.aes_tight(); seed = 2
ck = txclient.ClientKeys(p, seed=1000 + seed)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(ck.bsk, ck.ksk, ck.auto, ck.ss)
cl = ck.ctr_luts(key, seed=seed + 6); eng.load_ctr_luts(cl)
rng = np.random.default_rng(seed)
pt = rng.integers(0, 256, 16 * 64, dtype=np.uint8).tobytes()
ct = txclient.aes_ctr_crypt(key, iv, pt)
o2 = eng.transcipher(ct, iv=iv)
bits = ck.decrypt_bits(o2)
want_bits = np.unpackbits(np.frombuffer(pt, np.uint8))
wrong = np.nonzero(bits != want_bits)[0]
print("wrong bit positions:", wrong.tolist(), "blocks", sorted(set((wrong // 128).tolist())))
# phases of all outputs: distance to nearest multiple of 2^63
sk = ck.glwe_sk
ph = O.lwe_phase(sk, o2)
err = O.signed(ph - (bits.astype(np.uint64) << np.uint64(63)))
print("max |noise| log2 overall", np.log2(np.abs(err).max()), "at", int(np.abs(err).argmax()))
for wb in wrong[:4]:
    print("bit", wb, "noise log2", np.log2(abs(int(err[wb])) + 1))
b = int(wrong[0] // 128)
# rerun that block alone on GPU and oracle
g1 = eng.transcipher(ct[16 * b:16 * b + 16], iv=txdist.iv_add(iv, b))
print("block alone equals batch slice:", np.array_equal(g1, o2[128 * b:128 * b + 128]))
class K: pass
k = K(); k.p = O.aes_tight(); k.bsk = ck.bsk; k.ksk = ck.ksk; k.auto = ck.auto; k.ss = ck.ss
octx = O.Ctx(k)
t = time.time(); oo = octx.transcipher_ctr(cl, txdist.iv_add(iv, b), ct[16 * b:16 * b + 16]); print("oracle s", time.time() - t)
print("gpu == oracle on failing block:", np.array_equal(oo, g1))
