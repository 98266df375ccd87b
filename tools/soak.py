"""exploratory soak: several key seeds x an instance of BLOCKS AES blocks (default 64 = medium; 8 = the per-GPU share of an 8-GPU run:
k_blind_rotate8; 6, 4, 3 = the fair cooperative kernel), ECB and CTR, product-only path:  tools/soak.py first_seed last_seed [BLOCKS] [tfhe]
(a fourth argument `tfhe`: every decomposition under tfhe-rs' tie rule, tx_set_decomposer_tie_rule(ctx, TX_TIE_TFHE_RS) -- the rule whose
top-digit mean costs the bootstrap 3.7 sigma of margin: wrong plaintext bits are EXPECTED at a rate of about one bootstrap in 10^6)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np, txhip, txclient
key = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0"); iv = bytes.fromhex("64b92e64992534e495b9da182409427c")
p = txhip.aes_tight()
bad = 0
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 64
TFHE = len(sys.argv) > 4 and sys.argv[4] == "tfhe"
wrong_bits = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    ck = txclient.ClientKeys(p, seed=1000 + seed)
    eng = txhip.Engine(p, 0)
    if TFHE: eng.set_decomposer_tie_rule(True)
    eng.load_eval_keys(ck.bsk, ck.ksk, ck.auto, ck.ss)
    eng.load_round_luts(ck.round_luts(key, seed=seed + 5)); eng.load_ctr_luts(ck.ctr_luts(key, seed=seed + 6))
    rng = np.random.default_rng(seed)
    pt = rng.integers(0, 256, 16 * NB, dtype=np.uint8).tobytes()
    o1 = eng.transcipher(txclient.aes_ecb_encrypt(key, pt))
    o2 = eng.transcipher(txclient.aes_ctr_crypt(key, iv, pt), iv=iv)
    want = np.frombuffer(pt, np.uint8).view(">u2").astype(np.uint16)
    wbits = np.unpackbits(np.frombuffer(pt, np.uint8))
    b1, b2 = ck.decrypt_bits(o1), ck.decrypt_bits(o2)
    wrong_bits += int((b1 != wbits).sum()) + int((b2 != wbits).sum())
    ok1 = np.array_equal(txclient.bits_to_u16(b1), want)
    ok2 = np.array_equal(txclient.bits_to_u16(b2), want)
    mx = eng.max_u16(o1[:16 * min(64, 8 * NB)])
    okm = int(txclient.bits_to_u16(ck.decrypt_bits(mx))[0]) == int(want[:min(64, 8 * NB)].max())
    print("blocks", NB, "seed", seed, "rule", "tfhe-rs" if TFHE else "balanced", "ecb", ok1, "ctr", ok2, "max", okm, eng.last_blind_rotate_kernel(), flush=True)
    bad += (not ok1) + (not ok2) + (not okm)
    eng.close()
print("failures:", bad, " wrong plaintext bits of the AES results:", wrong_bits, "of", 2 * 128 * NB * (int(sys.argv[2]) - int(sys.argv[1])))
