#!/usr/bin/env python3
"""Generates tests/golden/harness_fixtures.json.

Data only: the reference's own harness/generate_dataset.py (numpy-only) is RUN from
/root/reference with a scratch cwd to produce db.txt for (size, seed); the AES key / IV follow
harness/aes_keygen_and_encrypt.py:34,52 (sha256 of str(seed)); the ciphertext is produced with the
system `openssl` (pyaes, which the harness uses, is not installed here): ECB for size 0, CTR for
sizes >= 1 (aes_keygen_and_encrypt.py:45-55).  max / inner product follow cleartext_impl.py:61-70.
The harness passes no --seed to aes_keygen_and_encrypt.py (run_submission.py:59), hence seed None.
"""
import hashlib, json, os, struct, subprocess, sys, tempfile

REF = "/root/reference/harness/generate_dataset.py"
NAMES = ["toy", "small", "medium"]


def openssl(mode, key, iv, data):
    cmd = ["openssl", "enc", "-e", f"-aes-128-{mode}", "-K", key.hex(), "-nopad"]
    if iv is not None:
        cmd += ["-iv", iv.hex()]
    return subprocess.run(cmd, input=data, stdout=subprocess.PIPE, check=True).stdout


def main():
    out = []
    for size in (0, 1, 2):
        for seed in (123, 7):
            with tempfile.TemporaryDirectory() as td:
                subprocess.run([sys.executable, REF, str(size), "--seed", str(seed)], cwd=td, check=True)
                db = [int(x) for x in open(os.path.join(td, "datasets", NAMES[size], "db.txt")).read().split()]
            key = hashlib.sha256(str(None).encode()).digest()[:16]
            iv = hashlib.sha256(b"iv" + str(None).encode()).digest()[:16]
            pt = struct.pack(">" + "H" * len(db), *db)
            ct = openssl("ecb", key, None, pt) if size == 0 else openssl("ctr", key, iv, pt)
            half = len(db) // 2
            out.append({
                "size": size, "dataset_seed": seed, "db": db, "aes_key": key.hex(), "aes_iv": iv.hex(),
                "db_hex": ct.hex(), "max_value": max(db),
                "inner_product": sum((x * y) % 65536 for x, y in zip(db[:half], db[half:])) % 65536,
            })
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness_fixtures.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
