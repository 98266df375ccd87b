#!/usr/bin/env python3
"""Counterpart of the reference's harness/run_submission.py for this repo's stage executables.

Same stage order, argv (<size>) and cwd contract (harness/run_submission.py:38-39, 68-115), same
dataset / key derivation (generate_dataset.py:26-29, aes_keygen_and_encrypt.py:34,42-55), same
expected files and exact-equality verification (cleartext_impl.py:57-70, verify_result.py:31-42),
same results.json schema (utils.py:130-141).  The Python reference cannot travel to the GPU box, so
its logic is restated here; AES comes from this repo's client library.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
BIN = os.path.join(os.path.dirname(HERE), "bin")
NAMES = ["toy", "small", "medium", "large"]
DB_BOUND = [8, 64, 512, 4096]               # harness/params.py:39; "large" (size 3) is this repo's definition, x8 again

STAGES = [
    (3, "client_preprocess"), (4, "client_key_generation"), (5, "client_encode_encrypt"),
    (6, "server_preprocess_dataset"), (7, "server_encrypted_aes_decryption"), (8, "server_encrypted_compute"),
    (9, "client_decrypt_decode_aes_decryption"), (10, "client_postprocess_aes_decryption"),
    (11, "client_decrypt_decode"), (12, "client_postprocess"),
]


def du(path):
    total = 0
    for d, _, files in os.walk(path):
        total += sum(os.path.getsize(os.path.join(d, f)) for f in files)
    return total


def run(size, root, seed=None, dataset_seed=None, fhe_seed=None, mini_workload=0, server_ranks=0, backend="nccl"):
    """server_ranks > 0: the two server stages are replaced by ONE run of run_sharded.py --io over that many rank processes
    (one per GPU; --backend gloo lets them share a card), which reads and writes the same files under io/<size>/"""
    import numpy as np
    import txclient
    name = NAMES[size]
    data, io = os.path.join(root, "datasets", name), os.path.join(root, "io", name)
    os.makedirs(data, exist_ok=True)
    if os.path.exists(io):
        shutil.rmtree(io)                                   # run_submission.py:41-45
    os.makedirs(io)
    per_stage, t_last = {}, time.time()

    def log(step, label):
        nonlocal t_last
        now = time.time()
        per_stage[f"{step}: {label}"] = round(now - t_last, 4)
        print(f"{time.strftime('%H:%M:%S')} [harness] {step}: {label} completed (elapsed: {now - t_last:.4f}s)", flush=True)
        t_last = now

    # 1. dataset
    if dataset_seed is not None:
        np.random.seed(dataset_seed)
    db = np.random.randint(0, 65536, size=DB_BOUND[size], dtype=np.uint16)
    open(os.path.join(data, "db.txt"), "w").write("\n".join(str(v) for v in db) + "\n")
    log(1, "Dataset generation")
    # 2. AES key, IV, ciphertext
    key = hashlib.sha256(str(seed).encode()).digest()[:16]
    iv = hashlib.sha256(b"iv" + str(seed).encode()).digest()[:16]
    pt = b"".join(int(v).to_bytes(2, "big") for v in db)
    if size == 0:
        ct = txclient.aes_ecb_encrypt(key, pt)
    else:
        ct = txclient.aes_ctr_crypt(key, iv, pt)
        open(os.path.join(data, "aes_iv.hex"), "w").write(iv.hex())
    open(os.path.join(data, "aes_key.hex"), "w").write(key.hex())
    open(os.path.join(data, "db.hex"), "w").write(ct.hex())
    log(2, "AES Key generation and message encryption with AES")
    # cleartext implementation
    dec = txclient.aes_ecb_decrypt(key, ct) if size == 0 else txclient.aes_ctr_crypt(key, iv, ct)
    vals = [int.from_bytes(dec[i:i + 2], "big") for i in range(0, len(dec), 2)]
    half = len(vals) // 2
    open(os.path.join(data, "expected_aes.txt"), "w").write("\n".join(map(str, vals)) + "\n")
    open(os.path.join(data, "max_value.txt"), "w").write(f"{max(vals)}\n")
    open(os.path.join(data, "inner_product.txt"), "w").write(
        f"{sum((x * y) % 65536 for x, y in zip(vals[:half], vals[half:])) % 65536}\n")
    log(2, "Cleartext implementation")
    env = dict(os.environ)
    env["TX_MINI_WORKLOAD"] = str(mini_workload)          # utils.py:52-53: 0 = max, 1 = dot product
    if fhe_seed:
        env["TX_SEED"] = str(fhe_seed)
    bandwidth = {}
    for step, stage in STAGES:
        if server_ranks and stage == "server_encrypted_compute":
            continue
        if server_ranks and stage == "server_encrypted_aes_decryption":
            for d in ("ciphertext_aes_download", "ciphertexts_download"):
                os.makedirs(os.path.join(io, d), exist_ok=True)
            subprocess.run([sys.executable, os.path.join(HERE, "run_sharded.py"), str(size), "--io", root, "--gpus", str(server_ranks),
                            "--backend", backend, "--mini_workload", str(mini_workload)] + (["--force-dist"] if server_ranks == 1 else []),
                           cwd=root, check=True, env=env)
            log(step, "server_encrypted_aes_decryption + server_encrypted_compute (run_sharded.py, %d ranks)" % server_ranks)
            bandwidth["Encrypted AES decryption result"] = du(os.path.join(io, "ciphertext_aes_download"))
            bandwidth["Encrypted compute result"] = du(os.path.join(io, "ciphertexts_download"))
            continue
        subprocess.run([os.path.join(BIN, stage), str(size)], cwd=root, check=True, env=env)
        log(step, stage)
        if stage == "client_encode_encrypt":
            bandwidth["Public and evaluation keys"] = du(os.path.join(io, "public_keys"))
            bandwidth["Encrypted database"] = du(os.path.join(io, "ciphertexts_upload"))
        if stage == "server_encrypted_aes_decryption":
            bandwidth["Encrypted AES decryption result"] = du(os.path.join(io, "ciphertext_aes_download"))
        if stage == "server_encrypted_compute":
            bandwidth["Encrypted compute result"] = du(os.path.join(io, "ciphertexts_download"))

    def same(a, b):                                          # verify_result.py:31-42
        return [int(x) for x in open(a).read().split()] == [int(x) for x in open(b).read().split()]

    ok_aes = same(os.path.join(data, "expected_aes.txt"), os.path.join(io, "result_aes.txt"))
    expect = "inner_product.txt" if mini_workload == 1 else "max_value.txt"       # run_submission.py:125-139
    ok_max = same(os.path.join(data, expect), os.path.join(io, "result.txt"))
    print(f"[harness] AES decryption: {'PASS' if ok_aes else 'FAIL'}; {'inner product' if mini_workload == 1 else 'max'}: {'PASS' if ok_max else 'FAIL'}", flush=True)
    meas = os.path.join(root, "measurements", name)
    os.makedirs(meas, exist_ok=True)
    json.dump({"total_latency_s": round(sum(per_stage.values()), 4), "per_stage": per_stage, "bandwidth": bandwidth},
              open(os.path.join(meas, "results.json"), "w"), indent=2)
    return ok_aes and ok_max


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("size", type=int, choices=[0, 1, 2, 3])
    ap.add_argument("--root", default=os.getcwd())
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--fhe_seed", type=int, default=0)
    ap.add_argument("--mini_workload", type=int, default=0, choices=[0, 1])
    ap.add_argument("--server_ranks", type=int, default=0,
                    help="run the server side as this many rank processes of run_sharded.py --io instead of the two server stage "
                         "executables (which shard across TX_DEVICES in-process)")
    ap.add_argument("--backend", default="nccl", help="--server_ranks: torch.distributed backend (gloo: ranks may share a GPU)")
    a = ap.parse_args()
    ds = None
    if a.seed is not None:                                  # run_submission.py:50-54
        import numpy as np
        ds = int(np.random.default_rng(a.seed).integers(0, 0x7fffffff))
    sys.exit(0 if run(a.size, a.root, None, ds, a.fhe_seed, a.mini_workload, a.server_ranks, a.backend) else 1)
