#!/usr/bin/env python3
"""One instance of the harness workload block-sharded over the GPUs of one node:
transcipher + post-transcipher max + inner product, verified on rank 0 against the cleartext.

    python temp-fhe-transciphering_amd/python/run_sharded.py <size> --gpus N [--backend nccl|gloo]
or  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        temp-fhe-transciphering_amd/python/run_sharded.py <size> [--backend nccl|gloo]

(the first form starts its own N rank processes before it has made any GPU call and fails fast on a node with fewer GPUs;
--force-dist sends a single rank through the process group and every collective as well)

--io ROOT: SERVER mode behind the harness's io/ contract (harness/run_submission.py:91-99): instead of generating a workload and
keys of its own, every rank reads datasets/<size>/{db.hex,aes_iv.hex} and the files the client stages wrote under
ROOT/io/<size>/ (public_keys/*.bin, ciphertexts_upload/trans_key.bin), and rank 0 writes what the two server stages write:
ciphertext_aes_download/result.bin and ciphertexts_download/result.bin (mini-workload --mini_workload / TX_MINI_WORKLOAD).  It is the
multi-PROCESS form of what `server_encrypted_aes_decryption` + `server_encrypted_compute` do in-process over TX_DEVICES.

One process per GPU, a full replica of the evaluation keys on each (same FHE seed, no broadcast), contiguous
slabs of AES blocks per rank; no collective on the data path of the transcipher.  Exchanges: one gather of the
result LWEs, one gather of the partial maxima, and for the inner product one point-to-point step that brings
every rank the remote operands of its pairs plus one gather of the partial sums (txdist.py).  With --backend
gloo several ranks may share one GPU (rehearsal on a one-GPU box).  Dataset, key and IV derivation follow
run_submission.py (harness/generate_dataset.py:26-29, aes_keygen_and_encrypt.py:34,42-55); sizes 1..3 are CTR.
"""
import argparse
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
DB_BOUND = [8, 64, 512, 4096]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("size", type=int, choices=[0, 1, 2, 3])
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--fhe_seed", type=int, default=99)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--n", type=int, default=768, help="LWE dimension (AES_TIGHT: 768)")
    ap.add_argument("--values", type=int, default=0,
                    help="use only the first VALUES u16 of the instance (a multiple of 8 and even): rehearses a per-GPU "
                         "share with fewer ranks than the node has GPUs")
    ap.add_argument("--gpus", type=int, default=0,
                    help="started as a plain command (no RANK in the environment): launch this many ranks as child processes")
    ap.add_argument("--force-dist", action="store_true", help="a single rank still initialises the process group and takes every collective")
    ap.add_argument("--io", default=None, metavar="ROOT",
                    help="server mode: inputs from ROOT/datasets/<size>/ and ROOT/io/<size>/ as the client stages left them, results "
                         "written where the two server stages write them (no workload or keys generated here)")
    ap.add_argument("--mini_workload", type=int, default=int(os.environ.get("TX_MINI_WORKLOAD", "0")), choices=[0, 1],
                    help="--io: which result goes to ciphertexts_download/result.bin (0 = max, 1 = inner product; both are computed)")
    args = ap.parse_args()

    import txdist
    if args.gpus > 1 and "RANK" not in os.environ:
        def prepare():
            sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
            import __graft_entry__
            __graft_entry__.build()
        sys.exit(txdist.launch_ranks(__file__, sys.argv[1:], args.gpus, backend=args.backend, prepare=prepare, name="run_sharded.py"))

    import numpy as np
    import torch
    import torch.distributed as dist
    import txhip
    import txclient

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("run_sharded.py needs a HIP device; there is no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where collective buffers live
    if args.gpus and args.gpus != world:
        raise SystemExit(f"run_sharded.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(txdist.free_port()) if world == 1 else "29513")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    d = dist if use_dist else None

    p = txhip.aes_tight(n=args.n)
    name = ["toy", "small", "medium", "large"][args.size]
    if args.io:
        # server mode: what a server that received the client's files does -- nothing is generated here
        data, io_dir = os.path.join(args.io, "datasets", name), os.path.join(args.io, "io", name)
        ct = bytes.fromhex(open(os.path.join(data, "db.hex")).read().strip())
        iv = bytes.fromhex(open(os.path.join(data, "aes_iv.hex")).read().strip()) if args.size else None
        if not ct or len(ct) % 16:
            raise SystemExit("db.hex must hold whole 16-byte blocks")
        keys = txclient.StageKeys(p, io_dir)
        exp = os.path.join(data, "expected_aes.txt")        # the harness's cleartext result, if it is there (a one-machine run)
        db = np.array([int(x) for x in open(exp).read().split()], np.uint16) if (keys.glwe_sk is not None and os.path.exists(exp)) else None
        n_blocks, n_values = len(ct) // 16, len(ct) // 2
    else:
        np.random.seed(args.seed)
        db = np.random.randint(0, 65536, size=DB_BOUND[args.size], dtype=np.uint16)
        if args.values:
            if args.values % 8 or args.values > len(db):
                raise SystemExit("--values must be a multiple of 8 (whole AES blocks) within the instance")
            db = db[: args.values]
        key = hashlib.sha256(str(args.seed).encode()).digest()[:16]
        iv = hashlib.sha256(b"iv" + str(args.seed).encode()).digest()[:16] if args.size else None
        pt = b"".join(int(v).to_bytes(2, "big") for v in db)
        ct = txclient.aes_ctr_crypt(key, iv, pt) if args.size else txclient.aes_ecb_encrypt(key, pt)
        n_blocks, n_values = len(ct) // 16, len(db)
        keys = txclient.ClientKeys(p, seed=args.fhe_seed)
        keys.trans_key = keys.ctr_luts(key, seed=args.fhe_seed + 1) if args.size else keys.round_luts(key, seed=args.fhe_seed + 1)
    eng = txhip.Engine(p, device=dev_index)
    eng.load_eval_keys(keys.bsk, keys.ksk, keys.auto, keys.ss)
    (eng.load_ctr_luts if args.size else eng.load_round_luts)(np.ascontiguousarray(keys.trans_key))
    words = eng.big

    def to_np(t):
        return np.ascontiguousarray(t.cpu().numpy().view(np.uint64))

    kept = {}

    def local_fn(ct_slab, iv_slab, n_local, n_pad):
        # the slab's result LWEs are written to HBM and stay there: the mini-workloads below read them in place
        # (tx_max_u16_io / tx_inner_product_u16_io, in_is_device = 1); only the gloo rehearsal stages them through the host
        out = torch.zeros((n_pad * 128, words), dtype=torch.int64, device=dev)
        if n_local:
            eng.transcipher(ct_slab, iv_slab, out_device_ptr=out.data_ptr())
        kept["mine"], kept["n"] = out, n_local
        return out if cdev == dev else out.cpu()

    def on_card(fn):
        """txdist's local_*_fn over the engine's device entry points: tensor [m * 16, words] in, [16, words] out, on cdev"""
        def run(t):
            t = t.to(dev).contiguous()
            o = torch.empty((16, words), dtype=torch.int64, device=dev)
            fn(t.data_ptr(), t.shape[0] // 16, o.data_ptr())
            return o.to(cdev)
        return run

    def sync():
        if d is not None:
            d.barrier()
        torch.cuda.synchronize()

    t = {}
    sync(); t0 = time.perf_counter()
    res = txdist.transcipher_sharded(local_fn, ct, iv, rank, world, d)
    sync(); t["transcipher_s"] = time.perf_counter() - t0
    mine, n_local_values = kept["mine"].to(cdev), kept["n"] * 8
    br_kernel = eng.last_blind_rotate_kernel()             # of this rank's share of the transcipher

    t0 = time.perf_counter()
    mx = txdist.max_sharded(on_card(eng.max_u16_device), mine, n_local_values, rank, world, d)
    sync(); t["max_s"] = time.perf_counter() - t0

    ranges = [(8 * s, 8 * c) for s, c in (txdist.shard(n_blocks, world, r) for r in range(world))]
    t0 = time.perf_counter()
    ip = txdist.inner_product_sharded(on_card(eng.inner_product_u16_device), mine[: n_local_values * 16],
                                      ranges, n_values, rank, world, d)
    sync(); t["inner_product_s"] = time.perf_counter() - t0

    ok = True
    if rank == 0 and args.io:
        # what the two server stages write (host/stages.cpp): the result LWEs and the mini-workload's 16 LWEs
        txclient.write_words(os.path.join(io_dir, "ciphertext_aes_download", "result.bin"), to_np(res))
        txclient.write_words(os.path.join(io_dir, "ciphertexts_download", "result.bin"), to_np(ip if args.mini_workload == 1 else mx))
    if rank == 0 and db is None:
        print(json.dumps({"size": args.size, "values": n_values, "blocks": n_blocks, "ranks": world, "backend": args.backend,
                          "checks": None, "blind_rotate_kernel": br_kernel, **{k: round(v, 4) for k, v in t.items()}}), flush=True)
    elif rank == 0:
        vals = [int(v) for v in db]
        half = len(vals) // 2
        got = txclient.bits_to_u16(keys.decrypt_bits(to_np(res)))
        got_max = int(txclient.bits_to_u16(keys.decrypt_bits(to_np(mx)))[0])
        got_ip = int(txclient.bits_to_u16(keys.decrypt_bits(to_np(ip)))[0])
        exp_ip = sum((vals[i] * vals[i + half]) % 65536 for i in range(half)) % 65536
        checks = {"aes": bool(np.array_equal(got, db)), "max": got_max == max(vals), "inner_product": got_ip == exp_ip}
        ok = all(checks.values())
        print(json.dumps({"size": args.size, "values": n_values, "blocks": n_blocks, "ranks": world, "backend": args.backend,
                          "checks": checks, "blind_rotate_kernel": br_kernel, **{k: round(v, 4) for k, v in t.items()}}), flush=True)
    if d is not None:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
        d.broadcast(flag, src=0)
        ok = bool(flag.item())
        d.destroy_process_group()
    eng.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
