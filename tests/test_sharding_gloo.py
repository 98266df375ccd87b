"""N > 1 path on CPU: world_size-2 gloo run of the block-sharding + gather logic (txdist), with the
oracle standing in for the per-rank GPU engine (the tests may use the oracle; the product may not)."""
import json
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_shard_covers_every_block_once():
    sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
    import txdist
    for n in (0, 1, 3, 8, 64, 65):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                s, c = txdist.shard(n, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(n))
    assert txdist.iv_add(bytes([0xFF] * 16), 1) == bytes(16)
    assert txdist.iv_add(bytes(15) + b"\xff", 1) == bytes(14) + b"\x01\x00"


def _worker(rank, world, port, mode, q, nb=3):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import txdist
    fx = [f for f in json.load(open(os.path.join(HERE, "golden", "harness_fixtures.json")))
          if f["size"] == 1 and f["dataset_seed"] == 123][0]
    key, iv = bytes.fromhex(fx["aes_key"]), bytes.fromhex(fx["aes_iv"])
    p = O.aes_tight(n=256)
    keys = O.Keys(p, seed=33)                       # same seed on both ranks: identical key replicas
    ctx = O.Ctx(keys, threads=4 if world <= 2 else 1)
    # nb = 3 on two ranks is ragged (2 + 1 blocks); nb = 5 on eight ranks leaves ranks 5..7 without a block
    pt = b"".join(int(v).to_bytes(2, "big") for v in fx["db"][: 8 * nb])
    if mode == "ctr":
        ct, luts, ivv = bytes.fromhex(fx["db_hex"])[: 16 * nb], keys.ctr_luts(key), iv
    else:
        rk = O.aes_expand(key)
        ct, luts, ivv = b"".join(O.aes_encrypt_block(rk, pt[i:i + 16]) for i in range(0, len(pt), 16)), keys.round_luts(key), None

    kept = {}

    def local_fn(ct_slab, iv_slab, n_local, n_pad):
        out = torch.zeros((n_pad * 128, ctx.big), dtype=torch.int64)
        if n_local:
            if world > 2:
                # world 8 checks the SHARDING (slabs, empty ranks, gather order, partial maxima), not the cipher again: the slab's
                # result LWEs are fresh encryptions of its plaintext bits -- what the engine's output decrypts to -- so that the
                # eight ranks do not each spend a minute of oracle time on a block (the world-2 case runs the homomorphic AES)
                start, _ = txdist.shard(nb, world, rank)
                bits = np.unpackbits(np.frombuffer(pt[16 * start:16 * (start + n_local)], np.uint8))
                r = O.lwe_encrypt_bits(p, keys.glwe_sk, bits, p.glwe_std, seed=100 + rank)
            else:
                r = ctx.transcipher_ctr(luts, iv_slab, ct_slab) if mode == "ctr" else ctx.transcipher(luts, ct_slab)
            out[: n_local * 128] = torch.from_numpy(r.view(np.int64))
        kept["mine"] = out[: n_local * 128]               # this rank's slab, as the sharded max consumes it
        return out

    res = txdist.transcipher_sharded(local_fn, ct, ivv, rank, world, dist)
    # the max mini-workload over the sharded result: local tournament, gather of the partial maxima, final tournament
    _, n_local = txdist.shard(nb, world, rank)
    mine = kept["mine"]

    def local_max(t):
        return torch.from_numpy(ctx.max_u16(np.ascontiguousarray(t.numpy().view(np.uint64)), tournament=True).view(np.int64))

    mx = txdist.max_sharded(local_max, mine, n_local * 8, rank, world, dist)
    if rank == 0:
        bits = O.lwe_decrypt_bits(keys.glwe_sk, res.numpy().view(np.uint64))
        ok = O.bits_msb_to_bytes(bits) == pt
        mbits = O.lwe_decrypt_bits(keys.glwe_sk, mx.numpy().view(np.uint64))
        ok = ok and int("".join(str(int(b)) for b in mbits), 2) == max(fx["db"][: 8 * nb])
        q.put(ok)
    else:
        assert res is None and mx is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world,nb", [("ctr", 2, 3), ("ctr", 8, 5)])
def test_sharded_transcipher_and_max_gloo(mode, world, nb):
    """world 2: ragged slabs; world 8 (the node size of BASELINE config 4): one block per rank and three ranks
    without work -- the gather of the result LWEs and the gather of the partial maxima with empty ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q, nb)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert ok


def _ip_worker(rank, world, port, q, vals=(0x1234, 0xFFFF, 0x0003, 0x8001)):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import txdist
    p = O.aes_tight(n=256)
    keys = O.Keys(p, seed=33)
    ctx = O.Ctx(keys, threads=4 if world <= 2 else 1)
    vals = list(vals)                                        # contiguous ranges of len(vals) / world values per rank
    per = len(vals) // world
    bits = np.array([(v >> (15 - i)) & 1 for v in vals for i in range(16)], np.uint8)
    ct = torch.from_numpy(O.lwe_encrypt_bits(p, keys.glwe_sk, bits, p.glwe_std, seed=8).view(np.int64))
    ranges = [(r * per, per) for r in range(world)]
    local = ct[ranges[rank][0] * 16:(ranges[rank][0] + ranges[rank][1]) * 16].clone()

    def local_ip(t):
        return torch.from_numpy(ctx.inner_product_u16(np.ascontiguousarray(t.numpy().view(np.uint64))).view(np.int64))

    out = txdist.inner_product_sharded(local_ip, local, ranges, len(vals), rank, world, dist)
    if rank == 0:
        dec = O.lwe_decrypt_bits(keys.glwe_sk, out.numpy().view(np.uint64))
        half = len(vals) // 2
        exp = sum((x * y) % 65536 for x, y in zip(vals[:half], vals[half:])) % 65536
        q.put(int("".join(str(int(b)) for b in dec), 2) == exp)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,vals", [(2, (0x1234, 0xFFFF, 0x0003, 0x8001)),
                                        (8, (0x1234, 0xFFFF, 0x0003, 0x8001, 0x7FFF, 0x0100, 0xABCD, 0x0001))])
def test_sharded_inner_product_gloo(world, vals):
    """pairs dealt out over the ranks, one point-to-point exchange of the remote operands, partial sums
    gathered and added by the same circuit on rank 0 (txdist.inner_product_sharded).  World 8 with eight values:
    one value per rank, four pairs, so ranks 0..3 each hold one operand and fetch the other, and ranks 4..7
    only send -- every branch of the exchange at the node size of BASELINE config 5."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150) + world
    procs = [ctx.Process(target=_ip_worker, args=(r, world, port, q, vals)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert ok
