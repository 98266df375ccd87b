"""Block sharding of the transciphering path across the GPUs of one node.

AES blocks are independent (ECB and CTR alike), so rank r processes a contiguous slab of blocks on
its own GPU with a full replica of the evaluation keys; there is NO data-path collective.  The only
exchange is the final gather of the result LWE ciphertexts (128 x 16392 B per block) to rank 0,
one `torch.distributed.gather` (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
import math


def shard(n_blocks, world, rank):
    """contiguous slab [start, start + count) of rank `rank`; slabs of ceil(n/world) blocks"""
    per = math.ceil(n_blocks / world) if world > 0 else n_blocks
    start = min(rank * per, n_blocks)
    return start, max(0, min(per, n_blocks - start))


def iv_add(iv, k):
    """128-bit big-endian counter arithmetic (harness/aes_keygen_and_encrypt.py:52-54)"""
    return ((int.from_bytes(iv, "big") + k) % (1 << 128)).to_bytes(16, "big")


def transcipher_sharded(local_fn, ct, iv, rank, world, dist=None, empty_like=None):
    """local_fn(ct_slab: bytes, iv_for_slab: bytes|None, n_local: int, n_pad: int) -> tensor
    [n_pad*128, words] (int64; rows past n_local*128 are padding).  Returns on rank 0 the gathered
    tensor [n_blocks*128, words]; None elsewhere."""
    n_blocks = len(ct) // 16
    start, cnt = shard(n_blocks, world, rank)
    per = math.ceil(n_blocks / world)
    local = local_fn(ct[16 * start:16 * (start + cnt)], iv_add(iv, start) if iv is not None else None, cnt, per)
    if world == 1 or dist is None:
        return local[: n_blocks * 128]
    import torch
    if rank == 0:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=parts, dst=0)
        rows = []
        for r in range(world):
            _, c = shard(n_blocks, world, r)
            rows.append(parts[r][: c * 128])
        return torch.cat(rows, dim=0)
    dist.gather(local, gather_list=None, dst=0)
    return None
