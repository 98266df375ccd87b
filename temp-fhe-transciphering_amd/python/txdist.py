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


def max_sharded(local_max_fn, local_lwe_bits, n_local_values, rank, world, dist=None):
    """post-transcipher max across ranks: every rank reduces its own values to one
    16-bit maximum (local_max_fn(tensor[n*16, words]) -> tensor[16, words]), the G partial maxima are
    gathered to rank 0 (16 LWEs = 262 KB each, one exchange step), and rank 0 finishes with one more
    tournament over the G survivors.  Ranks without values contribute nothing."""
    import torch
    part = local_max_fn(local_lwe_bits[: n_local_values * 16]) if n_local_values > 0 else None
    if world == 1 or dist is None:
        return part
    words = local_lwe_bits.shape[1]
    buf = part if part is not None else torch.zeros((16, words), dtype=local_lwe_bits.dtype, device=local_lwe_bits.device)
    have = torch.tensor([1 if part is not None else 0], dtype=torch.int64, device=local_lwe_bits.device)
    if rank == 0:
        parts = [torch.empty_like(buf) for _ in range(world)]
        flags = [torch.empty_like(have) for _ in range(world)]
        dist.gather(buf, gather_list=parts, dst=0)
        dist.gather(have, gather_list=flags, dst=0)
        survivors = [p for p, f in zip(parts, flags) if int(f.item()) == 1]
        return local_max_fn(torch.cat(survivors, dim=0))
    dist.gather(buf, gather_list=None, dst=0)
    dist.gather(have, gather_list=None, dst=0)
    return None
