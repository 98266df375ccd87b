"""Block sharding of the transciphering path across the GPUs of one node.

AES blocks are independent (ECB and CTR alike), so rank r processes a contiguous slab of blocks on
its own GPU with a full replica of the evaluation keys; there is NO data-path collective.  The only
exchange is the final gather of the result LWE ciphertexts (128 x 16392 B per block) to rank 0,
one `torch.distributed.gather` (RCCL over xGMI on GPUs, gloo in the CPU tests).

`dist` is the initialised `torch.distributed` module, or None for a single process without a process group.  A group
of ONE rank (bench.py / run_sharded.py --force-dist) goes through every collective below like any other world size.
"""
import math
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """number of HIP devices a process started from here would see, counted in a short-lived CHILD: on ROCm
    torch.cuda.device_count() may fall through to hipGetDeviceCount and initialise the HIP runtime in the caller, which a
    launcher that is about to start GPU ranks must not have done.  -1 if the count could not be taken."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else -1
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return -1


def launch_ranks(script, argv, n, backend="nccl", need_gpu=True, prepare=None, name=None):
    """Start the n ranks of `script argv` as fresh child processes of a parent that was started as a plain command (no RANK in
    its environment) and return the exit status for the parent.  Nothing here touches the GPU -- the devices are counted by a
    child process (visible_gpus) -- and no process is replaced: the ranks are ordinary subprocesses (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in their environment), rank 0's output reaches stdout through the
    inherited descriptor, and the status is the first non-zero status of a rank (the others are then stopped, by PID).
    Fails fast, with one line, on a node that has fewer GPUs than ranks (nccl: one rank per GPU; gloo rehearsals may share).
    `prepare` runs once in the parent before the ranks start (building the in-tree libraries)."""
    name = name or os.path.basename(script)
    if need_gpu:
        ndev = visible_gpus()
        if ndev <= 0:
            print(f"{name}: no HIP device on this node; there is no CPU fallback for the product path", file=sys.stderr)
            return 2
        if backend == "nccl" and ndev < n:
            print(f"{name}: {n} ranks need {n} GPUs on this node, it has {ndev} "
                  f"(a rehearsal with several ranks per GPU needs --backend gloo)", file=sys.stderr)
            return 2
    if prepare is not None:
        prepare()
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
    rc = 0

    def forward(signum, frame):                  # the launcher being stopped stops its ranks (exactly those PIDs)
        for p in procs:
            if p.poll() is None:
                p.send_signal(signum)
    import signal
    old = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.1)
            for p in list(live):
                r = p.poll()
                if r is not None:
                    live.remove(p)
                    if r != 0:
                        rc = r if r > 0 else 128 - r          # a rank killed by signal n: 128 + n, as a shell reports it
                        print(f"{name}: rank {procs.index(p)} exited with status {r}; stopping the other ranks", file=sys.stderr)
                        break
    finally:
        for p in procs:                          # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


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
    if dist is None:                       # no process group: one rank (a group of one still takes the collectives)
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
    if dist is None:                       # no process group: one rank (a group of one still takes the collectives)
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


def pair_shard(n_values, world, rank):
    """pairs (p, p + n/2), p in [start, start + count), handled by rank `rank`"""
    return shard(n_values // 2, world, rank)


def _overlap(a0, a1, b0, b1):
    lo, hi = max(a0, b0), min(a1, b1)
    return (lo, hi) if hi > lo else None


def inner_product_sharded(local_ip_fn, local_lwe_bits, ranges, n_values, rank, world, dist=None):
    """Inner-product mini-workload (harness/cleartext_impl.py:65-70) across ranks.

    Rank k holds the LWE bits of the contiguous values ranges[k] = (start, count) (the layout the sharded
    transcipher leaves behind: 16 rows per value).  The n/2 pairs (p, p + n/2) are dealt out evenly; ONE
    point-to-point exchange step brings every rank the operands of its pairs that live elsewhere, every rank
    computes its partial sum with local_ip_fn(tensor[2m*16, words]) -> tensor[16, words], the partial sums
    (16 LWEs each) are gathered to rank 0, and rank 0 adds them with one more call of the same circuit on
    (s_0, ..., s_{G-1}, 1, ..., 1), the ones being trivial ciphertexts.  Returns the 16 LWEs on rank 0."""
    import torch
    half = n_values // 2
    words = local_lwe_bits.shape[1]
    my0, myc = ranges[rank]
    if dist is None:                       # no process group: one rank (a group of one still takes the collectives)
        return local_ip_fn(local_lwe_bits[: n_values * 16])

    def wanted(r):
        """value intervals rank r needs, in the order it concatenates them"""
        ps, pc = pair_shard(n_values, world, r)
        return [(ps, ps + pc), (ps + half, ps + half + pc)] if pc else []

    # ---- one exchange step: rows this rank owns and another rank needs, and vice versa
    ops, recv_slots, sends = [], [], []
    ps, pc = pair_shard(n_values, world, rank)
    mine = torch.zeros((2 * pc * 16, words), dtype=local_lwe_bits.dtype, device=local_lwe_bits.device)
    for part, (w0, w1) in enumerate(wanted(rank)):
        for k in range(world):
            ov = _overlap(w0, w1, ranges[k][0], ranges[k][0] + ranges[k][1])
            if ov is None:
                continue
            dst = mine[(part * pc + ov[0] - w0) * 16:(part * pc + ov[1] - w0) * 16]
            if k == rank:
                dst.copy_(local_lwe_bits[(ov[0] - my0) * 16:(ov[1] - my0) * 16])
            else:
                buf = torch.empty_like(dst)
                recv_slots.append((dst, buf))
                ops.append(dist.P2POp(dist.irecv, buf, k))
    for r in range(world):
        if r == rank:
            continue
        for (w0, w1) in wanted(r):
            ov = _overlap(w0, w1, my0, my0 + myc)
            if ov is not None:
                t = local_lwe_bits[(ov[0] - my0) * 16:(ov[1] - my0) * 16].contiguous()
                sends.append(t)
                ops.append(dist.P2POp(dist.isend, t, r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for dst, buf in recv_slots:
        dst.copy_(buf)

    # ---- partial sums, gathered to rank 0
    part = local_ip_fn(mine) if pc else None
    buf = part if part is not None else torch.zeros((16, words), dtype=local_lwe_bits.dtype, device=local_lwe_bits.device)
    have = torch.tensor([1 if part is not None else 0], dtype=torch.int64, device=local_lwe_bits.device)
    if rank != 0:
        dist.gather(buf, gather_list=None, dst=0)
        dist.gather(have, gather_list=None, dst=0)
        return None
    parts = [torch.empty_like(buf) for _ in range(world)]
    flags = [torch.empty_like(have) for _ in range(world)]
    dist.gather(buf, gather_list=parts, dst=0)
    dist.gather(have, gather_list=flags, dst=0)
    sums = [p for p, f in zip(parts, flags) if int(f.item()) == 1]
    if len(sums) == 1:
        return sums[0]
    ones = torch.zeros((len(sums) * 16, words), dtype=buf.dtype, device=buf.device)
    ones[15::16, words - 1] = -(1 << 63)                    # value 1, MSB first: only bit 15 set, at 2^63, mask 0
    return local_ip_fn(torch.cat(sums + [ones], dim=0))
