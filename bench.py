#!/usr/bin/env python3
"""bench.py -- AES blocks transciphered per second on the medium instance (BASELINE.json).

One step = one pass of the hot path (tx_transcipher: first round by sample extraction from the
encrypted tables, then 9 x [LWE key switch, blind rotation, trace, scheme switch, LUT CMux tree,
linear layer]) over the
medium instance's 64 AES blocks, sharded block-wise over the ranks (one process per GPU), keys and
ciphertext bytes resident, result LWEs left in HBM and (N > 1) gathered to rank 0 with one RCCL
gather.  Prints ONE JSON line on rank 0.

`python bench.py --gpus N` is a complete N-rank run: started without RANK in the environment it launches its own N
rank processes (before this process has made any GPU call), forwards rank 0's line and exits with the ranks' status;
started under `python -m torch.distributed.run` (RANK set) it is one of the ranks.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector peak (256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)

# Counter-derived figures of the dominant kernel (HBM-side traffic per launch from separate rocprofv3 --pmc passes, FETCH_SIZE x 2
# + WRITE_SIZE per MI355X_MICROARCH.md's gfx950 correction; vector-pipe occupancy from an SQ pass) are NOT constants of this
# file: profiles/roofline_inputs.json names, for each figure, the CSV under profiles/ it was read from and the sha256 of the
# generated instruction stream (csrc/tx_br_asm_fair.inc) it was measured on.  A figure whose hash is not that of the stream
# in this tree is reported as null, never as a stale number.
ROOFLINE_INPUTS = os.path.join(ROOT, "profiles", "roofline_inputs.json")
BR_STREAM = os.path.join(ROOT, "temp-fhe-transciphering_amd", "csrc", "tx_br_asm_fair.inc")
# per wave and CMux step: v_add_f64 + v_mul_f64 + 2 x v_fma_f64 (`tools/asmgen/brgen.py --count` for the generated loop:
# 984 + 312 + 2 x 744 = 2784 flops in 2040 FP64 instructions; the same arithmetic in the hipcc kernels)
BR_F64_FLOP_PER_UNIT_STEP = 64 * (984 + 312 + 2 * 744)


def vector_instructions_per_unit_step(stream=BR_STREAM):
    """vector (VALU) instructions of one pass through the generated CMux loop of k_blind_rotate5, counted in the instruction stream of
    this tree: the lines between the loop's label and its backward branch that start with v_.  The loop's time follows this count
    (every vector instruction of its mix costs 4 cycles, profiles/r4ai_*), so count x 4 cycles x unit-steps / (SIMDs x time x clock) is
    the fraction of the vector pipe's issue slots the kernel uses."""
    try: lines = open(stream).read().split("\n")
    except OSError: return None
    labels = [i for i, l in enumerate(lines) if "L_step" in l and ":" in l]
    back = [i for i, l in enumerate(lines) if "s_cbranch" in l and "L_step" in l]
    if not labels or not back: return None
    body = lines[labels[0]:back[-1]]
    return sum(1 for l in body if l.lstrip().lstrip('"').lstrip().startswith("v_"))


def roofline_inputs(path=ROOFLINE_INPUTS, stream=BR_STREAM):
    """{key: entry} of profiles/roofline_inputs.json whose recorded sha256 is that of the instruction stream in this tree"""
    try:
        doc = json.load(open(path))
        have = hashlib.sha256(open(stream, "rb").read()).hexdigest()
    except (OSError, ValueError):
        return {}
    return {k: e for k, e in doc.get("entries", {}).items() if e.get("stream_sha256") == have}


CSRC = os.path.join(ROOT, "temp-fhe-transciphering_amd", "csrc")


def kernel_entries(path=ROOFLINE_INPUTS):
    """{kernel: entry} of profiles/roofline_inputs.json's `kernels` section: counter figures of the kernels of a step as they ran
    inside the pipeline (SQ pass: vector-pipe occupancy; FETCH_SIZE / WRITE_SIZE passes: traffic per launch), each tied to the
    sha256 over the source files the kernel is built from -- an entry whose files have changed since is dropped"""
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return {}
    out = {}
    for k, e in doc.get("kernels", {}).items():
        h = hashlib.sha256()
        try:
            for f in e.get("stream_files", []):
                h.update(open(os.path.join(CSRC, f), "rb").read())
        except OSError:
            continue
        if h.hexdigest() == e.get("stream_sha256"):
            out[k] = e
    return out


def stage_times(n_gpus, seed):
    """the harness's view (harness/utils.py:85-109 times whole stage PROCESSES): the medium instance once through this repo's ten
    stage executables (python/run_submission.py = the harness's stage order and verification), server stages sharded in-process
    over the first n_gpus devices (TX_DEVICES); seconds of the two server stages, and whether both verifications passed"""
    import tempfile
    import shutil
    import subprocess
    root = tempfile.mkdtemp(prefix="tx_bench_stage_")
    # run_submission.py as a process of its own: it clocks a stage from spawn to exit as the harness does, and a spawn from THIS
    # process (torch loaded, gigabytes of device memory mapped) costs tens of milliseconds that are not the stage's
    env = dict(os.environ, TX_DEVICES=",".join(str(i) for i in range(n_gpus)))
    try:
        runs = []
        for _ in range(2):          # twice: the first run of a fresh box pays for cold libraries and binaries (reported as first_run)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python", "run_submission.py"), "2",
                                "--root", root, "--seed", str(seed), "--fhe_seed", str(seed + 7)], capture_output=True, text=True, env=env, timeout=600)
            res = json.load(open(os.path.join(root, "measurements", "medium", "results.json")))
            st = res["per_stage"]
            runs.append({"aes_decryption": st["7: server_encrypted_aes_decryption"], "compute": st["8: server_encrypted_compute"],
                         "verified": r.returncode == 0 and "AES decryption: PASS; max: PASS" in r.stdout, "total_latency_s": res["total_latency_s"]})
        return {"aes_decryption": runs[1]["aes_decryption"], "compute": runs[1]["compute"],
                "verified": runs[0]["verified"] and runs[1]["verified"], "devices": n_gpus, "total_latency_s": runs[1]["total_latency_s"],
                "first_run": {"aes_decryption": runs[0]["aes_decryption"], "compute": runs[0]["compute"]},
                "note": "wall seconds of the stage PROCESSES of one medium instance (process start, HIP initialisation, key files -> "
                        "device, the computation, result file), max mini-workload; the harness's clock, not `value`'s; the instance is "
                        "run twice, the second run is reported (first_run: the first, with cold libraries on a fresh box)"}
    except (OSError, ValueError, KeyError, subprocess.TimeoutExpired) as ex:
        return {"aes_decryption": None, "compute": None, "verified": False, "devices": n_gpus, "error": repr(ex)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def algorithmic_bytes_per_block(M):
    """stage-materialised model, keys once per stage per batch of M blocks"""
    return 1.847e9 + 0.556e9 / M


# SURVEY.md 8(d), per block and FULL bootstrapped round: KS 2.89 + blind rotation 3.93 + extract/trace 25.17 + scheme switch 88.08 +
# LUT tree 74.45 MB scale with the units (bit ciphertexts) a round actually bootstraps; the linear layer's 10.49 MB does not
BOOTSTRAP_BYTES_PER_BLOCK_ROUND = (2.89 + 3.93 + 25.17 + 88.08 + 74.45) * 1e6


def path_algorithmic_bytes(n_blocks, units_bootstrapped, launches_of_the_keys=1):
    """algorithmic bytes of ONE pass over n_blocks blocks in which the blind rotations processed units_bootstrapped units in
    total (tx_timing.units_blind_rotate): the full model 9 x 128 units per block, minus the per-round terms of the units the
    schedule did not materialise (CTR: the shared second round bootstraps each distinct column once), keys once per batch"""
    full_units = 9 * 128 * n_blocks
    skipped_block_rounds = max(0, full_units - units_bootstrapped) / 128.0
    return n_blocks * 1.847e9 + launches_of_the_keys * 0.556e9 - skipped_block_rounds * BOOTSTRAP_BYTES_PER_BLOCK_ROUND


def br_launch_bytes(units, n, launches=1):
    """blind-rotation launches: per unit (bit) in (n+1) words + out 3*1024 words; BSK (standard size) once per launch"""
    return units * ((n + 1) * 8 + 3 * 1024 * 8) + launches * n * 9 * 1024 * 8


def cpu_baseline(n, seed, threads):
    """C restatement of the reference's CPU path (oracle/, kind 'port'), timed twice as SURVEY.md 8(d) asks: (ii) on all host
    cores this job may use (the 128 bits of a block in parallel) -- ONE WHOLE AES block, rounds 10/9 and all nine
    bootstrapped rounds, no scale factor: `value` = 1 / its seconds; (i) on ONE thread -- the reference is `default =
    ["monothread"]` (submission/cbs_lib/Cargo.toml:31-34) -- a bounded sample, two of the nine bootstrapped rounds
    (about 20 s), scaled and labelled as such under `one_thread`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    p = O.aes_tight(n=n)
    keys = O.Keys(p, seed=seed)
    aes_key = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0")
    luts = keys.round_luts(aes_key)
    ct = bytes(range(16))

    def leg(nthreads, rounds):
        ctx = O.Ctx(keys, threads=nthreads)
        ctx.transcipher_partial(luts, ct, 0)          # warm threads / tables
        t0 = time.time()
        ctx.transcipher_partial(luts, ct, rounds)
        dt = time.time() - t0
        ctx.close()
        th = f"{nthreads} thread{'s' if nthreads > 1 else ''}"
        if rounds == 9:
            return {"value": 1.0 / dt, "unit": "AES blocks/s", "cores": nthreads, "kind": "port",
                    "sample": f"1 whole AES block (rounds 10/9 + all 9 bootstrapped rounds) in {dt:.2f}s on {th}; not scaled"}
        return {"value": 1.0 / (dt * 9 / rounds), "unit": "AES blocks/s", "cores": nthreads, "kind": "port",
                "sample": f"1 block x {rounds} of 9 bootstrapped rounds in {dt:.1f}s on {th}, scaled x{9 / rounds:g}"}

    out = leg(threads, 9)
    out["one_thread"] = leg(1, 2)                     # about 10 s per round on one core
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=64, help="AES blocks in the instance (medium = 64)")
    ap.add_argument("--n", type=int, default=768, help="small LWE dimension (AES_TIGHT: 768)")
    ap.add_argument("--mode", choices=["ecb", "ctr"], default="ctr",
                    help="ctr = forward cipher on counter blocks, the mode the harness uses for the medium instance "
                         "(aes_keygen_and_encrypt.py:49-55); ecb = the inverse-cipher schedule of the reference's own "
                         "server_encrypted_aes_decryption.rs (same kernels, 4 instead of 3 LUT tables per round)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --blocks in total (the medium instance); weak: --blocks per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-equal-work-legs", "--no-unshared-leg", dest="no_equal_work_legs", action="store_true",
                    help="one GPU, CTR: skip the two legs that follow the timed region by default -- the same instance with every "
                         "column of the second round evaluated (value_without_sharing; outputs compared with the shared "
                         "schedule's on the device) and under the reference's own ECB inverse-cipher schedule "
                         "(value_ecb_schedule).  Profiler scripts pass this: the legs' launches would enter the per-kernel "
                         "averages of the command")
    ap.add_argument("--no-ctr-share", action="store_true",
                    help="CTR mode: bootstrap every state column of the second round, also those that hold the same ciphertexts "
                         "in every block of the batch (tx_set_ctr_column_sharing(c, 0); same results, for A/B timing)")
    ap.add_argument("--with-mini-workloads", action="store_true",
                    help="after the timed region run the two post-transcipher mini-workloads (max, inner product mod 2^16) on the "
                         "result LWEs where they lie in HBM (tx_max_u16_io / tx_inner_product_u16_io, device in, device out), "
                         "verify both against the cleartext and report their seconds beside `value`")
    ap.add_argument("--no-stage-times", action="store_true",
                    help="skip the run of the harness's stage executables in front of the timed region (stage_s: the wall time of "
                         "server_encrypted_aes_decryption / server_encrypted_compute as the harness clocks them, harness/utils.py:85-109)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 still goes through init_process_group and the collectives of the sharded path (one RCCL gather "
                         "of the result LWEs; with --with-mini-workloads the gathers of max / inner product)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / shard / gather logic only: no GPU, no engine, row patterns instead of ciphertexts "
                         "(CPU rehearsal of the N-rank launch with --backend gloo)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`bench.py --gpus N` started as a plain command (no RANK): the N ranks as fresh child processes (txdist.launch_ranks:
    before any GPU call in this process, no process replaced, fail-fast on a node with fewer GPUs)"""
    import txdist

    def prepare():
        import __graft_entry__
        __graft_entry__.build()                  # once, in the parent: the ranks then find the libraries up to date
    return txdist.launch_ranks(__file__, argv, args.gpus, backend=args.backend, need_gpu=not args.dry_run,
                               prepare=None if args.dry_run else prepare, name="bench.py")


def dry_run(args, rank, world):
    """The launch path without a GPU: rendezvous, block sharding, the gather of the result rows and the max-over-ranks
    reduction of the timed region, with row patterns (global block number, bit index) in place of ciphertexts."""
    import torch
    import txdist
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    n_blocks = args.blocks * (world if args.scaling == "weak" else 1)
    ct = bytes(16 * n_blocks)
    seen = {}

    def local_fn(ct_slab, iv_slab, n_local, n_pad):
        start, _ = txdist.shard(n_blocks, world, rank)
        out = torch.full((n_pad * 128, 4), -1, dtype=torch.int64)
        for b in range(n_local):
            out[b * 128:(b + 1) * 128, 0] = start + b
            out[b * 128:(b + 1) * 128, 1] = torch.arange(128)
        out[:, 2] = rank
        seen["n"] = n_local
        return out

    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        res = txdist.transcipher_sharded(local_fn, ct, None, rank, world, dist)
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = True
    if rank == 0:
        want_blocks = torch.arange(n_blocks).repeat_interleave(128)
        ok = (res.shape[0] == n_blocks * 128 and bool(torch.equal(res[:, 0], want_blocks))
              and bool(torch.equal(res[:, 1], torch.arange(128).repeat(n_blocks))))
        ranks = sorted(set(int(v) for v in res[:, 2].tolist()))
        print(json.dumps({"metric": "AES blocks transciphered/sec (medium instance)", "value": None, "unit": "AES blocks/s",
                          "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "backend": args.backend,
                          "blocks": n_blocks, "blocks_per_rank": [txdist.shard(n_blocks, world, r)[1] for r in range(world)],
                          "ranks_in_result": ranks, "gathered_in_block_order": ok, "scaling": args.scaling}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, argv))
    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != world:
        raise SystemExit(f"bench.py: --gpus {world} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if rank >= world:
        raise SystemExit(f"bench.py: RANK={rank} outside --gpus {world}")
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import txdist as _t
            os.environ["MASTER_PORT"] = str(_t.free_port()) if world == 1 else "29511"
    if args.dry_run:
        sys.exit(dry_run(args, rank, world))

    import numpy as np
    import torch
    import __graft_entry__
    __graft_entry__.build()                      # no-op when the in-tree libraries are up to date (file-locked)
    import txhip
    import txclient
    import txdist

    dist = None
    if use_dist:
        import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    # ---- the harness's clock, taken FIRST on rank 0 (this process holds no device memory yet; with N > 1 the other ranks wait
    # in the rendezvous with idle GPUs): one medium instance through the ten stage executables, the two server stages sharded
    # in-process over the N devices of the run (TX_DEVICES)
    stage_s = None
    if rank == 0 and not args.no_stage_times and args.n == 768:
        stage_s = stage_times(world, args.seed)
    if args.backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} GPUs on this node, it has {torch.cuda.device_count()}")
    # one rank per GPU; a rehearsal with more ranks than GPUs (gloo only) wraps around
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    on_device = args.backend == "nccl"           # RCCL moves device buffers; the gloo rehearsal stages them through the host
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # ---- synthetic medium instance: uniform u16 values, big-endian, AES-128 under the harness key
    n_blocks = args.blocks * (world if args.scaling == "weak" else 1)
    rng = np.random.RandomState(args.seed)
    db = rng.randint(0, 65536, size=n_blocks * 8).astype(np.uint16)
    pt = b"".join(int(v).to_bytes(2, "big") for v in db)
    aes_key = bytes.fromhex("dc937b59892604f5a86ac96936cd7ff0")
    aes_iv = bytes.fromhex("64b92e64992534e495b9da182409427c")
    if args.mode == "ecb":
        ct, iv = txclient.aes_ecb_encrypt(aes_key, pt), None
    else:
        ct, iv = txclient.aes_ctr_crypt(aes_key, aes_iv, pt), aes_iv

    # ---- keys (same seed on every rank -> identical replicas, no broadcast needed)
    p = txhip.aes_tight(n=args.n)
    t0 = time.time()
    keys = txclient.ClientKeys(p, seed=args.seed)
    luts = keys.round_luts(aes_key, seed=args.seed + 1) if args.mode == "ecb" else keys.ctr_luts(aes_key, seed=args.seed + 1)
    t_keygen = time.time() - t0
    eng = txhip.Engine(p, device=dev_index)
    if args.no_ctr_share:
        eng.set_ctr_column_sharing(False)
    t0 = time.time()
    eng.load_eval_keys(keys.bsk, keys.ksk, keys.auto, keys.ss)
    (eng.load_round_luts if args.mode == "ecb" else eng.load_ctr_luts)(luts)
    t_load = time.time() - t0

    words = 2 * 1024 + 1
    br_ms, br_launches, br_units, all_units = [0.0], [0], [0], [0]

    out_bufs = {}

    def local_fn(ct_slab, iv_slab, n_local, n_pad):
        # the result buffer (134 MB for 64 blocks) is allocated once and reused by every step: the library writes the
        # LWEs of the step into it (`out_is_device`), nothing is carried over
        out = out_bufs.get(n_pad)
        if out is None:
            out = out_bufs[n_pad] = torch.empty((n_pad * 128, words), dtype=torch.int64, device=dev)
        if n_local:
            eng.transcipher(ct_slab, iv_slab, out_device_ptr=out.data_ptr())
            tm = eng.timing()
            # the blind-rotation kernel that took most units (a call may split a ragged tail off to a small-launch kernel)
            br_ms[0] += tm["ms_br_main"]; br_launches[0] += tm["n_br_main"]; br_units[0] += tm["units_br_main"]
            all_units[0] += tm["units_blind_rotate"]
        return out if (on_device or dist is None) else out.cpu()

    def step():
        return txdist.transcipher_sharded(local_fn, ct, iv, rank, world, dist)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- correctness gate: decrypted result must equal the plaintext, bit for bit
    res = step()
    if rank == 0:
        got = keys.decrypt_bits(res.cpu().numpy().view(np.uint64))
        u16 = txclient.bits_to_u16(got)
        if not np.array_equal(u16, db):
            raise SystemExit("bench.py: transciphered result does not decrypt to the plaintext; no number reported")
    del res
    for _ in range(max(0, args.warmup - 1)):
        step()
    br_ms[0], br_launches[0], br_units[0], all_units[0] = 0.0, 0, 0, 0
    last_timing = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    last_timing = eng.timing()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if on_device else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    t_ms, t_launches, t_units, t_all_units = br_ms[0], br_launches[0], br_units[0], all_units[0]   # totals of the timed region
    _, m_local = txdist.shard(n_blocks, world, rank)

    # ---- equal-work legs (one GPU, CTR): the same instance with every column of the second round evaluated, and under the
    # reference's own ECB inverse-cipher schedule (server_encrypted_aes_decryption.rs:130-180) -- one warm-up + two timed steps each
    value_unshared = value_ecb = None
    if world == 1 and args.mode == "ctr" and not args.no_ctr_share and not args.no_equal_work_legs:
        eng2 = txhip.Engine(p, device=dev_index)
        eng2.set_ctr_column_sharing(False)
        eng2.load_eval_keys(keys.bsk, keys.ksk, keys.auto, keys.ss)
        eng2.load_ctr_luts(luts)
        out2 = torch.empty((n_blocks * 128, words), dtype=torch.int64, device=dev)

        def leg(ct_, iv_):
            eng2.transcipher(ct_, iv_, out_device_ptr=out2.data_ptr())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                eng2.transcipher(ct_, iv_, out_device_ptr=out2.data_ptr())
            torch.cuda.synchronize()
            return n_blocks * 2 / (time.perf_counter() - t1)

        value_unshared = leg(ct, iv)
        shared = out_bufs[m_local]
        eng.transcipher(ct, iv, out_device_ptr=shared.data_ptr())
        if not bool(torch.equal(out2, shared)):               # the two schedules give the same ciphertexts, bit for bit
            raise SystemExit("bench.py: shared and unshared CTR schedules differ; no number reported")
        eng2.load_round_luts(keys.round_luts(aes_key, seed=args.seed + 2))
        value_ecb = leg(txclient.aes_ecb_encrypt(aes_key, pt), None)
        if not np.array_equal(txclient.bits_to_u16(keys.decrypt_bits(out2.cpu().numpy().view(np.uint64))), db):
            raise SystemExit("bench.py: the ECB schedule's result does not decrypt to the plaintext; no number reported")
        eng2.close()
        del out2

    # ---- host-output leg (one GPU): the same instance with the result LWEs delivered to HOST memory -- what a caller without
    # device pointers gets (the stage executables): tx_transcipher_chunks into page-locked memory, the 134 MB travelling on the
    # copy stream.  PCIe-inclusive, reported beside `value`, never as `value`.
    value_host = None
    if world == 1 and not args.no_equal_work_legs:
        import ctypes as C
        L = txhip.lib()
        nbytes = n_blocks * 128 * words * 8
        hp = L.tx_host_alloc(nbytes)
        if hp:
            ctb = np.frombuffer(bytes(ct), np.uint8).copy()
            ivb = np.frombuffer(bytes(iv), np.uint8).copy() if iv is not None else None
            cb = txhip.CHUNK_FN(lambda u, a, b: None)

            def host_step():
                rc = L.tx_transcipher_chunks(eng.h, ctb.ctypes.data_as(txhip.u8p), n_blocks,
                                             ivb.ctypes.data_as(txhip.u8p) if ivb is not None else None, C.c_void_p(hp), 0, cb, None)
                if rc:
                    raise SystemExit("bench.py: host-output leg failed: " + L.tx_last_error().decode())
            host_step()
            t1 = time.perf_counter()
            for _ in range(2):
                host_step()
            value_host = n_blocks * 2 / (time.perf_counter() - t1)
            got = np.ctypeslib.as_array(C.cast(hp, txhip.u64p), shape=(n_blocks * 128 * words,))
            if not np.array_equal(txclient.bits_to_u16(keys.decrypt_bits(got)), db):
                raise SystemExit("bench.py: the host-output leg's result does not decrypt to the plaintext; no number reported")
            del got
            L.tx_host_free(C.c_void_p(hp))

    # ---- the two post-transcipher mini-workloads on the device-resident result (config 5 of BASELINE.json at --blocks 512)
    mini = None
    if args.with_mini_workloads:
        mine = out_bufs[-(-n_blocks // world)]
        n_local_values, n_values = m_local * 8, n_blocks * 8
        cdev = dev if (on_device or dist is None) else torch.device("cpu")

        def on_card(fn):
            def run(t):
                t = t.to(dev).contiguous()
                o = torch.empty((16, words), dtype=torch.int64, device=dev)
                fn(t.data_ptr(), t.shape[0] // 16, o.data_ptr())
                return o.to(cdev)
            return run

        mine_c = mine if cdev == dev else mine.cpu()
        barrier(); t1 = time.perf_counter()
        mx = txdist.max_sharded(on_card(eng.max_u16_device), mine_c, n_local_values, rank, world, dist)
        barrier(); t_max = time.perf_counter() - t1
        ranges = [(8 * s, 8 * c) for s, c in (txdist.shard(n_blocks, world, r) for r in range(world))]
        t1 = time.perf_counter()
        ip = txdist.inner_product_sharded(on_card(eng.inner_product_u16_device), mine_c[: n_local_values * 16], ranges, n_values,
                                          rank, world, dist)
        barrier(); t_ip = time.perf_counter() - t1
        if rank == 0:
            vals = [int(v) for v in db]
            half = len(vals) // 2
            got_max = int(txclient.bits_to_u16(keys.decrypt_bits(mx.cpu().numpy().view(np.uint64)))[0])
            got_ip = int(txclient.bits_to_u16(keys.decrypt_bits(ip.cpu().numpy().view(np.uint64)))[0])
            exp_ip = sum((vals[i] * vals[i + half]) % 65536 for i in range(half)) % 65536
            if got_max != max(vals) or got_ip != exp_ip:
                raise SystemExit("bench.py: a mini-workload result does not decrypt to the cleartext value; no number reported")
            mini = {"values": n_values, "max_s": t_max, "inner_product_s": t_ip, "verified": True,
                    "input": "result LWEs of the transcipher, device-resident (tx_max_u16_io / tx_inner_product_u16_io, in_is_device = 1)"}

    if rank == 0:
        value = n_blocks * args.steps / dt
        br_avg_ms = t_ms / max(1, t_launches)
        br_avg_units = t_units / max(1, t_launches)
        br_kernel = eng.last_blind_rotate_kernel()         # what run_br (csrc/tx_api.hip) launched, as the library reports it
        # all launches of that kernel in the timed region (they differ in size: in CTR mode the second round bootstraps
        # only the distinct state columns): algorithmic bytes of all of them over the time of all of them
        achieved = br_launch_bytes(t_units, args.n, t_launches) / (t_ms * 1e-3) / 1e9
        br_flops = t_units * args.n * BR_F64_FLOP_PER_UNIT_STEP
        vips = vector_instructions_per_unit_step()
        # counter-derived figures: only from profiles/roofline_inputs.json, only when measured on this tree's instruction stream
        # and on this launch mix (the entry states the average units per launch its CSV covers)
        rin = roofline_inputs() if (args.n == 768 and br_kernel == "k_blind_rotate5") else {}
        def counter(key):
            e = rin.get(key)
            return e if (e and abs(e.get("avg_units_per_launch", -1) - br_avg_units) < 0.5) else None
        tr, vb = counter("traffic_default_mix") or counter("traffic_8192_units"), counter("vector_pipe_busy")
        units_step = t_all_units / args.steps              # units this rank's blind rotations processed per step
        path_bytes = path_algorithmic_bytes(m_local, units_step)
        path_gbps = path_bytes / (dt / args.steps) / 1e9
        line = {
            "metric": "AES blocks transciphered/sec (medium instance)", "value": value, "unit": "AES blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"medium instance: {n_blocks} AES-128 blocks (512 x u16 per 64 blocks), "
                                   f"AES_TIGHT n={args.n}, {'ECB inverse-cipher schedule of the reference' if args.mode == 'ecb' else 'CTR forward-cipher schedule'}, "
                                   f"full 10 rounds, keys resident, result LWEs left in HBM",
                       "blocks": n_blocks, "blocks_per_gpu": m_local, "parallelism": f"block-sharded x{world}, one gather",
                       "backend": (args.backend if dist is not None else None)},
            "value_without_sharing": value_unshared, "value_ecb_schedule": value_ecb,
            "value_host_result": value_host,       # PCIe-inclusive: result LWEs delivered to page-locked host memory (never `value`)
            "roofline": {"bound": "fp64_valu", "bound_note": "what binds is the FP64 vector pipe (fp64_valu / vector_pipe_busy_pmc below); "
                                       "achieved / peak / frac are the HBM figure the contract asks for: algorithmic bytes of the dominant "
                                       "kernel per second against 8 TB/s",
                         "kernel": br_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (tr["bytes_per_launch"] if tr else None),
                         "traffic_source": (tr["source"] if tr else None),
                         "avg_launch_ms": br_avg_ms, "launches": t_launches, "avg_units_per_launch": br_avg_units,
                         "algorithmic_bytes_per_launch": br_launch_bytes(t_units, args.n, t_launches) / max(1, t_launches),
                         "units_bootstrapped_per_step": units_step, "units_of_the_full_schedule": 9 * 128 * m_local,
                         "path_algorithmic_bytes_per_step": path_bytes,
                         "path_algorithmic_GBps": path_gbps, "path_frac": path_gbps / HBM_PEAK_GBS,
                         "fp64_valu": {"achieved_TFLOPs": br_flops / (t_ms * 1e-3) / 1e12, "peak_TFLOPs": FP64_PEAK_TFLOPS,
                                       "frac": br_flops / (t_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                         "vector_issue": (None if not (vips and br_kernel == "k_blind_rotate5") else
                                          {"instructions_per_unit_and_step": vips, "cycles_per_instruction": 4, "simds": 1024, "nominal_clock_GHz": 2.4,
                                           "frac_at_nominal_clock": vips * 4.0 * t_units * args.n / (1024 * t_ms * 1e-3 * 2.4e9)}),
                         "vector_pipe_busy_pmc": (vb["value"] if vb else None),
                         "vector_pipe_busy_source": (vb["source"] if vb else None),
                         "note": "the path is f64 transform arithmetic, not a stream: HBM fraction is low by construction; "
                                 "path_* = SURVEY 8(d)'s stage-materialised bytes of the units this step actually bootstrapped "
                                 "(the CTR schedule's shared second round materialises fewer) over the step time; "
                                 "fp64_valu counts FP64 flops against the FMA peak at the nominal clock (half of this loop's FP64 "
                                 "instructions are adds, a fifth of its vector instructions 32-bit, and the chip runs it at "
                                 "2.1-2.2 GHz: power-limited, profiles/r4ar_*); vector_issue counts the loop's vector instructions in the "
                                 "stream of this tree at 4 cycles each against the SIMDs' issue slots at the nominal clock (the bound the "
                                 "loop's time follows); vector_pipe_busy_pmc is the fraction of cycles a SIMD's vector pipe "
                                 "executes an instruction (SQ_ACTIVE_INST_VALU x waves per SIMD / SQ_WAVE_CYCLES); traffic and "
                                 "vector_pipe_busy_pmc come from profiles/roofline_inputs.json and are null unless they were "
                                 "measured on this tree's instruction stream and this launch mix"},
            "ctr_shared_columns": (None if args.mode != "ctr" else
                                   {"enabled": not args.no_ctr_share, "value_without_sharing": value_unshared,
                                    "note": "CTR: the state entering the second round is a public function of the counters, so columns fed by "
                                            "the same counter bytes hold the same ciphertexts in every block; each distinct column is "
                                            "bootstrapped once (64 consecutive counters: 268 of 1024 bytes in that round, the other 8 "
                                            "rounds in full); outputs identical to evaluating every column (value_without_sharing: "
                                            "timed after the timed region on a second context with column sharing switched off, outputs compared on "
                                            "the device; value_ecb_schedule: the reference's own inverse-cipher schedule, every column "
                                            "in every round)"}),
            "mini_workloads": mini, "stage_s": None,
            "kernel_ms_last_step": {k: round(v, 3) for k, v in last_timing.items() if k.startswith("ms_")},
            "setup_s": {"client_keygen": round(t_keygen, 2), "key_upload_and_transform": round(t_load, 3)},
        }
        # the three kernels that make up 99 % of a step, with the counter figures measured on them inside the pipeline
        ke = kernel_entries() if args.n == 768 else {}
        fam = (("blind rotation", br_kernel, "ms_blind_rotate"), ("extract + trace + scheme switch", "k_trace_scheme_switch4", "ms_trace"),
               ("LUT tree", "k_lut_eval", "ms_lut_eval"))
        line["roofline"]["kernels"] = [
            {"stage": stage, "kernel": k, "ms_per_step": round(last_timing[key], 3),
             "share_of_kernel_time": round(last_timing[key] / max(1e-9, last_timing["ms_total"]), 4),
             "vector_pipe_busy": (ke[k].get("vector_pipe_busy") if k in ke else None),
             "traffic": (ke[k].get("bytes_per_launch") if k in ke else None),
             "launches_per_step": (ke[k].get("launches_per_step") if k in ke else None),
             "source": (ke[k].get("source") if k in ke else None)} for stage, k, key in fam]
        line["stage_s"] = stage_s
        if not args.no_cpu_baseline and world == 1:
            # the GPU box gives one GPU's job a share of 16 host cores; never more threads than that
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(args.n, args.seed, max(1, min(avail, 16)))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
