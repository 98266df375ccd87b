#!/usr/bin/env python3
"""bench.py -- AES blocks transciphered per second on the medium instance (BASELINE.json).

One step = one pass of the hot path (tx_transcipher: first round by sample extraction from the
encrypted tables, then 9 x [LWE key switch, blind rotation, trace, scheme switch, LUT CMux tree,
linear layer]) over the
medium instance's 64 AES blocks, sharded block-wise over the ranks (one process per GPU), keys and
ciphertext bytes resident, result LWEs left in HBM and (N > 1) gathered to rank 0 with one RCCL
gather.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector peak (256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)

# HBM-side traffic of ONE blind-rotation launch over 8192 units (medium on one GPU), from separate rocprofv3 --pmc
# passes: FETCH_SIZE x 2 (gfx950 counts half of a 16 B/lane streaming read, MI355X_MICROARCH.md HBM) + WRITE_SIZE.
# The key stream (54 MiB of Fourier keys) is re-read once per XCD per resident batch of waves, mostly from the
# Infinity Cache.  k_blind_rotate5 (two waves per SIMD, persistent; runs above 4 units per CU):
# profiles/r2_final_pmc_{fetch,write}_size_blind_rotate5.csv (each XCD streams the key 4 times, once per unit of a wave);
# k_blind_rotate4: profiles/r2_pmc_{fetch,write}_size_blind_rotate4.csv; k_blind_rotate: profiles/r1_pmc_{fetch,write}_size_br_trace.csv.
# The default command (medium, CTR, shared columns): its nine k_blind_rotate5 launches per step are eight of 8192 units and
# one of 2048 (the second round's distinct columns; the 96 units of its ragged tail go to k_blind_rotate6).  Average per
# launch over exactly those launches, separate --pmc passes over `bench.py --steps 1` (tools/profile_round.sh):
# profiles/r3g_pmc_{fetch,write}_size_bench_command.csv: FETCH_SIZE 852 378 KiB (x 2), WRITE_SIZE 191 488 KiB.
BR_TRAFFIC_AVG_DEFAULT_MIX = (852378 * 2 + 191488) * 1024
BR_TRAFFIC_BYTES_8192 = {"k_blind_rotate5": (930104 * 2 + 207872) * 1024, "k_blind_rotate4": (1779967 * 2 + 196608) * 1024,
                         "k_blind_rotate": (1795972 * 2 + 196608) * 1024}
# PMC pass over the same launch (tools/pmc_kernel.sh, profiles/r2c_pmc_sq_blind_rotate5.csv): SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
# = 0.401 per wave, two waves per SIMD
BR_VALU_BUSY_PMC = {"k_blind_rotate5": 0.80}
# per wave and CMux step: v_add_f64 + v_mul_f64 + 2 x v_fma_f64 (`tools/asmgen/brgen.py --count` for the generated loop:
# 984 + 312 + 2 x 744 = 2784 flops in 2040 FP64 instructions; the same arithmetic in the hipcc kernels)
BR_F64_FLOP_PER_UNIT_STEP = 64 * (984 + 312 + 2 * 744)


def algorithmic_bytes_per_block(M):
    """stage-materialised model, keys once per stage per batch of M blocks"""
    return 1.847e9 + 0.556e9 / M


def br_launch_bytes(units, n, launches=1):
    """blind-rotation launches: per unit (bit) in (n+1) words + out 3*1024 words; BSK (standard size) once per launch"""
    return units * ((n + 1) * 8 + 3 * 1024 * 8) + launches * n * 9 * 1024 * 8


def cpu_baseline(n, seed, threads):
    """C restatement of the reference's CPU path (oracle/, kind 'port') on a bounded sample of the same workload:
    ONE AES block through rounds 10/9 plus FOUR (one thread: TWO) bootstrapped rounds (a block has 9 such rounds,
    which are >99.9% of its cost), timed twice as SURVEY.md 8(d) asks: (i) on ONE thread -- the reference is `default = ["monothread"]`
    (submission/cbs_lib/Cargo.toml:31-34) -- and (ii) on all host cores this job may use (the 128 bits of a block in
    parallel).  About 20 core-seconds each.  `value` is leg (ii); leg (i) is reported under `one_thread`."""
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
        return {"value": 1.0 / (dt * 9 / rounds), "unit": "AES blocks/s", "cores": nthreads, "kind": "port",
                "sample": f"1 block x {rounds} of 9 bootstrapped rounds in {dt:.1f}s on {nthreads} thread{'s' if nthreads > 1 else ''}, scaled x{9 / rounds:g}"}

    out = leg(threads, 4)
    out["one_thread"] = leg(1, 2)                     # about 10 s per round on one core
    return out


def main():
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
    ap.add_argument("--with-unshared-leg", action="store_true",
                    help="CTR mode, one GPU: after the timed region also time the same instance with every column of the second "
                         "round evaluated, compare the two results on the device and report that rate as "
                         "ctr_shared_columns.value_without_sharing (off by default: its launches would enter a profiler's "
                         "per-kernel averages of this command)")
    ap.add_argument("--no-ctr-share", action="store_true",
                    help="CTR mode: bootstrap every state column of the second round, also those that hold the same ciphertexts "
                         "in every block of the batch (TX_CTR_SHARE=0; same results, for A/B timing)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    if args.no_ctr_share:
        os.environ["TX_CTR_SHARE"] = "0"             # read once by tx_create
    import numpy as np
    import torch
    import __graft_entry__
    __graft_entry__.build()                      # no-op when the in-tree libraries are up to date (file-locked)
    import txhip
    import txclient
    import txdist

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    # one rank per GPU; a rehearsal with more ranks than GPUs (gloo only) wraps around
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
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
    t0 = time.time()
    eng.load_eval_keys(keys.bsk, keys.ksk, keys.auto, keys.ss)
    (eng.load_round_luts if args.mode == "ecb" else eng.load_ctr_luts)(luts)
    t_load = time.time() - t0

    words = 2 * 1024 + 1
    br_ms, br_launches, br_units = [0.0], [0], [0]

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
        return out

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
    br_ms[0], br_launches[0], br_units[0] = 0.0, 0, 0
    last_timing = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    last_timing = eng.timing()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    t_ms, t_launches, t_units = br_ms[0], br_launches[0], br_units[0]      # blind-rotation totals of the timed region
    # the same instance with every column of the CTR schedule's second round evaluated (no sharing): reported beside `value`
    value_unshared = None
    if world == 1 and args.mode == "ctr" and not args.no_ctr_share and args.with_unshared_leg:
        os.environ["TX_CTR_SHARE"] = "0"
        eng2 = txhip.Engine(p, device=dev_index)
        del os.environ["TX_CTR_SHARE"]
        eng2.load_eval_keys(keys.bsk, keys.ksk, keys.auto, keys.ss)
        eng2.load_ctr_luts(luts)
        out2 = torch.empty((n_blocks * 128, words), dtype=torch.int64, device=dev)
        eng2.transcipher(ct, iv, out_device_ptr=out2.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            eng2.transcipher(ct, iv, out_device_ptr=out2.data_ptr())
        torch.cuda.synchronize()
        value_unshared = n_blocks * 2 / (time.perf_counter() - t0)
        same = bool(torch.equal(out2, step()))            # the two schedules give the same ciphertexts, bit for bit
        eng2.close()
        if not same:
            raise SystemExit("bench.py: shared and unshared CTR schedules differ; no number reported")

    if rank == 0:
        _, m_local = txdist.shard(n_blocks, world, 0)
        value = n_blocks * args.steps / dt
        br_avg_ms = t_ms / max(1, t_launches)
        br_avg_units = t_units / max(1, t_launches)
        br_kernel = eng.last_blind_rotate_kernel()         # what run_br (csrc/tx_api.hip) launched, as the library reports it
        # all launches of that kernel in the timed region (they differ in size: in CTR mode the second round bootstraps
        # only the distinct state columns): algorithmic bytes of all of them over the time of all of them
        achieved = br_launch_bytes(t_units, args.n, t_launches) / (t_ms * 1e-3) / 1e9
        br_flops = t_units * args.n * BR_F64_FLOP_PER_UNIT_STEP
        default_mix = (m_local == 64 and args.n == 768 and args.mode == "ctr" and not args.no_ctr_share and br_kernel == "k_blind_rotate5")
        line = {
            "metric": "AES blocks transciphered/sec (medium instance)", "value": value, "unit": "AES blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"medium instance: {n_blocks} AES-128 blocks (512 x u16 per 64 blocks), "
                                   f"AES_TIGHT n={args.n}, {'ECB inverse-cipher schedule of the reference' if args.mode == 'ecb' else 'CTR forward-cipher schedule'}, "
                                   f"full 10 rounds, keys resident, result LWEs left in HBM",
                       "blocks": n_blocks, "blocks_per_gpu": m_local, "parallelism": f"block-sharded x{world}, one gather"},
            "roofline": {"bound": "hbm", "kernel": br_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (BR_TRAFFIC_AVG_DEFAULT_MIX if default_mix else
                                     (BR_TRAFFIC_BYTES_8192.get(br_kernel) if abs(br_avg_units - 8192) < 0.5 and args.n == 768 else None)),
                         "avg_launch_ms": br_avg_ms, "launches": t_launches, "avg_units_per_launch": br_avg_units,
                         "algorithmic_bytes_per_launch": br_launch_bytes(t_units, args.n, t_launches) / max(1, t_launches),
                         "path_algorithmic_GBps": value / world * algorithmic_bytes_per_block(m_local) / 1e9,
                         "path_frac": value / world * algorithmic_bytes_per_block(m_local) / 1e9 / HBM_PEAK_GBS,
                         "fp64_valu": {"achieved_TFLOPs": br_flops / (t_ms * 1e-3) / 1e12, "peak_TFLOPs": FP64_PEAK_TFLOPS,
                                       "frac": br_flops / (t_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                         "vector_pipe_busy_pmc": (BR_VALU_BUSY_PMC.get(br_kernel) if (m_local * 128 == 8192 and args.n == 768) else None),
                         "note": "the path is f64 transform arithmetic, not a stream: HBM fraction is low by construction; "
                                 "fp64_valu counts FP64 flops against the FMA peak at the nominal clock (half of this loop's FP64 "
                                 "instructions are adds, a fifth of its vector instructions 32-bit, and the chip runs it at "
                                 "1.85-1.9 GHz: power-bound); vector_pipe_busy_pmc is the fraction of cycles a SIMD's vector pipe "
                                 "executes an instruction (SQ_ACTIVE_INST_VALU x waves per SIMD / SQ_WAVE_CYCLES, "
                                 "profiles/r2c_pmc_sq_blind_rotate5.csv)"},
            "ctr_shared_columns": (None if args.mode != "ctr" else
                                   {"enabled": not args.no_ctr_share, "value_without_sharing": value_unshared,
                                    "note": "CTR: the state entering the second round is a public function of the counters, so columns fed by "
                                            "the same counter bytes hold the same ciphertexts in every block; each distinct column is "
                                            "bootstrapped once (64 consecutive counters: 268 of 1024 bytes in that round, the other 8 "
                                            "rounds in full); outputs identical to evaluating every column (--no-ctr-share does; "
                                            "--with-unshared-leg times and compares both in one run: 130.0 against 120.4 blocks/s, "
                                            "profiles/r3m_bench_line_with_unshared_leg.json)"}),
            "kernel_ms_last_step": {k: round(v, 3) for k, v in last_timing.items() if k.startswith("ms_")},
            "setup_s": {"client_keygen": round(t_keygen, 2), "key_upload_and_transform": round(t_load, 3)},
        }
        if not args.no_cpu_baseline and world == 1:
            # the GPU box gives one GPU's job a share of 16 host cores; never more threads than that
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(args.n, args.seed, max(1, min(avail, 16)))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
