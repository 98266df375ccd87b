"""Runs the generated COOPERATIVE blind-rotation loop (tools/asmgen/brgen3.py: three waves per unit, wave c owns
polynomial c) as a workgroup of three emulated waves -- one shared LDS image, every s_barrier honoured -- against the
same reference as tests/asmgen_emu_blind_rotate.py (the CPU oracle's primitives).  The waves are scheduled between
barriers first in the order 0, 1, 2 and then 2, 1, 0: a missing barrier around the published transforms shows as a
mismatch in one of the two.  Must match bit for bit; a register read before its load was waited for raises.

    python3 tests/asmgen_emu_blind_rotate3.py [steps=2] [seed=1]
"""
import ctypes as C
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))
sys.path.insert(0, HERE)
import gcn
import brgen
import brgen3
import asmgen_emu_blind_rotate as one

N, H, KP = one.N, one.H, one.KP
TW_BYTES = one.TW_BYTES
TILE_BYTES = 8320
LDS_BYTES = TW_BYTES + 3 * TILE_BYTES + 1024 + 16        # + the dump area of the key prefetch + the workgroup's fairness slot


def run_group_soft(emus, prog, order):
    """soft-barrier form: no s_barrier in the loop; a wave that polls yields at its s_sleep and the next one (in `order`) runs.
    A full round in which every live wave only slept is a deadlock."""
    for e in emus: e.yield_on_sleep = True
    gens = [e.run_gen(prog) for e in emus]
    alive = [True] * len(gens)
    idle_rounds = 0
    while any(alive):
        progressed = False
        for k in order:
            if not alive[k]: continue
            before = emus[k].nexec
            try:
                st = next(gens[k])
                assert st == "sleep", st
            except StopIteration:
                alive[k] = False
            progressed = progressed or emus[k].nexec - before > 16 or not alive[k]
        idle_rounds = 0 if progressed else idle_rounds + 1
        if idle_rounds > 4: raise RuntimeError("every wave is polling: deadlock")


def run_group_ordered(emus, prog, order):
    gens = [e.run_gen(prog) for e in emus]
    alive = [True] * len(gens)
    while any(alive):
        states = []
        for k in order:
            if not alive[k]: continue
            try:
                states.append(next(gens[k]))
            except StopIteration:
                alive[k] = False
                states.append("end")
        if len(set(states)) != 1: raise RuntimeError("waves disagree at a barrier: %s" % states)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = one.oracle()
    H_ = gcn._helpers()
    p = one.p
    tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)
    key = np.zeros((steps, KP, KP, H), np.complex128)
    L.txo_fft_forward_torus_scaled.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    for i in range(steps):
        for q in range(KP):
            for c in range(KP):
                pol = rng.integers(0, 1 << 63, N, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, N, dtype=np.uint64)
                out = np.zeros(N, np.float64)
                L.txo_fft_forward_torus_scaled(N, p(pol, C.c_uint64), p(out, C.c_double))
                key[i, q, c] = out.view(np.complex128)
    lwe = rng.integers(0, 1 << 63, steps + 1, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, steps + 1, dtype=np.uint64)
    if steps >= 2: lwe[1] = 0                                        # a zero mask element: rotation by 0
    ref = one.reference(L, H_, lwe, key, steps)
    rot0 = 2 * N - int(L.txo_modswitch(C.c_uint64(int(lwe[-1])), 10, 3))
    tv = np.zeros(N, np.uint64)
    for i in range(N):
        v = np.uint64(1) << np.uint64(63 - 2 * ((i & 7) + 1))
        tv[i] = (np.uint64(0) - v) if i < N // 2 else v
    SOFT = os.environ.get("BR3_SOFT") == "1"        # k_blind_rotate8's loop: software barriers per unit, LDS step counters
    gen = brgen3.CoopGen(stamps=bool(os.environ.get("BR_STAMPS")), soft=SOFT)
    prog, prog_fair = gen.build(), gen.cu_fair
    FLAGS = TW_BYTES + 3 * TILE_BYTES                # (the emulated workgroup's spare kilobyte) arrival words, then the step counters
    ok = True
    for order in ([0, 1, 2], [2, 1, 0]):
        lds = np.zeros(LDS_BYTES, np.uint8)
        lw = lds.view(np.complex128)
        lw[:8 * 64] = tw[:8].reshape(-1)
        for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
        # what the kernel's HIP part leaves in the tiles: zero mask polynomials, the rotated test vector as the body
        lds[TW_BYTES + 2 * TILE_BYTES: TW_BYTES + 2 * TILE_BYTES + 8192].view(np.uint64)[:] = one.rotate(tv, rot0)
        CT, ROW, OUT, TAB = 0x1000_0000, 0x2000_0000, 0x3000_0000, 0x5000_0000
        # the CU's entry of the fairness table: this workgroup's word (slot 2), a competitor of this launch two steps ahead of
        # the start, one that has finished, and a stale word of an older launch -- every priority path is taken over three steps
        epoch = 0x0123 << 16
        tab_buf = np.array([epoch | 2, epoch | 0xFFFF, 0xDEAD0007, (0x0122 << 16) | 1], np.uint32).view(np.uint8).copy()
        ct_buf = lwe.copy().view(np.uint8)
        key_buf = key.reshape(-1).view(np.uint8).copy()
        out_buf = np.zeros(KP * N * 8, np.uint8)
        fault_buf = np.zeros(4, np.uint8)           # the context's fault word (soft-barrier form)
        SG = brgen.SG
        emus = []
        for c in range(3):
            e = gcn.Emu(lds, [(CT, ct_buf), (ROW, key_buf), (OUT, out_buf), (TAB, tab_buf)])
            e.s[brgen3.FTAB.i] = TAB; e.s[brgen3.FTAB.i + 1] = 0
            # key prefetch: piece 5 + c of the row PF_ROWS steps ahead, relative to this wave's column pointer
            pfoff = brgen3.PF_ROWS * KP * KP * H * 16 + (5 + c) * 1024 - c * H * 16
            e.s[brgen3.FSLOT.i] = (8 | (pfoff << 4)) & 0xFFFFFFFF; e.s[brgen3.FEPOCH.i] = epoch
            for s, val in ((SG.CT, CT), (SG.ROW, ROW + c * H * 16), (SG.OUT, OUT + c * N * 8)):
                e.s[s.i] = val & 0xFFFFFFFF; e.s[s.i + 1] = val >> 32
            e.s[SG.NSTEP.i] = steps
            e.s[SG.TILE.i] = TW_BYTES + c * TILE_BYTES
            e.s[brgen3.TILE0.i] = TW_BYTES
            e.s[SG.TIEMODE.i] = 0 if os.environ.get('EMU_TIE') == 'tfhe' else 1
            if SOFT:
                e.g.append((0x6000_0000, fault_buf))
                e.s[brgen3.SFAULT.i] = 0x6000_0000; e.s[brgen3.SFAULT.i + 1] = 0
                e.s[brgen3.SFLAGS.i] = FLAGS; e.s[brgen3.SMYFLAG.i] = FLAGS + 4 * c
                e.s[brgen3.SPROG.i] = FLAGS + 16; e.s[brgen3.SMYPROG.i] = FLAGS + 16 + 4 * 1        # this unit is unit 1 of its workgroup
            if os.environ.get("BR_STAMPS"):
                dbg = np.zeros(128, np.uint8); e.g.append((0x4000_0000 + 0x1000 * c, dbg))
                e.s[brgen.ST_DBG.i] = 0x4000_0000 + 0x1000 * c; e.s[brgen.ST_DBG.i + 1] = 0
            emus.append(e)
        t = time.time()
        if SOFT:
            # the other units' step counters: one two steps ahead, one finished, one behind (every priority path over three steps)
            lds[FLAGS + 16: FLAGS + 32].view(np.uint32)[:] = [2, 0, 0xFFFFFFF0, 0]
            if os.environ.get("BR3_SOFT_SPIN"):
                # the give-up path (generator built with a poll budget of a few): wave 0 runs ALONE, its partners never arrive --
                # it must reach the end of the program all the same, and the fault word must say so
                emus[0].yield_on_sleep = True
                for _ in emus[0].run_gen(prog): pass
                gave_up = int(fault_buf.view(np.uint32)[0]) == 1
                print("soft barriers, wave 0 alone with a poll budget of %s: ended, fault word %d -> %s" %
                      (os.environ["BR3_SOFT_SPIN"], fault_buf.view(np.uint32)[0], "GAVE-UP-REPORTED" if gave_up else "NOT REPORTED"))
                sys.exit(0 if gave_up else 1)
            run_group_soft(emus, prog, order)
            same = np.array_equal(out_buf.view(np.uint64).reshape(KP, N), ref)
            cnt = lds[FLAGS: FLAGS + 32].view(np.uint32)
            same = same and list(cnt[:3]) == [2 * steps] * 3 and cnt[4 + 1] == 0xFFFFFFF0 and cnt[4] == 2 and cnt[4 + 2] == 0xFFFFFFF0
            same = same and int(fault_buf.view(np.uint32)[0]) == 0
            ok = ok and same
            print("soft barriers, steps %d seed %d order %s: %d instructions per wave, %.1f s -> %s" %
                  (steps, seed, order, emus[0].nexec, time.time() - t, "BIT-IDENTICAL" if same else "MISMATCH"))
            continue
        run_group_ordered(emus, prog, order)
        got = out_buf.view(np.uint64).reshape(KP, N)
        same = np.array_equal(got, ref)
        if prog_fair:            # the workgroup's word: "finished" at the end, the others untouched
            tab = tab_buf.view(np.uint32)
            same = same and tab[2] == (epoch | 0xFFFF) and tab[0] == (epoch | 2) and tab[3] == ((0x0122 << 16) | 1)
        ok = ok and same
        top = max(max((x.i + x.n) for x in i.a if isinstance(x, gcn.V)) for i in prog.ins if any(isinstance(x, gcn.V) for x in i.a))
        print("steps %d seed %d order %s: %d instructions per wave, top VGPR %d, %.1f s -> %s" %
              (steps, seed, order, emus[0].nexec, top, time.time() - t, "BIT-IDENTICAL" if same else "MISMATCH"))
        if not same:
            for q in range(KP):
                bad = np.nonzero(got[q] != ref[q])[0]
                print("  poly %d: %d mismatches" % (q, bad.size), bad[:8], [hex(int(x)) for x in got[q][bad[:3]]], [hex(int(x)) for x in ref[q][bad[:3]]])
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
