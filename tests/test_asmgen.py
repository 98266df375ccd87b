"""CPU tests of the generated-assembly CMux loop (tools/asmgen): the emitted instruction stream, executed in the
lane-exact emulator, equals a reference built from the oracle's primitives bit for bit (the emulator also fails on
any register read before its load was waited for), and the files under csrc/ are what the generator emits now."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "asmgen")
CSRC = os.path.join(ROOT, "temp-fhe-transciphering_amd", "csrc")


def test_generated_loop_equals_oracle_in_the_emulator(oracle):
    for steps, seed in ((3, 1), (2, 77)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate.py"), str(steps), str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, r.stdout + r.stderr


def test_fair_variant_is_the_same_arithmetic_on_both_priority_paths(oracle):
    for fair in ("0", "1"):     # partner never / always ahead
        env = dict(os.environ, BR_FAIR=fair)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate.py"), "2", "6"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, r.stdout + r.stderr


def test_digit_forms_are_the_same_arithmetic(oracle):
    """the rotated difference of rounds 2-3 (conditional negation with the carry forms, nine instructions per word) and the digit's
    carry-in add (e as the even-lane mask in vcc) stay generator options next to the seven-instruction form: same values"""
    for env in ({"BR_DIGIT": "long"}, {"BR_EREG": "0"}, {"BR_DIGIT": "long", "BR_EREG": "0"}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate.py"), "2", "9"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **env))
        assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, str(env) + r.stdout + r.stderr
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate3.py"), "2", "9"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.count("BIT-IDENTICAL") == 2, str(env) + r.stdout + r.stderr


def test_cooperative_loop_equals_oracle_three_waves(oracle):
    """tools/asmgen/brgen3.py (k_blind_rotate6: three waves per unit, one per polynomial): a workgroup of three emulated
    waves sharing the LDS image and meeting at both barriers of every step, scheduled in both orders between barriers,
    against the same oracle-built reference as the one-wave loops; the stamped variant is the same arithmetic; without
    the barriers (timing ablation) the emulation must NOT match -- the test can see a missing barrier"""
    exe = os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate3.py")
    for steps, seed, env in ((3, 1, {}), (2, 77, {}), (2, 5, {"BR_STAMPS": "1"})):
        r = subprocess.run([sys.executable, exe, str(steps), str(seed)], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.count("BIT-IDENTICAL") == 2, r.stdout + r.stderr
    r = subprocess.run([sys.executable, exe, "2", "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, BR3_ABL="B"))
    assert r.returncode == 1 and "MISMATCH" in r.stdout, r.stdout + r.stderr


def test_cooperative_loop_with_soft_barriers_equals_oracle(oracle):
    """the soft-barrier form of that loop (k_blind_rotate8: the unit's three waves meet at arrival words in LDS they poll, the
    units of a workgroup are kept level through step counters in LDS; csrc/tx_br3s_asm.inc): three emulated waves, a wave that
    polls in vain yields to the next one, both scheduling orders; the arrival words end at two per step, the unit's counter at
    "finished", the other units' counters untouched; without the barriers (ablation) it must NOT match.  The emulator also
    refuses what the first version of this loop did on the hardware: a v_readfirstlane_b32 of a register the instruction
    before it wrote (one wait state on gfx940+: it read the OLD value and every poll passed at once)"""
    exe = os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate3.py")
    for steps, seed in ((3, 1), (2, 77)):
        r = subprocess.run([sys.executable, exe, str(steps), str(seed)], capture_output=True, text=True, timeout=600, env=dict(os.environ, BR3_SOFT="1"))
        assert r.returncode == 0 and r.stdout.count("BIT-IDENTICAL") == 2, r.stdout + r.stderr
    r = subprocess.run([sys.executable, exe, "2", "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, BR3_SOFT="1", BR3_ABL="B"))
    assert r.returncode == 1 and "MISMATCH" in r.stdout, r.stdout + r.stderr
    # a wave whose partners never arrive gives up after its poll budget, ENDS, and says so in the context's fault word (the host
    # then fails the call: csrc/tx_api.hip timing_collect)
    r = subprocess.run([sys.executable, exe, "2", "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, BR3_SOFT="1", BR3_SOFT_SPIN="3"))
    assert r.returncode == 0 and "GAVE-UP-REPORTED" in r.stdout, r.stdout + r.stderr
    sys.path.insert(0, GEN)
    import gcn
    from gcn import V, S, Lit
    p = gcn.Prog()
    p.emit("v_mov_b32", V(13), Lit(5))
    p.emit("v_readfirstlane_b32", S(46), V(13))
    with pytest.raises(gcn.Poison, match="1 wait state"):
        gcn.Emu(np.zeros(64, np.uint8), []).run(p)


@pytest.mark.experiments
def test_paired_loops_equal_oracle_two_waves(oracle):
    """tools/asmgen/brgen2.py (two waves per unit, A = the mask polynomials and columns 0, 1, B = the body and column 2; two
    different programs; an experiment that is NOT in the library -- faster than the one-wave kernel standalone, slower under
    the pipeline's sustained load, DESIGN_HISTORY.md section 11 -- kept runnable): two emulated waves sharing the LDS image and
    meeting at both barriers of every step, in both scheduling orders, against the oracle-built reference; without the
    barriers it must NOT match"""
    exe = os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate2.py")
    for steps, seed in ((3, 1), (2, 77)):
        r = subprocess.run([sys.executable, exe, str(steps), str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.count("BIT-IDENTICAL") == 2, r.stdout + r.stderr
    r = subprocess.run([sys.executable, exe, "2", "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, BR2_ABL="B"))
    assert r.returncode == 1 and "MISMATCH" in r.stdout, r.stdout + r.stderr


@pytest.mark.experiments
def test_cooperative_trace_experiment_equals_oracle_two_waves(oracle):
    """tools/asmgen/trgen2.py (two waves per unit: A the low-split pass, B the high-split pass, mask polynomials in
    registers; an experiment that is NOT in the library -- it measured slower than the one-wave loop, DESIGN_HISTORY.md section 11 --
    kept runnable): two emulated waves in both scheduling orders against txo_trace() with real keys, ten steps"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_trace2.py"), "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("BIT-IDENTICAL") == 2, r.stdout + r.stderr


def test_stamped_variant_is_the_same_arithmetic(oracle):
    env = dict(os.environ, BR_STAMPS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_blind_rotate.py"), "2", "5"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, r.stdout + r.stderr


def test_generated_trace_loop_equals_oracle_four_waves(oracle):
    """a whole workgroup: four emulated waves share the LDS image, fill the key ring by LDS-DMA quarters and meet at
    every s_barrier; result against txo_trace() with real automorphism keys"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_trace.py"), "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, r.stdout + r.stderr


def test_generated_scheme_switch_equals_oracle_four_waves(oracle):
    """tools/asmgen/ssgen.py: a workgroup of four emulated waves against txo_scheme_switch() + txo_ggsw_to_fourier()"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_ss.py"), "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, r.stdout + r.stderr


@pytest.mark.experiments
def test_generated_lut_tree_equals_oracle(oracle):
    """tools/asmgen/lutgen.py (k_lut_eval4: since round 5 only in the experiment build libtfhe_mi355x_exp.so, where TX_LUT_VARIANT=4
    selects it; the product library does not contain it, so its emulator run rides with the experiments): eight CMux steps in the emulator against a reference
    built from the oracle's primitives on the whole accumulator, and against txo_lut_eval's four extracted LWEs"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_lut.py"), "8", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("BIT-IDENTICAL") == 2, r.stdout + r.stderr


def test_trace_loop_variants_are_the_same_arithmetic(oracle):
    """the generator's alternatives kept for experiments: keys through the workgroup's shared LDS ring instead of streamed by
    every wave (the default until round 4), a deeper ring, transforms without the spread exchange stores"""
    for env in ({"TR_KEYS": "ring"}, {"TR_KEYS": "ring", "TR_SLOTS": "6"}, {"TR_PIPE": "0"}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asmgen_emu_trace.py"), "7"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **env))
        assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, str(env) + r.stdout + r.stderr


def test_committed_assembly_is_current(tmp_path):
    subprocess.check_call([sys.executable, os.path.join(GEN, "brgen.py"), "--out", str(tmp_path)])
    subprocess.check_call([sys.executable, os.path.join(GEN, "trgen.py"), "--out", str(tmp_path)])
    subprocess.check_call([sys.executable, os.path.join(GEN, "fftconst.py"), "--out", str(tmp_path)])
    subprocess.check_call([sys.executable, os.path.join(GEN, "ssgen.py"), "--out", str(tmp_path)])
    subprocess.check_call([sys.executable, os.path.join(GEN, "brgen3.py"), "--out", str(tmp_path)])
    subprocess.check_call([sys.executable, os.path.join(GEN, "lutgen.py"), "--out", str(tmp_path)])
    # (the *_stamps.inc diagnostic variants the generators also write are build-time output, not tracked)
    for f in ("tx_fft_consts.inc", "tx_ss_asm.inc", "tx_br_asm.inc", "tx_br_asm_fair.inc", "tx_br_asm_clobbers.inc",
              "tx_trace_asm.inc", "tx_trace_asm_clobbers.inc", "tx_br3_asm.inc", "tx_br3s_asm.inc", "tx_br3_asm_clobbers.inc", "tx_br3_asm_consts.inc", "tx_lut_asm.inc"):
        assert open(os.path.join(tmp_path, f)).read() == open(os.path.join(CSRC, f)).read(), f + " is stale: run tools/asmgen/brgen.py --out csrc"


def test_twist_constants_are_the_oracles(oracle):
    """the literals the HIP templates and the generated assembly carry (tools/asmgen/fftconst.py, correctly rounded
    from nested radicals) are the oracle's C16[] (long-double cos / sin rounded once) bit for bit"""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, GEN)
    import fftconst
    L = oracle.lib()
    L.txo_twist_consts_1024.restype = C.POINTER(C.c_double)
    got = np.ctypeslib.as_array(L.txo_twist_consts_1024(), shape=(8, 2))
    assert np.array_equal(got.view(np.uint64), np.array(fftconst.C16, np.float64).view(np.uint64))


def _code_object_kernels(lib="libtfhe_mi355x.so"):
    """{kernel name: {private_segment_fixed_size, vgpr_count, ...}} of the gfx950 code object inside the built library"""
    import re
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    so = os.path.join(ROOT, "temp-fhe-transciphering_amd", lib)
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", co], text=True)
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+\.(\w+):\s+(\S+)", line)
        if not m: continue
        if m.group(1) == "name" and m.group(2).startswith("_ZN3txd"): cur = out.setdefault(m.group(2), {})
        elif cur is not None and m.group(1) in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count"):
            cur[m.group(1)] = int(m.group(2))
    return out


def test_scratch_of_the_built_kernels():
    """the BUILT code object, not only the generator's stream: the kernels around the generated trace / scheme-switching
    loops have no scratch frame at all; k_blind_rotate5's frame (what hipcc keeps live across a loop that clobbers all 256
    vector registers; stored and reloaded once per unit, outside the loop) stays within 128 bytes per lane; the two-wave
    kernels are built for exactly 256 registers"""
    k = _code_object_kernels()
    def find(stem, k=k):
        hits = [v for n, v in k.items() if stem in n]
        assert len(hits) == 1, (stem, sorted(k))
        return hits[0]
    for stem in ("16k_extract_trace4", "16k_scheme_switch4", "22k_trace_scheme_switch4", "15k_blind_rotate6", "15k_blind_rotate8"):
        assert find(stem)["private_segment_fixed_size"] == 0, stem
    assert find("15k_blind_rotate5")["private_segment_fixed_size"] <= 128
    assert find("15k_blind_rotate5")["vgpr_count"] == 256
    assert find("15k_blind_rotate6")["vgpr_count"] == 168            # three waves per SIMD
    assert find("15k_blind_rotate8")["vgpr_count"] == 168            # twelve waves of one workgroup: three per SIMD
    # the product library holds none of the superseded kernels; the experiment build holds them, built to the same budgets
    assert not [n for n in k if "15k_blind_rotate4" in n or "15k_blind_rotate7" in n or "11k_lut_eval4" in n or "15k_extract_traceE" in n]
    kx = _code_object_kernels("libtfhe_mi355x_exp.so")
    assert find("15k_blind_rotate4", kx)["private_segment_fixed_size"] == 0 and find("15k_blind_rotate4", kx)["vgpr_count"] == 256
    assert find("15k_blind_rotate7", kx)["vgpr_count"] == 168 and find("15k_blind_rotate5", kx)["vgpr_count"] == 256 and find("22k_trace_scheme_switch4", kx)["private_segment_fixed_size"] == 0


def test_register_budget_and_no_scratch():
    """the generated loops fit 256 VGPRs and issue no scratch instruction (what hipcc keeps around them is bounded by
    test_scratch_of_the_built_kernels): two waves per SIMD is the whole point"""
    sys.path.insert(0, GEN)
    import brgen, brgen3, trgen, ssgen
    _check_budget([(brgen.Gen().build(), 256), (trgen.TraceGen().build(), 256), (ssgen.SsGen().build(), 256),
                   (brgen3.CoopGen().build(), 168)])


@pytest.mark.experiments
def test_register_budget_of_the_experiments():
    sys.path.insert(0, GEN)
    import brgen2, trgen2
    _check_budget([(trgen2.Trace2Gen().build(), 256), (brgen2.PairGen("A").build(), 256), (brgen2.PairGen("B").build(), 192)])


def _check_budget(progs):
    import gcn
    for prog, budget in progs:
        top = 0
        for i in prog.ins:
            for x in i.a:
                if isinstance(x, gcn.V): top = max(top, x.i + x.n)
                if isinstance(x, gcn.S) and x.i not in (124, 106, 107): assert x.i + x.n <= 96, "s96..s101 are flat scratch / xnack; vcc and m0 are named"
        assert top <= budget
        assert not any(i.op.startswith("scratch_") for i in prog.ins)


def _tie_words(B, L, rng):
    """64 torus words, most of them exact TOP-LEVEL ties of Decomp<B, L> (the digit of level 1 is B/2 with nothing above it),
    each with random low bits below the rounding point and random lower-level digits that do not carry into the top"""
    total = B * L
    words = np.zeros(64, np.uint64)
    for lane in range(64):
        top = (1 << (B - 1)) if lane % 8 != 7 else int(rng.integers(0, 1 << B))           # seven ties in eight lanes
        lower = 0
        for lev in range(L - 1):                                                           # lower digits < B/2: no carry upwards
            lower = (lower << B) | int(rng.integers(0, 1 << (B - 1)))
        state = (top << (B * (L - 1))) | lower
        junk = int(rng.integers(0, 1 << (63 - total))) if total < 63 else 0                # below the rounding bit: rounds down
        words[lane] = np.uint64(((state << (64 - total)) | junk) & 0xFFFFFFFFFFFFFFFF)
    return words


@pytest.mark.parametrize("tfhe", [False, True])
def test_generated_digit_sequences_follow_the_tie_mode_operand(oracle, tfhe):
    """tx_set_decomposer_tie_rule: the tie-mode operand of the generated loops (1 = balanced, 0 = tfhe-rs' SignedDecomposer) reaches
    the digit sequences.  On words that ARE top-level ties -- random data meets one per 2^23 (blind rotation), 2^13 (trace) or 2^17
    (scheme switch) coefficients -- the blind rotation's digit_pair and the trace / scheme-switching digits_bl give, lane by lane,
    what txo_decompose_idx gives under the same rule; and the two rules differ exactly on the odd lanes that tie"""
    sys.path.insert(0, GEN)
    import ctypes as C
    import gcn, brgen, trgen, ssgen
    from gcn import V, Lit
    L = oracle.lib()
    L.txo_decompose_idx.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int64)]
    rng = np.random.default_rng(7)
    mode = 0 if tfhe else 1
    oracle.set_top_digit_rule(tfhe)
    try:
        def want(words, B, Lv):
            out = np.zeros((64, Lv), np.int64)
            d = (C.c_int64 * Lv)()
            for lane in range(64):
                L.txo_decompose_idx(C.c_uint64(int(words[lane])), B, Lv, lane & 1, d)
                out[lane] = list(d)
            return out

        # ---- blind rotation: digit_pair (B = 23, one level) of brgen.Gen, the prologue's own mode-dependent registers
        g = brgen.Gen()
        e = g.e
        e("v_mbcnt_lo_u32_b32", brgen.LANE, Lit(-1), Lit(0)); e("v_mbcnt_hi_u32_b32", brgen.LANE, Lit(-1), brgen.LANE)
        g.tie_lane(brgen.T0.lo())
        e("v_lshlrev_b32", brgen.EREG, Lit(9), brgen.T0.lo()); e("v_add_u32", brgen.EREG, Lit(-0x100), brgen.EREG)
        e("v_xor_b32", g.e_reg, Lit(1), brgen.T0.lo())
        g.even_lanes_to_vcc()
        g.digit_pair(V(20, 2), V(22), V(23), V(24), V(26, 2), brgen.CARRY[0])
        words = _tie_words(23, 1, rng)
        emu = gcn.Emu(np.zeros(64, np.uint8), [])
        emu.s[brgen.SG.TIEMODE.i] = mode
        if g.neg_acc:         # the loop's words are those of -acc and the sign mask is inverted: d = own' + word' with own' = 0
            emu.v[24][:] = 0
        emu.v[20][:] = (words & np.uint64(0xFFFFFFFF)).astype(np.uint32); emu.v[21][:] = (words >> np.uint64(32)).astype(np.uint32)
        emu.run(g.p)
        got = (emu.v[20].astype(np.uint64) | (emu.v[21].astype(np.uint64) << np.uint64(32))).view(np.float64)
        exp = want(words, 23, 1)[:, 0]
        assert np.array_equal(got, exp.astype(np.float64)), (got[:8], exp[:8])
        ties = np.array([lane % 8 != 7 for lane in range(64)])
        assert (np.abs(exp[ties]) == 1 << 22).all()
        assert ((exp[ties & (np.arange(64) % 2 == 1)] > 0) == tfhe).all()          # odd lanes: -B/2 balanced, +B/2 under tfhe-rs' rule

        # ---- trace (B = 13, 3 levels) and scheme switch (B = 17, 2 levels): digits_bl with the prologue's tie mask
        for gen, B, Lv in ((trgen.TraceGen(), 13, 3), (ssgen.SsGen(), 17, 2)):
            e = gen.e
            e("v_mbcnt_lo_u32_b32", brgen.LANE, Lit(-1), Lit(0)); e("v_mbcnt_hi_u32_b32", brgen.LANE, Lit(-1), brgen.LANE)
            gen.tie_lane(brgen.T0.lo(), trgen.TS.TIEMODE)
            e("v_lshlrev_b32", trgen.TIE, Lit(B - 1), brgen.T0.lo())
            e("s_mov_b32", trgen.TS.C_1P24, Lit(1 << (63 - B * Lv)))
            outs = [V(40 + 2 * k, 2) for k in range(Lv)]
            gen.digits_bl(V(20, 2), None, outs, [V(30), V(31), V(32)], brgen.CARRY[0], B, Lv)
            words = _tie_words(B, Lv, rng)
            emu = gcn.Emu(np.zeros(64, np.uint8), [])
            emu.s[trgen.TS.TIEMODE.i] = mode
            emu.v[20][:] = (words & np.uint64(0xFFFFFFFF)).astype(np.uint32); emu.v[21][:] = (words >> np.uint64(32)).astype(np.uint32)
            emu.run(gen.p)
            exp = want(words, B, Lv)
            for k in range(Lv):
                got = (emu.v[40 + 2 * k].astype(np.uint64) | (emu.v[41 + 2 * k].astype(np.uint64) << np.uint64(32))).view(np.float64)
                assert np.array_equal(got, exp[:, k].astype(np.float64)), (B, k, got[:8], exp[:8, k])
            top = exp[:, Lv - 1]
            assert (np.abs(top[ties]) == 1 << (B - 1)).all()
            assert ((top[ties & (np.arange(64) % 2 == 1)] > 0) == tfhe).all()
    finally:
        oracle.set_top_digit_rule(False)
