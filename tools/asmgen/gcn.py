"""Tiny gfx950 assembly DSL + lane-exact emulator for the generated blind-rotation loop.

The generator (brgen.py) builds a list of `Ins` objects; `render()` turns them into the text of ONE
inline-asm block, `Emu` executes them on 64 lanes with numpy so that the instruction stream can be
checked bit-for-bit against the CPU oracle WITHOUT a GPU.  The emulator also models the two memory
counters: a load's destination registers are poisoned until an `s_waitcnt` retires the load, and any
instruction that reads a poisoned register raises -- a missing or too-weak wait is a test failure
here, not a silent race on the hardware.

Only the instruction subset the generator uses is implemented.  Development tool: not part of the product
path, not shipped in the library (the generated text is).
"""
import ctypes
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------ operands
class V:
    """vector register (32-bit) or aligned tuple of `n` registers starting at `i`"""
    __slots__ = ("i", "n", "neg")

    def __init__(self, i, n=1, neg=False):
        assert 0 <= i and i + n <= 256, (i, n)
        if n == 2: assert i % 2 == 0, "64-bit operands must be even-aligned: v%d" % i
        if n == 4: assert i % 2 == 0, "128-bit tuples must be even-aligned: v%d" % i
        self.i, self.n, self.neg = i, n, neg

    def __neg__(self): return V(self.i, self.n, not self.neg)
    def lo(self): return V(self.i)
    def hi(self): return V(self.i + 1)
    def sub(self, k, n=2): return V(self.i + k, n)

    def __str__(self):
        s = "v%d" % self.i if self.n == 1 else "v[%d:%d]" % (self.i, self.i + self.n - 1)
        return "-" + s if self.neg else s


class S:
    __slots__ = ("i", "n", "neg")

    def __init__(self, i, n=1, neg=False):
        if n == 2: assert i % 2 == 0
        self.i, self.n, self.neg = i, n, neg

    def __neg__(self): return S(self.i, self.n, not self.neg)
    def lo(self): return S(self.i)
    def hi(self): return S(self.i + 1)

    def __str__(self):
        if self.i == 124 and self.n == 1: return "m0"
        if self.i == 106: return "vcc" if self.n == 2 else "vcc_lo"
        if self.i == 107 and self.n == 1: return "vcc_hi"
        s = "s%d" % self.i if self.n == 1 else "s[%d:%d]" % (self.i, self.i + self.n - 1)
        return "-" + s if self.neg else s


class A:
    """accumulation register (AGPR): reachable by VALU only through v_accvgpr_read/write"""
    __slots__ = ("i", "n")

    def __init__(self, i, n=1):
        assert 0 <= i and i + n <= 256
        self.i, self.n = i, n

    def __str__(self): return "a%d" % self.i if self.n == 1 else "a[%d:%d]" % (self.i, self.i + self.n - 1)


M0 = S(124)
VCC_LO, VCC_HI, VCC64 = S(106), S(107), S(106, 2)      # gfx9 numbering of the vcc halves: what SALU instructions write


class Lit:
    """32-bit literal / inline constant; `f64` marks an inline double constant (0.0, 1.0, ...)"""
    __slots__ = ("v", "f64")

    def __init__(self, v, f64=None): self.v, self.f64 = v, f64

    def __str__(self):
        if self.f64 is not None: return "0" if float(self.f64) == 0.0 else repr(float(self.f64))
        return str(self.v) if -16 <= self.v <= 64 else hex(self.v & 0xFFFFFFFF)


VCC = "vcc"


class Ins:
    __slots__ = ("op", "a", "kw", "cmt")

    def __init__(self, op, *a, **kw):
        self.cmt = kw.pop("cmt", None)
        self.op, self.a, self.kw = op, a, kw

    def text(self):
        if self.op == "label": return "%s:" % self.a[0]
        if self.op == "comment": return "; %s" % self.a[0]
        if self.op == "s_waitcnt":
            parts = []
            if self.kw.get("vm") is not None: parts.append("vmcnt(%d)" % self.kw["vm"])
            if self.kw.get("lgkm") is not None: parts.append("lgkmcnt(%d)" % self.kw["lgkm"])
            return "s_waitcnt " + " ".join(parts)
        ops = ", ".join(str(x) for x in self.a)
        mods = ""
        if "offset" in self.kw and self.kw["offset"]: mods += " offset:%d" % self.kw["offset"]
        if self.kw.get("dpp"): mods += " " + self.kw["dpp"]
        if self.kw.get("cache"): mods += " " + self.kw["cache"]            # e.g. "sc0 sc1": a load that must see other workgroups' stores
        if self.kw.get("off"): ops = ops  # placeholder
        t = self.op + (" " + ops if ops else "") + mods
        return t


# ------------------------------------------------------------------------------------------ program
class Prog:
    def __init__(self):
        self.ins = []

    def emit(self, op, *a, **kw):
        self.ins.append(Ins(op, *a, **kw))

    def label(self, name): self.ins.append(Ins("label", name))
    def comment(self, t): self.ins.append(Ins("comment", t))

    def render(self):
        out = []
        for i in self.ins:
            t = i.text()
            out.append(t if i.op in ("label",) else "  " + t)
        return out

    def count(self):
        c = {}
        for i in self.ins:
            if i.op in ("label", "comment"): continue
            c[i.op] = c.get(i.op, 0) + 1
        return c


# ------------------------------------------------------------------------------------------ emulator
def _helpers():
    so = os.path.join(HERE, "_emu_helpers.so")
    src = os.path.join(HERE, "emu_helpers.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src, "-lm"])
    L = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    L.vfma.argtypes = [ctypes.c_int, dp, dp, dp, dp]
    return L


class Poison(Exception):
    pass


class Emu:
    """one wavefront; LDS is a bytearray shared with whoever set it up; global memory is a dict
    base address -> numpy uint8 array (addresses are matched by range)"""

    def __init__(self, lds_bytes, globals_):
        self.v = np.zeros((256, 64), np.uint32)
        self.acc = np.zeros((256, 64), np.uint32)    # AGPRs
        self.pend_v = np.zeros(256, bool)            # destination of an un-retired load
        self.s = np.zeros(128, np.uint32)
        self.vcc = np.zeros(64, bool)
        self.smask = {}
        self.scc = False
        self.lds = lds_bytes                         # numpy uint8
        self.g = globals_                            # list of (base, np.uint8 array)
        self.vmq = []                                # in-flight vector memory ops: (regs, apply_fn)
        self.lgq = []
        self.H = _helpers()
        self.nexec = 0
        self.stats = {}
        self.valu_recent = [set(), set()]            # VGPRs VALU instructions wrote one / two wait states ago

    # ---- register access
    def rd32(self, x):
        if isinstance(x, V):
            if self.pend_v[x.i]: raise Poison("read of v%d before its load was waited for" % x.i)
            return self.v[x.i].copy()
        if isinstance(x, S): return np.full(64, self.s[x.i], np.uint32)
        if isinstance(x, Lit): return np.full(64, x.v & 0xFFFFFFFF, np.uint32)
        raise TypeError(x)

    def rd64(self, x):
        if isinstance(x, Lit):
            assert x.f64 is not None
            r = np.full(64, float(x.f64))
            return r
        if isinstance(x, V):
            assert x.n == 2, x
            if self.pend_v[x.i] or self.pend_v[x.i + 1]: raise Poison("read of v[%d:%d] before its load was waited for" % (x.i, x.i + 1))
            bits = self.v[x.i].astype(np.uint64) | (self.v[x.i + 1].astype(np.uint64) << np.uint64(32))
        else:
            assert x.n == 2
            bits = np.full(64, int(self.s[x.i]) | (int(self.s[x.i + 1]) << 32), np.uint64)
        r = bits.view(np.float64).copy()
        return -r if x.neg else r

    def wr32(self, d, val):
        assert isinstance(d, V) and d.n == 1
        if self.pend_v[d.i]: raise Poison("write to v%d while a load into it is in flight" % d.i)
        self.v[d.i] = val.astype(np.uint32)

    def wr64(self, d, val):
        assert isinstance(d, V) and d.n == 2
        if self.pend_v[d.i] or self.pend_v[d.i + 1]: raise Poison("write to v[%d:%d] while a load into it is in flight" % (d.i, d.i + 1))
        bits = np.ascontiguousarray(val, np.float64).view(np.uint64)
        self.v[d.i] = (bits & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        self.v[d.i + 1] = (bits >> np.uint64(32)).astype(np.uint32)

    def fma(self, a, b, c):
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64); c = np.ascontiguousarray(c, np.float64)
        o = np.empty(64, np.float64)
        dp = ctypes.POINTER(ctypes.c_double)
        self.H.vfma(64, a.ctypes.data_as(dp), b.ctypes.data_as(dp), c.ctypes.data_as(dp), o.ctypes.data_as(dp))
        return o

    # ---- memory
    def gmem(self, addr, nbytes):
        for base, arr in self.g:
            if base <= addr and addr + nbytes <= base + arr.size:
                return arr, addr - base
        raise IndexError("global access outside every buffer: 0x%x +%d" % (addr, nbytes))

    def retire(self, q, keep):
        while len(q) > keep:
            regs, fn = q.pop(0)
            fn()
            for r in regs: self.pend_v[r] = False

    # ---- execution
    def run_gen(self, prog, max_ins=50_000_000):
        """executes until the end; yields at every s_barrier (the caller decides when the wave may go on)"""
        ins = prog.ins
        labels = {i.a[0]: k for k, i in enumerate(ins) if i.op == "label"}
        pc = 0
        while pc < len(ins):
            i = ins[pc]
            pc += 1
            if i.op in ("label", "comment"): continue
            self.nexec += 1
            self.stats[i.op] = self.stats.get(i.op, 0) + 1
            if self.nexec > max_ins: raise RuntimeError("instruction budget exceeded")
            if i.op == "s_barrier":
                yield "barrier"
                continue
            if i.op == "s_sleep" and getattr(self, "yield_on_sleep", False):
                yield "sleep"              # a polling wave leaves the SIMD: the scheduler of the test runs another wave
                continue
            j = self.step(i)
            if j is not None: pc = labels[j]
        # program end: everything must have been waited for
        if self.vmq or self.lgq: raise Poison("program ended with memory operations in flight")

    def run(self, prog, max_ins=50_000_000):
        for _ in self.run_gen(prog, max_ins): pass          # a wave alone: barriers are trivially passed

    def step(self, i):
        op, a, kw = i.op, i.a, i.kw
        f = getattr(self, "op_" + op, None)
        if f is None: raise NotImplementedError(op)
        r = f(*a, **kw)
        # VGPRs written by VALU instructions, for the last two wait states (s_nop N counts N + 1)
        wrote = set()
        if op.startswith("v_") and a and isinstance(a[0], V):
            wrote = set(range(a[0].i, a[0].i + a[0].n))
            if "_swap_" in op: wrote |= set(range(a[1].i, a[1].i + a[1].n))
        n = (int(a[0].v) + 1) if op == "s_nop" else 1
        for _ in range(min(n, 2)): self.valu_recent = [wrote, self.valu_recent[0]]; wrote = set()
        return r

    # ---- SALU
    def sval(self, x):
        if isinstance(x, S): return int(self.s[x.i])
        if isinstance(x, Lit): return x.v & 0xFFFFFFFF
        raise TypeError(x)

    def op_s_mov_b32(self, d, a):
        self.s[d.i] = self.sval(a)
        if d.i in (106, 107): self._vcc_from_s()

    def _vcc_from_s(self):
        bits = int(self.s[106]) | (int(self.s[107]) << 32)
        self.vcc = np.array([(bits >> l) & 1 for l in range(64)], bool)

    def op_s_not_b64(self, d, a):
        assert d.i == 106 and a.i == 106 and d.n == 2, "only vcc is modelled as a 64-bit scalar mask"
        self.vcc = ~self.vcc
        bits = sum(1 << l for l in range(64) if self.vcc[l])
        self.s[106], self.s[107] = bits & 0xFFFFFFFF, bits >> 32
        self.scc = bits != 0
    def op_s_add_u32(self, d, a, b):
        r = self.sval(a) + self.sval(b); self.scc = r >> 32 != 0; self.s[d.i] = r & 0xFFFFFFFF
    def op_s_addc_u32(self, d, a, b):
        r = self.sval(a) + self.sval(b) + int(self.scc); self.scc = r >> 32 != 0; self.s[d.i] = r & 0xFFFFFFFF
    def op_s_sub_u32(self, d, a, b):
        r = self.sval(a) - self.sval(b); self.scc = r < 0; self.s[d.i] = r & 0xFFFFFFFF
    def op_s_subb_u32(self, d, a, b):
        r = self.sval(a) - self.sval(b) - int(self.scc); self.scc = r < 0; self.s[d.i] = r & 0xFFFFFFFF
    def op_s_memtime(self, d):
        self.nclk = getattr(self, "nclk", 0) + 1000
        def apply(v=self.nclk, i=d.i): self.s[i] = v & 0xFFFFFFFF; self.s[i + 1] = v >> 32
        self.lgq.append(([], apply))
    def op_s_memrealtime(self, d):
        self.nrt = getattr(self, "nrt", 0) + 40
        def apply(v=self.nrt, i=d.i): self.s[i] = v & 0xFFFFFFFF; self.s[i + 1] = v >> 32
        self.lgq.append(([], apply))
    def op_s_lshr_b32(self, d, a, b): self.s[d.i] = self.sval(a) >> (self.sval(b) & 31)
    def op_s_lshl_b32(self, d, a, b): self.s[d.i] = (self.sval(a) << (self.sval(b) & 31)) & 0xFFFFFFFF
    def op_s_and_b32(self, d, a, b): self.s[d.i] = self.sval(a) & self.sval(b)
    def op_s_or_b32(self, d, a, b): self.s[d.i] = self.sval(a) | self.sval(b)
    def op_s_xor_b32(self, d, a, b): self.s[d.i] = self.sval(a) ^ self.sval(b)
    def op_v_min3_u32(self, d, a, b, c): self.wr32(d, np.minimum(np.minimum(self.rd32(a), self.rd32(b)), self.rd32(c)))
    def op_s_min_u32(self, d, a, b):
        x, y = self.sval(a), self.sval(b); self.scc = x < y; self.s[d.i] = min(x, y)
    def op_s_cmp_lt_u32(self, a, b): self.scc = self.sval(a) < self.sval(b)
    def op_s_cbranch_scc1(self, lab): return lab if self.scc else None
    def op_s_cbranch_scc0(self, lab): return None if self.scc else lab
    def op_s_branch(self, lab): return lab
    def op_s_cmp_lg_u32(self, a, b): self.scc = self.sval(a) != self.sval(b)
    def op_s_cmp_eq_u32(self, a, b): self.scc = self.sval(a) == self.sval(b)
    def op_s_cmp_ge_u32(self, a, b): self.scc = self.sval(a) >= self.sval(b)
    def op_s_cselect_b32(self, d, a, b):
        self.s[d.i] = (self.sval(a) if self.scc else self.sval(b)) & 0xFFFFFFFF
        if d.i in (106, 107): self._vcc_from_s()
    def op_s_cmp_gt_i32(self, a, b):
        x, y = self.sval(a), self.sval(b)
        x = x - (1 << 32) if x >> 31 else x; y = y - (1 << 32) if y >> 31 else y
        self.scc = x > y
    def op_s_mul_i32(self, d, a, b): self.s[d.i] = (self.sval(a) * self.sval(b)) & 0xFFFFFFFF
    def op_s_nop(self, n): pass
    def op_s_setprio(self, n): pass
    def op_s_sleep(self, n): pass

    def op_s_waitcnt(self, vm=None, lgkm=None):
        if vm is not None: self.retire(self.vmq, vm)
        if lgkm is not None: self.retire(self.lgq, lgkm)

    # ---- VALU 32-bit
    def op_v_mov_b32(self, d, a): self.wr32(d, self.rd32(a))
    def op_v_xor_b32(self, d, a, b): self.wr32(d, self.rd32(a) ^ self.rd32(b))
    def op_v_and_b32(self, d, a, b): self.wr32(d, self.rd32(a) & self.rd32(b))
    def op_v_or_b32(self, d, a, b): self.wr32(d, self.rd32(a) | self.rd32(b))
    def op_v_add_u32(self, d, a, b): self.wr32(d, self.rd32(a) + self.rd32(b))
    def op_v_sub_u32(self, d, a, b): self.wr32(d, self.rd32(a) - self.rd32(b))
    def op_v_subrev_u32(self, d, a, b): self.wr32(d, self.rd32(b) - self.rd32(a))
    def op_v_lshlrev_b32(self, d, a, b): self.wr32(d, self.rd32(b) << (self.rd32(a) & 31))
    def op_v_lshrrev_b32(self, d, a, b): self.wr32(d, self.rd32(b) >> (self.rd32(a) & 31))
    def op_v_lshl_add_u32(self, d, a, b, c): self.wr32(d, (self.rd32(a) << (self.rd32(b) & 31)) + self.rd32(c))
    def op_v_add3_u32(self, d, a, b, c): self.wr32(d, self.rd32(a) + self.rd32(b) + self.rd32(c))
    def op_v_xad_u32(self, d, a, b, c): self.wr32(d, (self.rd32(a) ^ self.rd32(b)) + self.rd32(c))

    def op_v_ashrrev_i32(self, d, a, b):
        self.wr32(d, (self.rd32(b).view(np.int32) >> (self.rd32(a) & 31).astype(np.int32)).view(np.uint32))
    def op_v_lshl_or_b32(self, d, a, b, c): self.wr32(d, (self.rd32(a) << (self.rd32(b) & 31)) | self.rd32(c))
    def op_v_cndmask_b32(self, d, a, b, m):
        """VOP3 form: lane mask in an SGPR pair (set by scalar moves: wave-uniform selects)"""
        bits = int(self.s[m.i]) | (int(self.s[m.i + 1]) << 32)
        lanes = np.array([(bits >> l) & 1 for l in range(64)], bool)
        self.wr32(d, np.where(lanes, self.rd32(b), self.rd32(a)))

    def op_v_alignbit_b32(self, d, hi, lo, sh):
        x = (self.rd32(hi).astype(np.uint64) << np.uint64(32)) | self.rd32(lo).astype(np.uint64)
        self.wr32(d, (x >> (self.rd32(sh).astype(np.uint64) & np.uint64(31))) & np.uint64(0xFFFFFFFF))
    def op_v_mul_lo_u32(self, d, a, b):
        self.wr32(d, (self.rd32(a).astype(np.uint64) * self.rd32(b).astype(np.uint64)) & np.uint64(0xFFFFFFFF))
    def op_v_accvgpr_write_b32(self, d, a): self.acc[d.i] = self.rd32(a)
    def op_v_accvgpr_read_b32(self, d, a): self.wr32(d, self.acc[a.i].copy())

    def op_v_bfe_u32(self, d, a, off, w):
        o, wd = int(self.rd32(off)[0]) & 31, int(self.rd32(w)[0]) & 31
        self.wr32(d, (self.rd32(a) >> o) & np.uint32((1 << wd) - 1))

    def op_v_bfe_i32(self, d, a, off, w):
        o, wd = int(self.rd32(off)[0]) & 31, int(self.rd32(w)[0]) & 31
        x = ((self.rd32(a) >> o) & np.uint32((1 << wd) - 1)).astype(np.int64)
        x = np.where(x >= (1 << (wd - 1)), x - (1 << wd), x)
        self.wr32(d, x.astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF))

    # carry operands: "vcc" or an SGPR pair (VOP3 form); each is its own 64-lane mask
    def _cget(self, c):
        if isinstance(c, str): return self.vcc
        return self.smask.get(c.i, np.zeros(64, bool))

    def _cset(self, c, m):
        if isinstance(c, str): self.vcc = m
        else:
            if not hasattr(self, "smask"): self.smask = {}
            self.smask[c.i] = m

    def op_v_add_co_u32(self, d, co, a, b):
        r = self.rd32(a).astype(np.uint64) + self.rd32(b).astype(np.uint64)
        self.wr32(d, r & np.uint64(0xFFFFFFFF)); self._cset(co, (r >> np.uint64(32)) != 0)

    def op_v_addc_co_u32(self, d, co, a, b, ci):
        r = self.rd32(a).astype(np.uint64) + self.rd32(b).astype(np.uint64) + self._cget(ci).astype(np.uint64)
        self.wr32(d, r & np.uint64(0xFFFFFFFF)); self._cset(co, (r >> np.uint64(32)) != 0)

    def op_v_sub_co_u32(self, d, co, a, b):
        x, y = self.rd32(a).astype(np.int64), self.rd32(b).astype(np.int64)
        r = x - y
        self.wr32(d, r.astype(np.uint64) & np.uint64(0xFFFFFFFF)); self._cset(co, r < 0)

    def op_v_subb_co_u32(self, d, co, a, b, ci):
        x, y = self.rd32(a).astype(np.int64), self.rd32(b).astype(np.int64)
        r = x - y - self._cget(ci).astype(np.int64)
        self.wr32(d, r.astype(np.uint64) & np.uint64(0xFFFFFFFF)); self._cset(co, r < 0)

    def op_v_lshl_add_u64(self, d, a, sh, b):
        """64-bit (a << sh) + b on register pairs (gfx940+): one 4-cycle instruction where v_add_co_u32 + v_addc_co_u32 are two"""
        def rd(x):
            assert isinstance(x, V) and x.n == 2, x
            if self.pend_v[x.i] or self.pend_v[x.i + 1]: raise Poison("read of v[%d:%d] before its load was waited for" % (x.i, x.i + 1))
            return self.v[x.i].astype(np.uint64) | (self.v[x.i + 1].astype(np.uint64) << np.uint64(32))
        k = int(self.rd32(sh)[0]) & 63
        assert k <= 4, "shift amounts above 4 are not defined for v_lshl_add_u64"
        r = (rd(a) << np.uint64(k)) + rd(b)
        assert isinstance(d, V) and d.n == 2
        if self.pend_v[d.i] or self.pend_v[d.i + 1]: raise Poison("write to v[%d:%d] while a load into it is in flight" % (d.i, d.i + 1))
        self.v[d.i] = (r & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        self.v[d.i + 1] = (r >> np.uint64(32)).astype(np.uint32)

    def op_v_mbcnt_lo_u32_b32(self, d, mask, add):
        m = int(self.rd32(mask)[0])
        cnt = np.array([bin(m & ((1 << min(l, 32)) - 1)).count("1") for l in range(64)], np.uint32)
        self.wr32(d, cnt + self.rd32(add))

    def op_v_mbcnt_hi_u32_b32(self, d, mask, add):
        m = int(self.rd32(mask)[0])
        cnt = np.array([bin(m & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(64)], np.uint32)
        self.wr32(d, cnt + self.rd32(add))

    def op_v_readfirstlane_b32(self, d, a):
        # gfx940+: 1 wait state between a VALU write of a VGPR and a v_readfirstlane / v_readlane that reads it (the hardware
        # would deliver the OLD value); found in round 5 on the GPU, refused here since
        if isinstance(a, V) and a.i in self.valu_recent[0]:
            raise Poison("v%d read by v_readfirstlane_b32 in the instruction after a VALU wrote it (1 wait state needed)" % a.i)
        self.s[d.i] = self.rd32(a)[0]

    # ---- cross-lane (gfx950): half / row swaps between two registers, DPP row rotation
    def _wait_states(self, *regs):
        """2 wait states between a VALU write of a VGPR and a v_permlane*_swap / DPP instruction that reads it: the generator
        pads with s_nop (nothing inside an asm string is padded by the assembler); the emulator refuses a stream that does not"""
        for x in regs:
            if x.i in self.valu_recent[0] or x.i in self.valu_recent[1]:
                raise Poison("v%d read by a cross-lane instruction less than 2 wait states after a VALU wrote it" % x.i)

    def op_v_permlane32_swap_b32(self, d, a):
        """lanes 32..63 of the first operand <-> lanes 0..31 of the second"""
        self._wait_states(d, a)
        x, y = self.rd32(d), self.rd32(a)
        nx, ny = x.copy(), y.copy()
        nx[32:], ny[:32] = y[:32], x[32:]
        self.wr32(d, nx); self.wr32(a, ny)

    def op_v_permlane16_swap_b32(self, d, a):
        """odd rows (16 lanes each) of the first operand <-> even rows of the second"""
        self._wait_states(d, a)
        x, y = self.rd32(d), self.rd32(a)
        nx, ny = x.copy(), y.copy()
        for row in (0, 2):
            nx[16 * row + 16:16 * row + 32], ny[16 * row:16 * row + 16] = y[16 * row:16 * row + 16], x[16 * row + 16:16 * row + 32]
        self.wr32(d, nx); self.wr32(a, ny)

    @staticmethod
    def _dpp_source_lanes(dpp):
        """lane each lane reads src0 from under DPP control string `dpp` (-1: outside the row, reads 0 with bound_ctrl:0).
        Modelled: quad_perm:[a,b,c,d], row_shl:n (lane i <- i + n), row_shr:n (lane i <- i - n), row_ror:n (lane i <- (i - n) mod 16
        within its row of 16); full row / bank masks; the shifts only with bound_ctrl:0 (every lane executes)."""
        import re
        parts = dpp.split()
        assert "row_mask:0xf" in parts and "bank_mask:0xf" in parts, dpp
        ctrl = [x for x in parts if not x.startswith(("row_mask", "bank_mask", "bound_ctrl"))]
        assert len(ctrl) == 1, dpp
        c = ctrl[0]
        src = np.zeros(64, np.int64)
        m = re.match(r"quad_perm:\[(\d),(\d),(\d),(\d)\]$", c)
        for l in range(64):
            row, i = l & 48, l & 15
            if m: src[l] = (l & ~3) | int(m.group(1 + (l & 3)))
            elif c.startswith("row_ror:"): src[l] = row | ((i - int(c[8:])) & 15)
            elif c.startswith("row_shr:"): src[l] = row | (i - int(c[8:])) if i - int(c[8:]) >= 0 else -1
            elif c.startswith("row_shl:"): src[l] = row | (i + int(c[8:])) if i + int(c[8:]) <= 15 else -1
            else: raise NotImplementedError(dpp)
        if (src < 0).any(): assert "bound_ctrl:0" in parts, "a DPP shift without bound_ctrl:0 disables the lanes that read outside their row"
        return src

    def op_v_cndmask_b32_dpp(self, d, a, b, m, dpp=""):
        """VOP2 with DPP on src0: d = vcc ? b : dpp(a)"""
        assert m == "vcc", m
        self._wait_states(a)
        src = self._dpp_source_lanes(dpp)
        x = self.rd32(a)
        moved = np.where(src >= 0, x[np.maximum(src, 0)], 0).astype(np.uint32)
        self.wr32(d, np.where(self.vcc, self.rd32(b), moved))

    # ---- VALU 64-bit float
    def op_v_mov_b64(self, d, a):
        self.wr32(V(d.i), self.rd32(V(a.i))); self.wr32(V(d.i + 1), self.rd32(V(a.i + 1)))
    def op_v_add_f64(self, d, a, b): self.wr64(d, self.rd64(a) + self.rd64(b))
    def op_v_mul_f64(self, d, a, b): self.wr64(d, self.rd64(a) * self.rd64(b))
    def op_v_fma_f64(self, d, a, b, c): self.wr64(d, self.fma(self.rd64(a), self.rd64(b), self.rd64(c)))
    def op_v_rndne_f64(self, d, a): self.wr64(d, np.rint(self.rd64(a)))
    def op_v_cvt_f64_i32(self, d, a): self.wr64(d, self.rd32(a).view(np.int32).astype(np.float64))
    def op_v_cvt_f64_u32(self, d, a): self.wr64(d, self.rd32(a).astype(np.float64))

    # ---- LDS (lgkmcnt, in order)
    def _lds_addr(self, a, offset): return (self.rd32(a).astype(np.int64) + offset)

    def _ds_read(self, d, a, offset, nd):
        addr = self._lds_addr(a, offset)
        assert (addr % (4 * min(nd, 2)) == 0).all(), "misaligned LDS read"
        if nd == 4: assert (addr % 16 == 0).all(), "ds_read_b128 must be 16-byte aligned"
        assert addr.min() >= 0 and addr.max() + 4 * nd <= self.lds.size, "LDS read out of range: %d..%d" % (addr.min(), addr.max())
        regs = list(range(d.i, d.i + nd))
        for r in regs:
            if self.pend_v[r]: raise Poison("load into v%d while an older load into it is in flight" % r)
            self.pend_v[r] = True

        def apply(addr=addr, base=d.i, nd=nd):
            w = self.lds.view(np.uint32)
            for k in range(nd): self.v[base + k] = w[(addr >> 2) + k]
        self.lgq.append((regs, apply))

    def op_ds_read_b32(self, d, a, offset=0): self._ds_read(d, a, offset, 1)
    def op_ds_read_b64(self, d, a, offset=0): self._ds_read(d, a, offset, 2)
    def op_ds_read_b128(self, d, a, offset=0): self._ds_read(d, a, offset, 4)

    def _ds_write(self, a, src, offset, nd):
        addr = self._lds_addr(a, offset)
        if nd == 4: assert (addr % 16 == 0).all(), "ds_write_b128 must be 16-byte aligned"
        assert (addr % 8 == 0).all()
        assert addr.min() >= 0 and addr.max() + 4 * nd <= self.lds.size, "LDS write out of range"
        vals = [self.rd32(V(src.i + k)) for k in range(nd)]           # data is read at issue

        def apply(addr=addr, vals=vals):
            w = self.lds.view(np.uint32)
            for k, x in enumerate(vals): w[(addr >> 2) + k] = x
        self.lgq.append(([], apply))

    def op_ds_write_b32(self, a, src, offset=0):
        addr = self._lds_addr(a, offset)
        assert (addr % 4 == 0).all() and addr.min() >= 0 and addr.max() + 4 <= self.lds.size
        val = self.rd32(src)
        def apply(addr=addr, val=val): self.lds.view(np.uint32)[addr >> 2] = val
        self.lgq.append(([], apply))
    def op_ds_write_b64(self, a, src, offset=0): self._ds_write(a, src, offset, 2)
    def op_ds_write_b128(self, a, src, offset=0): self._ds_write(a, src, offset, 4)

    # ---- global (vmcnt, in order); saddr form: base in an SGPR pair + 32-bit VGPR offset + immediate
    def _gaddr(self, voff, sbase, offset):
        base = int(self.s[sbase.i]) | (int(self.s[sbase.i + 1]) << 32)
        return base + self.rd32(voff).astype(np.int64) + offset

    def _gload(self, d, voff, sbase, offset, nd):
        assert -4096 <= offset <= 4095, "global immediate offset is 13-bit signed"
        addr = self._gaddr(voff, sbase, offset)
        regs = list(range(d.i, d.i + nd))
        for r in regs:
            if self.pend_v[r]: raise Poison("load into v%d while an older load into it is in flight" % r)
            self.pend_v[r] = True

        def apply(addr=addr, base=d.i, nd=nd):
            for l in range(64):
                arr, o = self.gmem(int(addr[l]), 4 * nd)
                self.v[base:base + nd, l] = arr[o:o + 4 * nd].view(np.uint32)
        self.vmq.append((regs, apply))

    def op_global_load_lds_dwordx4(self, voff, sbase, offset=0):
        """LDS-DMA: lane l's 16 bytes go to LDS[M0 + 16 l]; counted by vmcnt; no destination registers"""
        assert offset == 0, "keep LDS-DMA loads free of an instruction offset (it applies to the LDS address too)"
        addr = self._gaddr(voff, sbase, offset)
        m0 = int(self.s[124])
        assert m0 % 16 == 0 and 0 <= m0 and m0 + 1024 <= self.lds.size, "LDS-DMA destination out of range: %d" % m0

        def apply(addr=addr, m0=m0):
            for l in range(64):
                arr, o = self.gmem(int(addr[l]), 16)
                self.lds[m0 + 16 * l: m0 + 16 * l + 16] = arr[o:o + 16]
        self.vmq.append(([], apply))

    def op_global_load_dwordx4(self, d, voff, sbase, offset=0, cache=None): self._gload(d, voff, sbase, offset, 4)
    def op_global_load_dwordx2(self, d, voff, sbase, offset=0): self._gload(d, voff, sbase, offset, 2)

    def _gstore(self, voff, src, sbase, offset, nd):
        assert -4096 <= offset <= 4095
        addr = self._gaddr(voff, sbase, offset)
        vals = [self.rd32(V(src.i + k)) for k in range(nd)]

        def apply(addr=addr, vals=vals):
            for l in range(64):
                arr, o = self.gmem(int(addr[l]), 4 * nd)
                arr[o:o + 4 * nd].view(np.uint32)[:] = [v[l] for v in vals]
        self.vmq.append(([], apply))

    def op_global_store_dword(self, voff, src, sbase, offset=0, cache=None): self._gstore(voff, src, sbase, offset, 1)
    def op_global_store_dwordx2(self, voff, src, sbase, offset=0, cache=None): self._gstore(voff, src, sbase, offset, 2)
    def op_global_store_dwordx4(self, voff, src, sbase, offset=0, cache=None): self._gstore(voff, src, sbase, offset, 4)     # cache: "nt" etc. (no effect on values)


def run_group(emus, prog):
    """several waves of one workgroup (they share `lds`): each runs to its next s_barrier, then all go on"""
    gens = [e.run_gen(prog) for e in emus]
    alive = [True] * len(gens)
    while any(alive):
        states = []
        for k, g in enumerate(gens):
            if not alive[k]: continue
            try:
                states.append(next(g))
            except StopIteration:
                alive[k] = False
                states.append("end")
        if len(set(states)) != 1: raise RuntimeError("waves disagree at a barrier: %s" % states)
