#!/usr/bin/env python3
"""tools/roofline_inputs.py <tag>: turn the counter CSVs of one GPU-box session (tools/r5_session*.sh <tag>: gpurun_out/prof_<tag>/
and gpurun_out/<tag>_pmc_<kernel>/) into profiles/roofline_inputs.json -- the only place bench.py takes counter-derived figures
from -- and copy the rows it read into profiles/ as <tag>_*.csv.

Two sections:
  entries  the dominant kernel's figures behind `roofline.traffic` / `vector_pipe_busy_pmc` (k_blind_rotate5), tied to the sha256 of
           its generated instruction stream (csrc/tx_br_asm_fair.inc);
  kernels  (round 5) the same figures for each of the three kernels that make up a step -- k_blind_rotate5,
           k_trace_scheme_switch4, k_lut_eval -- as they ran INSIDE the pipeline (`bench.py --steps 1 --warmup 1`), each tied to
           the sha256 over the source files the kernel is built from: behind `roofline.kernels`.
bench.py reports a figure only while the hash is that of the files in the tree.  Run in the repo, after gpurun has merged
gpurun_out/ back, BEFORE the kernel sources are touched again."""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "temp-fhe-transciphering_amd", "csrc")
KERNEL = "k_blind_rotate5"
# kernel -> (substring the PMC session was filtered with, waves per SIMD, source files the kernel is built from)
KERNELS = {
    "k_blind_rotate5": ("blind_rotate5", 2, ["tx_br_asm_fair.inc"]),
    "k_trace_scheme_switch4": ("trace_scheme_switch4", 1, ["tx_trace_asm.inc", "tx_ss_asm.inc"]),
    "k_lut_eval": ("lut_eval", 1, ["tx_kernels.hpp", "tx_device.hpp"]),
}


def rows(path, kernel=KERNEL):
    return [r for r in csv.DictReader(open(path)) if kernel + "(" in r["Kernel_Name"]]


def one(pattern, optional=False):
    hits = glob.glob(pattern) + glob.glob(pattern.replace("/*counter", "/*/*counter"))
    if len(hits) != 1:
        if optional and not hits:
            return None
        raise SystemExit("expected one file for %s, found %s" % (pattern, hits))
    return hits[0]


def keep(src, name, match="blind_rotate"):
    """only the matching kernels' rows go to profiles/ (the full CSV of a bench command is megabytes of elementwise launches)"""
    dst = os.path.join(ROOT, "profiles", name)
    rd = csv.DictReader(open(src))
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rd.fieldnames)
        w.writeheader()
        for r in rd:
            if match in r["Kernel_Name"]:
                w.writerow(r)
    return "profiles/" + name


def sha_files(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def sums(path, kernel):
    a = {}
    for r in rows(path, kernel):
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return a


def main():
    tag = sys.argv[1]
    prof = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    sha = sha_files(["tx_br_asm_fair.inc"])
    entries = {}

    def per_launch(path, kernel=KERNEL):
        rs = rows(path, kernel)
        return sum(float(r["Counter_Value"]) for r in rs) / len(rs), len(rs)

    # traffic of the default command's own launch mix (FETCH_SIZE / WRITE_SIZE in KiB, separate passes)
    f_b, w_b = one(prof + "/fetch_bench/*counter_collection.csv"), one(prof + "/write_bench/*counter_collection.csv")
    (fk, nf), (wk, nw) = per_launch(f_b), per_launch(w_b)
    # bench.py --steps 1 --warmup 1: the gate step + one timed step, nine k_blind_rotate5 launches each (eight of 8192, one of 2048)
    assert nf == nw and nf % 9 == 0, (nf, nw)
    src = keep(f_b, tag + "_pmc_fetch_size_bench_command.csv", "k_") + " + " + keep(w_b, tag + "_pmc_write_size_bench_command.csv", "k_")
    entries["traffic_default_mix"] = {
        "kernel": KERNEL, "avg_units_per_launch": (8 * 8192 + 2048) / 9.0, "launches_measured": nf,
        "fetch_size_kib": fk, "write_size_kib": wk, "bytes_per_launch": (fk * 2 + wk) * 1024,
        "formula": "FETCH_SIZE x 2 + WRITE_SIZE (KiB; gfx950 counts half of a 16 B/lane streaming read, MI355X_MICROARCH.md HBM section), "
                   "average over the k_blind_rotate5 launches of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs --no-stage-times`",
        "source": src, "stream_sha256": sha}
    f_1, w_1 = one(prof + "/fetch/*counter_collection.csv", True), one(prof + "/write/*counter_collection.csv", True)
    if f_1 and w_1:
        (fk1, nf1), (wk1, nw1) = per_launch(f_1), per_launch(w_1)
        src1 = keep(f_1, tag + "_pmc_fetch_size_blind_rotate5_8192_units.csv") + " + " + keep(w_1, tag + "_pmc_write_size_blind_rotate5_8192_units.csv")
        entries["traffic_8192_units"] = {
            "kernel": KERNEL, "avg_units_per_launch": 8192.0, "launches_measured": nf1, "fetch_size_kib": fk1, "write_size_kib": wk1,
            "bytes_per_launch": (fk1 * 2 + wk1) * 1024, "formula": "FETCH_SIZE x 2 + WRITE_SIZE (KiB), launches of 8192 units (tools/br_only.py 768 8192 1)",
            "source": src1, "stream_sha256": sha}

    # ---- per kernel, inside the pipeline: SQ + LDS passes (tools/r3_pmc_pipeline.sh) and the FETCH / WRITE passes above
    kernels = {}
    for kernel, (pat, wps, files) in KERNELS.items():
        d = os.path.join(ROOT, "gpurun_out", "%s_pmc_%s" % (tag, pat))
        sq, lds = one(d + "/sq/*counter_collection.csv", True), one(d + "/lds/*counter_collection.csv", True)
        if not sq or not lds:
            print("no SQ / LDS pass for", kernel, "in", d)
            continue
        a_sq, a_lds = sums(sq, kernel), sums(lds, kernel)            # two passes: each has its own SQ_WAVE_CYCLES
        short = kernel[2:]
        srcs = keep(sq, "%s_pmc_sq_%s_in_pipeline.csv" % (tag, short), kernel + "(") + " + " + keep(lds, "%s_pmc_lds_%s_in_pipeline.csv" % (tag, short), kernel + "(")
        frac = {k: v / a_sq["SQ_WAVE_CYCLES"] for k, v in sorted(a_sq.items()) if k.startswith(("SQ_ACTIVE", "SQ_WAIT"))}
        frac.update({k: v / a_lds["SQ_WAVE_CYCLES"] for k, v in sorted(a_lds.items()) if k.startswith(("SQ_ACTIVE", "SQ_LDS"))})
        (fk, nf), (wk, nw) = per_launch(f_b, kernel), per_launch(w_b, kernel)
        n_sq = len([r for r in rows(sq, kernel) if r["Counter_Name"] == "SQ_WAVE_CYCLES"])
        kernels[kernel] = {
            "waves_per_simd": wps, "vector_pipe_busy": wps * a_sq["SQ_ACTIVE_INST_VALU"] / a_sq["SQ_WAVE_CYCLES"],
            "per_wave_fraction_of_wave_cycles": frac, "counters_sq_pass": a_sq, "counters_lds_pass": a_lds,
            "launches_measured": n_sq, "launches_per_step": nf // 2,
            "fetch_size_kib": fk, "write_size_kib": wk, "bytes_per_launch": (fk * 2 + wk) * 1024,
            "formula": "vector_pipe_busy = waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; bytes_per_launch = (FETCH_SIZE x 2 + WRITE_SIZE) KiB, "
                       "both over this kernel's launches of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs --no-stage-times` "
                       "(separate rocprofv3 --pmc passes)",
            "source": srcs + " + " + src, "stream_files": files, "stream_sha256": sha_files(files)}
        if kernel == KERNEL:
            entries["vector_pipe_busy"] = {
                "kernel": KERNEL, "avg_units_per_launch": (8 * 8192 + 2048) / 9.0, "value": kernels[kernel]["vector_pipe_busy"],
                "per_wave_fraction_of_wave_cycles": frac, "counters_sq_pass": a_sq, "counters_lds_pass": a_lds,
                "formula": "2 waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, summed over the k_blind_rotate5 launches of `bench.py --steps 1 --warmup 1`",
                "source": srcs, "stream_sha256": sha}
    old = {}
    try:
        old = json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
    except (OSError, ValueError):
        pass
    if "vector_pipe_busy" not in entries and "vector_pipe_busy" in old.get("entries", {}):
        entries["vector_pipe_busy"] = old["entries"]["vector_pipe_busy"]       # (kept while its stream hash still matches; bench.py checks)
    doc = {"_about": "Counter-derived inputs of bench.py's roofline object. `entries`: the dominant kernel (k_blind_rotate5), each entry tied to the "
                     "sha256 of temp-fhe-transciphering_amd/csrc/tx_br_asm_fair.inc; `kernels`: the three kernels of a step as they ran inside the "
                     "pipeline, each tied to the sha256 over its `stream_files` under csrc/. bench.py prints null for a figure whose hash is not "
                     "that of the files in the tree, or (entries) whose launch mix (avg_units_per_launch) is not the one it measured. Written by "
                     "tools/roofline_inputs.py from the rocprofv3 --pmc passes of a GPU-box session.",
           "session": tag, "entries": entries, "kernels": kernels}
    json.dump(doc, open(os.path.join(ROOT, "profiles", "roofline_inputs.json"), "w"), indent=1)
    for k, e in entries.items():
        print(k, {x: e[x] for x in e if x in ("bytes_per_launch", "value", "fetch_size_kib", "write_size_kib", "launches_measured")})
    for k, e in kernels.items():
        print(k, {x: (round(e[x], 4) if isinstance(e[x], float) else e[x]) for x in ("vector_pipe_busy", "bytes_per_launch", "launches_per_step", "launches_measured")},
              {x: round(v, 4) for x, v in e["per_wave_fraction_of_wave_cycles"].items()})
    # the kernel-stats summary of the same session
    for f in glob.glob(prof + "/stats/*kernel_stats.csv") + glob.glob(prof + "/stats/*/*kernel_stats.csv"):
        shutil.copy(f, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    for f, n in ((os.path.join(prof, "bench_line_profiled.json"), tag + "_bench_line_profiled.json"),):
        if os.path.exists(f):
            shutil.copy(f, os.path.join(ROOT, "profiles", n))


if __name__ == "__main__":
    main()
