#!/usr/bin/env python3
"""tools/roofline_inputs.py <tag>: turn the counter CSVs of one GPU-box session (tools/r4_session.sh <tag>: gpurun_out/prof_<tag>/
and gpurun_out/<tag>_pmc_blind_rotate5/) into profiles/roofline_inputs.json -- the only place bench.py takes `roofline.traffic` and
`vector_pipe_busy_pmc` from -- and copy the CSVs it read into profiles/ as <tag>_*.csv.  Every entry records the sha256 of the
instruction stream (csrc/tx_br_asm_fair.inc) of the tree the session ran: bench.py reports an entry only while that hash is the
tree's.  Run in the repo, after gpurun has merged gpurun_out/ back, BEFORE the generated stream is touched again."""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_blind_rotate5"


def rows(path, kernel=KERNEL):
    return [r for r in csv.DictReader(open(path)) if kernel + "(" in r["Kernel_Name"]]


def one(pattern):
    hits = glob.glob(pattern) + glob.glob(pattern.replace("/*counter", "/*/*counter"))
    if len(hits) != 1:
        raise SystemExit("expected one file for %s, found %s" % (pattern, hits))
    return hits[0]


def keep(src, name):
    """only this kernel's rows go to profiles/ (the full CSV of a bench command is megabytes of elementwise launches)"""
    dst = os.path.join(ROOT, "profiles", name)
    rd = csv.DictReader(open(src))
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rd.fieldnames)
        w.writeheader()
        for r in rd:
            if "blind_rotate" in r["Kernel_Name"]:
                w.writerow(r)
    return "profiles/" + name


def main():
    tag = sys.argv[1]
    prof = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    sha = hashlib.sha256(open(os.path.join(ROOT, "temp-fhe-transciphering_amd", "csrc", "tx_br_asm_fair.inc"), "rb").read()).hexdigest()
    entries = {}

    def per_launch(path):
        rs = rows(path)
        grid_units = None
        return sum(float(r["Counter_Value"]) for r in rs) / len(rs), len(rs)

    # traffic of the default command's own launch mix (FETCH_SIZE / WRITE_SIZE in KiB, separate passes)
    f_b, w_b = one(prof + "/fetch_bench/*counter_collection.csv"), one(prof + "/write_bench/*counter_collection.csv")
    (fk, nf), (wk, nw) = per_launch(f_b), per_launch(w_b)
    # bench.py --steps 1 --warmup 1: the gate step + one timed step, nine k_blind_rotate5 launches each (eight of 8192, one of 2048)
    assert nf == nw and nf % 9 == 0, (nf, nw)
    src = keep(f_b, tag + "_pmc_fetch_size_bench_command.csv") + " + " + keep(w_b, tag + "_pmc_write_size_bench_command.csv")
    entries["traffic_default_mix"] = {
        "kernel": KERNEL, "avg_units_per_launch": (8 * 8192 + 2048) / 9.0, "launches_measured": nf,
        "fetch_size_kib": fk, "write_size_kib": wk, "bytes_per_launch": (fk * 2 + wk) * 1024,
        "formula": "FETCH_SIZE x 2 + WRITE_SIZE (KiB; gfx950 counts half of a 16 B/lane streaming read, MI355X_MICROARCH.md HBM section), "
                   "average over the k_blind_rotate5 launches of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs`",
        "source": src, "stream_sha256": sha}
    f_1, w_1 = one(prof + "/fetch/*counter_collection.csv"), one(prof + "/write/*counter_collection.csv")
    (fk, nf), (wk, nw) = per_launch(f_1), per_launch(w_1)
    src = keep(f_1, tag + "_pmc_fetch_size_blind_rotate5_8192_units.csv") + " + " + keep(w_1, tag + "_pmc_write_size_blind_rotate5_8192_units.csv")
    entries["traffic_8192_units"] = {
        "kernel": KERNEL, "avg_units_per_launch": 8192.0, "launches_measured": nf, "fetch_size_kib": fk, "write_size_kib": wk,
        "bytes_per_launch": (fk * 2 + wk) * 1024, "formula": "FETCH_SIZE x 2 + WRITE_SIZE (KiB), launches of 8192 units (tools/br_only.py 768 8192 1)",
        "source": src, "stream_sha256": sha}
    # vector-pipe occupancy of the kernel as it runs in the pipeline: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave, two waves per SIMD
    sq = one(os.path.join(ROOT, "gpurun_out", tag + "_pmc_blind_rotate5", "sq", "*counter_collection.csv"))
    lds = one(os.path.join(ROOT, "gpurun_out", tag + "_pmc_blind_rotate5", "lds", "*counter_collection.csv"))
    def sums(path):
        a = {}
        for r in rows(path):
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        return a
    a_sq, a_lds = sums(sq), sums(lds)            # two passes: each has its own SQ_WAVE_CYCLES
    src = keep(sq, tag + "_pmc_sq_blind_rotate5_in_pipeline.csv") + " + " + keep(lds, tag + "_pmc_lds_blind_rotate5_in_pipeline.csv")
    per_wave = a_sq["SQ_ACTIVE_INST_VALU"] / a_sq["SQ_WAVE_CYCLES"]
    frac = {k: v / a_sq["SQ_WAVE_CYCLES"] for k, v in sorted(a_sq.items()) if k.startswith(("SQ_ACTIVE", "SQ_WAIT"))}
    frac.update({k: v / a_lds["SQ_WAVE_CYCLES"] for k, v in sorted(a_lds.items()) if k.startswith(("SQ_ACTIVE", "SQ_LDS"))})
    entries["vector_pipe_busy"] = {
        "kernel": KERNEL, "avg_units_per_launch": (8 * 8192 + 2048) / 9.0, "value": 2 * per_wave,
        "per_wave_fraction_of_wave_cycles": frac, "counters_sq_pass": a_sq, "counters_lds_pass": a_lds,
        "formula": "2 waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, summed over the k_blind_rotate5 launches of `bench.py --steps 1 --warmup 1`",
        "source": src, "stream_sha256": sha}
    doc = {"_about": "Counter-derived inputs of bench.py's roofline object. Each entry names the CSV(s) under profiles/ it was read from and "
                     "the sha256 of temp-fhe-transciphering_amd/csrc/tx_br_asm_fair.inc (the generated CMux loop of k_blind_rotate5) it was "
                     "measured on; bench.py prints null for an entry whose hash is not that of the stream in the tree, or whose launch mix "
                     "(avg_units_per_launch) is not the one it measured. Written by tools/roofline_inputs.py from the rocprofv3 --pmc passes "
                     "of tools/r4_session.sh.",
           "session": tag, "entries": entries}
    json.dump(doc, open(os.path.join(ROOT, "profiles", "roofline_inputs.json"), "w"), indent=1)
    for k, e in entries.items():
        print(k, {x: e[x] for x in e if x in ("bytes_per_launch", "value", "fetch_size_kib", "write_size_kib", "launches_measured")})
    # the kernel-stats summary of the same session
    for f in glob.glob(prof + "/stats/*kernel_stats.csv") + glob.glob(prof + "/stats/*/*kernel_stats.csv"):
        shutil.copy(f, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    for f, n in ((os.path.join(prof, "bench_line_profiled.json"), tag + "_bench_line_profiled.json"),):
        if os.path.exists(f):
            shutil.copy(f, os.path.join(ROOT, "profiles", n))


if __name__ == "__main__":
    main()
