#!/bin/bash
# two PMC passes (SQ activity; LDS) over one of the kernel-only tools; usage:
#   tools/pmc_kernel.sh <outdir> <kernel name substring> <tool.py> [tool args]      e.g.  ... prof blind_rotate tools/br_only.py 768 8192 1
# counters are collected with --pmc only (no trace domains), the program itself after `--`
set -e
out=$1; pat=$2; tool=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM -d $R/$out/sq -o sq --output-format csv -- python3 $R/$tool "$@" > $R/$out/sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $R/$out/lds -o lds --output-format csv -- python3 $R/$tool "$@" > $R/$out/lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/$out/*/*counter_collection.csv") + glob.glob("$R/$out/*/*/*counter_collection.csv"):
    acc = collections.OrderedDict(); meta = None
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"]); acc[k] = acc.get(k, 0) + float(r["Counter_Value"])
            meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["Workgroup_Size"], r["Grid_Size"])
    print(f.split("/")[-1], meta)
    for k, v in acc.items(): print("  ", k[0], k[1], "%.4g" % v)
PY
