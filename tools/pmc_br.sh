#!/bin/bash
# two PMC passes (SQ activity; LDS) over tools/br_only.py; usage: tools/pmc_br.sh <outdir> [br_only args]
# TX_BR_VARIANT etc. are taken from the environment (exported by the caller, never through `env`)
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM -d $R/$out/sq -o sq --output-format csv -- python3 $R/tools/br_only.py "$@" > $R/$out/sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $R/$out/lds -o lds --output-format csv -- python3 $R/tools/br_only.py "$@" > $R/$out/lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/$out/*/*counter_collection.csv") + glob.glob("$R/$out/*/*/*counter_collection.csv"):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "blind_rotate" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"]); acc[k] = acc.get(k, 0) + float(r["Counter_Value"])
            meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["Workgroup_Size"], r["Grid_Size"])
    print(f.split("/")[-1], meta)
    for k, v in acc.items(): print("  ", k[0], k[1], "%.4g" % v)
PY
