#!/bin/bash
# GPU box: SQ counters of the self-attention launch for kernel $1 (w4 | x16), printed per self-attention call.
set -o pipefail
cd "$(dirname "$0")/.."
export LX2V_ATTN_KERNEL=$1
d=gpurun_out/sq_$1
mkdir -p $d
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $d -o p -- ./build/kernel_harness selfattn 75600 40 1 > $d.log 2>&1 || { echo "run failed"; tail -5 $d.log; exit 1; }
f=$(find $d -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))   # counter -> kernel symbol -> values
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_fwd" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, by in sorted(acc.items()):   # per self-attention call: the means of its launches (main + tail) summed
    print(f"{k:28s} {sum(sum(v) / len(v) for v in by.values()):.4g}")
PY
