#!/bin/bash
# GPU box: HBM-side traffic of one self-attention launch (Wan-14B 720p shape), three separate
# counter passes over the native harness, written to gpurun_out/traffic/attn_traffic.json (copy it to profiles/<round>_attn_traffic.json).
set -o pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/traffic; mkdir -p $out
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  timeout -s KILL 90 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/$tag -o p -- ./build/kernel_harness selfattn 75600 40 1 > $out/$tag.log 2>&1 || { echo "pass $tag failed"; exit 1; }
done
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def per_call(tag, name):
    # One attention call is a main launch plus a tail launch (two kernel names): sum over all
    # launches, divided by the number of calls (= launches of any one name).
    f = glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "attn_fwd" in r["Kernel_Name"] and r["Counter_Name"] == name]
    names = sorted({r["Kernel_Name"] for r in rows})
    calls = sum(r["Kernel_Name"] == names[0] for r in rows)
    return sum(float(r["Counter_Value"]) for r in rows) / calls, calls, names
fetch, n, names = per_call("FETCH_SIZE", "FETCH_SIZE")
write, _, _ = per_call("WRITE_SIZE", "WRITE_SIZE")
hit, _, _ = per_call("TCC_HIT_sum_TCC_MISS_sum", "TCC_HIT_sum")
miss, _, _ = per_call("TCC_HIT_sum_TCC_MISS_sum", "TCC_MISS_sum")
S, H, d = 75600, 40, 128
rec = {
    "kernel": " + ".join(names) + " (one self-attention call)",
    "workload": {"tokens": S, "heads": H, "head_dim": d},
    "collected": f"tools/measure_attn_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate runs of build/kernel_harness selfattn 75600 40 1 ({n} calls each)",
    "FETCH_SIZE_KB_per_call": fetch, "WRITE_SIZE_KB_per_call": write,
    "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: read bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 is exact",
    "read_bytes_per_call": fetch * 1024 * 2, "write_bytes_per_call": write * 1024,
    "traffic_bytes_per_call": fetch * 1024 * 2 + write * 1024,
    "algorithmic_bytes_per_call": 4 * S * H * d * 2,
    "l2_hit_rate": hit / (hit + miss),
}
json.dump(rec, open(out + "/attn_traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
