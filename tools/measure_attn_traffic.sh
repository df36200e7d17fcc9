#!/bin/bash
# GPU box: HBM-side traffic of one self-attention launch (Wan-14B 720p shape), three separate
# counter passes over the native harness, written to profiles/r01_attn_traffic.json.
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
def mean(tag, name):
    f = glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn_fwd" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v), sorted({r["Kernel_Name"] for r in csv.DictReader(open(f)) if "attn_fwd" in r["Kernel_Name"]})
fetch, n, names = mean("FETCH_SIZE", "FETCH_SIZE")
write, _, _ = mean("WRITE_SIZE", "WRITE_SIZE")
hit, _, _ = mean("TCC_HIT_sum_TCC_MISS_sum", "TCC_HIT_sum")
miss, _, _ = mean("TCC_HIT_sum_TCC_MISS_sum", "TCC_MISS_sum")
S, H, d = 75600, 40, 128
rec = {
    "kernel": names[0] + " (self-attention)",
    "workload": {"tokens": S, "heads": H, "head_dim": d},
    "collected": f"tools/measure_attn_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate runs of build/kernel_harness selfattn 75600 40 1 ({n} launches each)",
    "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
    "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: read bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 is exact",
    "read_bytes_per_launch": fetch * 1024 * 2, "write_bytes_per_launch": write * 1024,
    "traffic_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "algorithmic_bytes_per_launch": 4 * S * H * d * 2,
    "l2_hit_rate": hit / (hit + miss),
}
json.dump(rec, open("profiles/r01_attn_traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
