#!/bin/bash
# round 4, call 37: address-translation counters of the headline launch with 2.7 GB and with 5.3 GB of rows (one --pmc pass per counter)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c37
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cp $ROOT/profiles/scratch/output_size_pmc.py /tmp/osp.py
for C in TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_LFIFO_NOT_RES GRBM_UTCL2_BUSY TCP_UTCL1_STALL_MULTI_MISS_sum; do
  for K in 500 1000; do
    ( cd $ROOT && timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/${C}_$K -o pmc -- python3 profiles/scratch/output_size_pmc.py $K > $OUT/${C}_$K.log 2>&1 ) || { tail -5 $OUT/${C}_$K.log; echo "pass $C $K failed"; }
  done
done
python3 - <<PY
import csv, glob
out = "$OUT"
for C in ("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_STALL_LFIFO_NOT_RES", "GRBM_UTCL2_BUSY", "TCP_UTCL1_STALL_MULTI_MISS_sum"):
    for K in (500, 1000):
        fs = glob.glob(f"{out}/{C}_{K}/**/pmc_counter_collection.csv", recursive=True)
        if not fs:
            print(C, K, "no file"); continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "rollout_kernel" in r["Kernel_Name"]]
        byd = {}
        for r in rows:
            byd.setdefault(r["Dispatch_Id"], 0.0)
            byd[r["Dispatch_Id"]] += float(r["Counter_Value"])
        vals = [byd[k] for k in sorted(byd, key=int)][-10:]
        print(f"{C} K={K}: {len(byd)} dispatches, mean of the last {len(vals)}: {sum(vals) / max(1, len(vals)):.4g} per launch = {sum(vals) / max(1, len(vals)) / K:.4g} per env-step")
PY
