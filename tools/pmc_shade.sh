#!/bin/bash
# GPU-box script: counters of the integrator's kernels (k_shade, k_sky, k_accumulate and the queued traversal) on
# config 3 at 16 spp.  One rocprofv3 run per counter group.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_shade; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
            "SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --spectral-spp 16 > $OUT/$name.log 2>&1 || { echo "pass $pass failed"; tail -3 $OUT/$name.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_shade", "k_sky", "k_accumulate", "k_generate", "k_trace_fast_persistent<12, true"):
            if key in k:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print("##", k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} launches={len(v):4d} sum={sum(v):.6g} mean={sum(v)/len(v):.6g}")
PY
