#!/bin/bash
# PMC passes for the trace / seed kernels (run on the GPU box through gpurun).  Each pass is its own
# rocprofv3 run with --pmc only (never combined with tracing options).
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 1 --warmup 0 --spp-per-step 4 --no-counters --no-cpu-baseline ${BENCH_ARGS:-}"
PASSES=${PASSES:-7}
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" ; do
  i=$((i+1))
  if [ $i -gt $PASSES ]; then break; fi
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$?" >> "$OUT/status.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE", "TCC_HIT_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "SQ_ACTIVE_INST_VALU", "WRITE_SIZE", "TA_BUSY_avr"):
            cnt[(k, row["Counter_Name"])] += 1
with open(out + "/summary.txt", "w") as o:
    for k, d in agg.items():
        if "trace" not in k and "seed" not in k:
            continue
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write("   %-36s %.6g\n" % (c, v))
        o.write("   dispatch counts: %s\n" % {c: n for (kk, c), n in cnt.items() if kk == k})
print(open(out + "/summary.txt").read())
PY
