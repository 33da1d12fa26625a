#!/bin/bash
# PMC passes for the trace / seed kernels (run on the GPU box through gpurun).  Each pass is its own rocprofv3 run with --pmc
# only (never combined with tracing options).  Writes $OUT/summary.txt and $OUT/pmc_traffic.json; copy the latter to
# profiles/rNN_pmc_traffic.json — bench.py reads the newest one for roofline.traffic and roofline.issue.
#   usage: tools/prof_pmc.sh [outdir] [round-tag] [scene]      e.g. tools/prof_pmc.sh gpurun_out/pmc_r02 r02     (scene: default rtcamp6_v3_1;
#   another scene's file goes to profiles/rNN_pmc_traffic_<scene>.json and is what bench.py --scene <scene> reads)
set -u
OUT=${1:-gpurun_out/pmc}
TAG=${2:-r03}
SCENE=${3:-rtcamp6_v3_1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --scene $SCENE --steps 1 --warmup 0 --spp-per-step 4 --no-counters --no-cpu-baseline ${BENCH_ARGS:-}"
PASSES=${PASSES:-7}
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
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
python3 - "$OUT" "$TAG" "$SCENE" <<'PY'
import csv, glob, json, sys, collections
out, tag, scene = sys.argv[1], sys.argv[2], sys.argv[3]
suffix = "" if scene == "rtcamp6_v3_1" else "_" + scene
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k][row["Counter_Name"]].add(row.get("Dispatch_Id", "?"))
PATHS = 1920 * 1080 * 4 * 4   # one launch of either kernel under this command
kernels = {}
with open(out + "/summary.txt", "w") as o:
    for k, d in agg.items():
        if "trace_kernel" not in k and "seed_" not in k:
            continue
        o.write(k + "\n")
        per = {}
        for c, v in sorted(d.items()):
            n = max(1, len(disp[k][c]))
            o.write("   %-36s %.6g   (%d dispatches)\n" % (c, v, n))
            per[c] = v / n
        name = "trace_kernel" if "trace_kernel" in k else ("seed_seg_kernel" if "seed_seg" in k else "seed_pc_kernel" if "seed_pc" in k else k.replace("void ", "").strip())
        e = {}
        # FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (x2)
        if "FETCH_SIZE" in per: e["fetch_bytes_per_path"] = per["FETCH_SIZE"] * 1024 * 2.0 / PATHS
        if "WRITE_SIZE" in per: e["write_bytes_per_path"] = per["WRITE_SIZE"] * 1024 / PATHS
        for c, key in (("SQ_INSTS_VALU", "valu_per_path"), ("SQ_INSTS_SALU", "salu_per_path"), ("SQ_INSTS_VMEM_RD", "vmem_per_path"), ("SQ_INSTS_LDS", "lds_per_path")):
            if c in per: e[key] = per[c] / PATHS
        if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per: e["l2_hit_rate"] = per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
        if "SQ_THREAD_CYCLES_VALU" in per and "SQ_ACTIVE_INST_VALU" in per: e["valu_lane_utilisation"] = per["SQ_THREAD_CYCLES_VALU"] / max(1.0, per["SQ_ACTIVE_INST_VALU"] * 64)
        # texture addresser busy: TA_TA_BUSY_sum counts busy cycles over all CUs; GRBM_GUI_ACTIVE = the kernel's cycles (kernels run serialised under PMC)
        if "TA_TA_BUSY_sum" in per and "GRBM_GUI_ACTIVE" in per: e["ta_busy_frac"] = per["TA_TA_BUSY_sum"] / max(1.0, 256.0 * per["GRBM_GUI_ACTIVE"] / 8.0)   # GRBM_GUI_ACTIVE comes summed over the 8 XCDs
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in per: e["l1_line_accesses_per_path"] = per["TCP_TOTAL_CACHE_ACCESSES_sum"] / PATHS
        if "TCP_TCC_READ_REQ_sum" in per: e["l1_to_l2_requests_per_path"] = per["TCP_TCC_READ_REQ_sum"] / PATHS
        if name in kernels: continue
        kernels[name] = e
sys.path.insert(0, ".")
from bench import kernel_source_sha
json.dump({"paths_per_launch": PATHS, "scene": scene, "kernels": kernels, "csrc_sha": kernel_source_sha(),
           "source": "profiles/%s_pmc_summary%s.txt (tools/prof_pmc.sh: separate rocprofv3 --pmc passes; FETCH_SIZE x2 gfx950 correction; kernels run serialised under PMC)" % (tag, suffix)},
          open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/summary.txt").read())
print(open(out + "/pmc_traffic.json").read())
PY
