#!/bin/bash
# Everything under profiles/ for one round, on the GPU box:  tools/final_profiles.sh r03
TAG=${1:-r03}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|libdrm" | tail -5 > $OUT/pytest_gpu.log
# the bench line and the kernel-trace statistics of the SAME run
rocprofv3 --kernel-trace --stats -d $OUT/prof -o ${TAG} -- python bench.py > $OUT/${TAG}_bench_full.json.log 2> $OUT/bench_full.err
DB=$(find $OUT/prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" $OUT/${TAG}_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py" > /dev/null
# unprofiled run of the same command
python bench.py > $OUT/${TAG}_bench_full_unprofiled.json.log 2>> $OUT/bench_full.err
# other BASELINE configurations on one GPU
python bench.py --scene spheres --steps 4 --spp-per-step 16 --no-cpu-baseline > $OUT/${TAG}_bench_c2_spheres.json.log 2>> $OUT/bench_full.err
python bench.py --scene rtcamp6_dodeca --width 3840 --height 2160 --steps 64 --warmup 1 --spp-per-step 4 --no-cpu-baseline > $OUT/${TAG}_bench_c5_4k_dodeca.json.log 2>> $OUT/bench_full.err
python bench.py --bvh-builder 1 --steps 16 --no-cpu-baseline > $OUT/${TAG}_bench_lbvh.json.log 2>> $OUT/bench_full.err
python bench.py --bvh-builder 2 --steps 16 --no-cpu-baseline > $OUT/${TAG}_bench_ploc.json.log 2>> $OUT/bench_full.err
# N > 1 exactly as the driver runs N = 1 (no launcher): one process, two contexts; on this 1-GPU box both share device 0
python bench.py --gpus 2 --steps 8 --no-cpu-baseline > $OUT/${TAG}_bench_gpus2_one_device.json.log 2>> $OUT/bench_full.err
python bench.py --russian-roulette 3 --steps 16 --no-cpu-baseline > $OUT/${TAG}_bench_russian_roulette_nonparity.json.log 2>> $OUT/bench_full.err
python -m pytest tests -m gpu -q -s -k "radiance_accumulator or crops or far_from or million or russian" 2>&1 | grep -E "^\.?parity|simple \+|device builder|russian roulette|passed|failed" > $OUT/${TAG}_parity_lines.txt
python tools/parity_report.py $OUT/${TAG}_parity_report.json > $OUT/parity.log 2>&1
python tools/seedprof.py 16 > $OUT/${TAG}_seed_phases.txt 2>&1
python tools/seedprof_ps.py > $OUT/${TAG}_seed_ps_phases.txt 2>&1
python tools/seedprof.py 16 4 > $OUT/${TAG}_seed_w5_phases.txt 2>&1
tools/bin/issueprobe > $OUT/${TAG}_issueprobe.txt 2>&1
tools/bin/roundprobe2 > $OUT/${TAG}_roundprobe2.txt 2>&1
tools/bin/roundprobe3 > $OUT/${TAG}_roundprobe3.txt 2>&1
tools/bin/roundprobe4 > $OUT/${TAG}_roundprobe4_raw.txt 2>&1
tools/bin/simdprobe > $OUT/${TAG}_simdprobe.txt 2>&1
tools/prof_pmc.sh $OUT/pmc $TAG > $OUT/pmc.log 2>&1
cp $OUT/pmc/summary.txt $OUT/${TAG}_pmc_summary.txt; cp $OUT/pmc/pmc_traffic.json $OUT/${TAG}_pmc_traffic.json
tail -3 $OUT/pytest_gpu.log; cat $OUT/${TAG}_bench_full_unprofiled.json.log | head -c 400; echo; cat $OUT/${TAG}_bench_kernel_stats.md | head -12
