#!/bin/bash
# Everything under profiles/ for one round, on the GPU box:  tools/final_profiles.sh r06
# SKIP_PMC=1: the bench lines, kernel statistics and parity lines only — when bench.py changed but the kernel sources did not (the PMC passes
# under profiles/ stay valid: bench.py checks their sha of csrc/)
TAG=${1:-r06}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -z "$SKIP_PMC" ]; then
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|libdrm" | tail -5 > $OUT/pytest_gpu.log
# the PMC passes first: every bench line below reads its `traffic` from profiles/${TAG}_pmc_traffic[_scene].json and checks the kernel sources' hash against it
tools/prof_pmc.sh $OUT/pmc $TAG > $OUT/pmc.log 2>&1
cp $OUT/pmc/summary.txt $OUT/${TAG}_pmc_summary.txt; cp $OUT/pmc/pmc_traffic.json $OUT/${TAG}_pmc_traffic.json
cp $OUT/${TAG}_pmc_summary.txt $OUT/${TAG}_pmc_traffic.json profiles/
fi
# the scenes whose trace kernel is the slower kernel of the pair (or level with it): their own PMC passes, bench line and kernel statistics
for sc in rtcamp6_v2 rtcamp6_v1 tbf3; do
  if [ -z "$SKIP_PMC" ]; then
  tools/prof_pmc.sh $OUT/pmc_$sc $TAG $sc > $OUT/pmc_$sc.log 2>&1
  cp $OUT/pmc_$sc/summary.txt $OUT/${TAG}_pmc_summary_$sc.txt; cp $OUT/pmc_$sc/pmc_traffic.json $OUT/${TAG}_pmc_traffic_$sc.json
  cp $OUT/${TAG}_pmc_summary_$sc.txt $OUT/${TAG}_pmc_traffic_$sc.json profiles/
  fi
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$sc -o ${TAG}_$sc -- python bench.py --scene $sc --spp-per-step 16 --steps 32 --no-cpu-baseline > $OUT/${TAG}_bench_$sc.json.log 2>> $OUT/bench_full.err
  DB=$(find $OUT/prof_$sc -name "*_results.db" | head -1)
  python tools/rocprof_summary.py "$DB" $OUT/${TAG}_bench_${sc}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --scene $sc --spp-per-step 16 --steps 32 --no-cpu-baseline" > /dev/null
done
# the bench line and the kernel-trace statistics of the SAME run
rocprofv3 --kernel-trace --stats -d $OUT/prof -o ${TAG} -- python bench.py > $OUT/${TAG}_bench_full.json.log 2> $OUT/bench_full.err
DB=$(find $OUT/prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" $OUT/${TAG}_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py" > /dev/null
# unprofiled run of the same command, and the driver's form of it (--steps 20)
python bench.py > $OUT/${TAG}_bench_full_unprofiled.json.log 2>> $OUT/bench_full.err
python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_steps20_as_the_driver.json.log 2>> $OUT/bench_full.err
# the shortcuts of round 5 switched off, same box: every NEE shadow ray traced (bit-identical image), every trace workgroup kept
python bench.py --spp-per-step 16 --steps 32 --no-cpu-baseline --debug nee_cull=0 > $OUT/${TAG}_bench_full_nee_cull_off.json.log 2>> $OUT/bench_full.err
python bench.py --spp-per-step 16 --steps 32 --no-cpu-baseline --debug trace_budget=1536 > $OUT/${TAG}_bench_full_wave_budget_off.json.log 2>> $OUT/bench_full.err
# other BASELINE configurations on one GPU
python bench.py --scene spheres --steps 4 --spp-per-step 16 --no-cpu-baseline > $OUT/${TAG}_bench_c2_spheres.json.log 2>> $OUT/bench_full.err
python bench.py --scene rtcamp6_dodeca --width 3840 --height 2160 --steps 64 --warmup 1 --spp-per-step 4 --no-cpu-baseline > $OUT/${TAG}_bench_c5_4k_dodeca.json.log 2>> $OUT/bench_full.err
python bench.py --bvh-builder 1 --spp-per-step 16 --steps 16 --no-cpu-baseline > $OUT/${TAG}_bench_lbvh.json.log 2>> $OUT/bench_full.err
python bench.py --bvh-builder 2 --spp-per-step 16 --steps 16 --no-cpu-baseline > $OUT/${TAG}_bench_ploc.json.log 2>> $OUT/bench_full.err
# N > 1 exactly as the driver runs N = 1 (no launcher): one process, two contexts; on this 1-GPU box both share device 0
python bench.py --gpus 2 --spp-per-step 16 --steps 8 --no-cpu-baseline > $OUT/${TAG}_bench_gpus2_one_device.json.log 2>> $OUT/bench_full.err
# the command lines of BASELINE configs 4 and 5 with everything but the node: eight contexts on this one device at full size (8 x 2 hand-off
# buffers of 4.25 GB + 8 scenes in HBM), weak scaling as the driver runs it and the fixed totals of C4 / C5 cut to 64 / 32 samplings
python bench.py --gpus 8 --spp-per-step 16 --steps 2 --no-cpu-baseline --no-counters > $OUT/${TAG}_bench_gpus8_one_device.json.log 2>> $OUT/bench_full.err
python bench.py --gpus 8 --total-samplings 64 --steps 2 --no-cpu-baseline --no-counters > $OUT/${TAG}_bench_c4_strong_64_of_4096_one_device.json.log 2>> $OUT/bench_full.err
python bench.py --scene rtcamp6_dodeca --width 3840 --height 2160 --gpus 8 --total-samplings 32 --steps 2 --warmup 1 --max-tail-gib 5 --no-cpu-baseline --no-counters > $OUT/${TAG}_bench_c5_strong_32_of_1024_one_device.json.log 2>> $OUT/bench_full.err
python bench.py --russian-roulette 3 --spp-per-step 16 --steps 16 --no-cpu-baseline > $OUT/${TAG}_bench_russian_roulette_nonparity.json.log 2>> $OUT/bench_full.err
# every scene at 1080p, one line each
for sc in rtcamp6_v3_1 rtcamp6_dodeca spheres rtcamp6_v3 rtcamp6_v2 rtcamp6_v1 rtcamp5 tbf3 material_examples simple cornell_mini; do
  python bench.py --scene $sc --spp-per-step 16 --steps 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; g = r['priority_governor']
print('%-18s %7.1f Mpaths/s  trace %.2f ms (alone %.2f)  seed %.2f ms  rays/path %.2f (+ %.2f culled)  nodes/ray %.1f  tris/ray %.1f  lanes/box %.1f  8d-frac %.2f  pair bound %s  gov level %d wgs %s' % ('$sc', d['value'], r['avg_launch_ms'], r.get('avg_launch_ms_alone', 0), r['seed_kernel_avg_ms'], r['rays_per_path'], r['nee_shadow_rays_culled_per_path'], r['node_tests_per_ray'], r['tri_tests_per_ray'], r['lanes_per_box_pass'], r['frac'], r['pair_bound'], g['level'], g['trace_workgroups']))"
done > $OUT/${TAG}_all_scenes_1080p.txt
python -m pytest tests -m gpu -q -s -k "radiance_accumulator or crops or far_from or million or russian or per_path or post_chain or full_length or ragged or nee_culls" 2>&1 | grep -E "^\.?parity|simple \+|device builder|builder [0-9]|4 M tri|4K post|per-path|russian roulette|config [45] full|ragged sizes|nee culls|passed|failed" > $OUT/${TAG}_parity_lines.txt
python tools/parity_report.py $OUT/${TAG}_parity_report.json > $OUT/parity.log 2>&1
python tools/seedprof.py 16 > $OUT/${TAG}_seed_phases.txt 2>&1
# round 6: the split pipeline against the megakernel and with precise shading, pair and kernel by kernel; precise shading's parity; the CLI's rate
python tools/ab/split_ab.py --samplings 128 --scenes rtcamp6_v3_1,rtcamp6_v2,rtcamp6_v1,rtcamp6_dodeca,tbf3,rtcamp5,spheres --modes 0,1,3,2 2>&1 | grep -v libdrm > $OUT/${TAG}_split_ab.txt
python tools/ab/split_ab.py --samplings 128 --scenes simple,material_examples,cornell_mini --modes 0,3 2>&1 | grep -v libdrm >> $OUT/${TAG}_split_ab.txt
python tools/ab/split_ab.py --scenes rtcamp6_v3_1,rtcamp6_v2,spheres --modes 2 --profile-only --counters 2>&1 | grep -v libdrm > $OUT/${TAG}_split_profile.txt
python tools/ab/precise_check.py 2>&1 | grep -v libdrm > $OUT/${TAG}_precise_parity_480x270.txt
# precise shading: fp32 / megakernel form / split form on every scene; the figures behind PATH_LIMITS_PRECISE; the draws' residuals on and off
python tools/ab/split_ab.py --modes 0,3,2 --scenes spheres,simple,material_examples,cornell_mini,rtcamp6_v3_1,rtcamp6_v3,rtcamp5,tbf3,rtcamp6_dodeca,rtcamp6_v2,rtcamp6_v1 2>&1 | grep -v libdrm > $OUT/${TAG}_precise_pipelines.txt
bash tools/ab/precise_limits.sh 2>/dev/null | grep precise > $OUT/${TAG}_precise_limits.txt
( echo "# draw_residuals 1 (default), then 0: modes 0 = fp32 shading, 3 = precise in the megakernel, 2 = precise in the split pipeline"
  python tools/ab/split_ab.py --modes 0,3,2 --scenes spheres,rtcamp6_v3_1 --opt draw_residuals=1 2>&1 | grep -v libdrm
  python tools/ab/split_ab.py --modes 0,3,2 --scenes spheres,rtcamp6_v3_1 --opt draw_residuals=0 2>&1 | grep -v libdrm ) > $OUT/${TAG}_precise_draw_residuals_ab.txt
python -m pytest tests -m gpu -q -s -k "precise or variants or path_draws" 2>&1 | grep -E "precise|passed|failed" > $OUT/${TAG}_precise_tests_gpu.txt
bash tools/ab/cli_batch.sh > $OUT/${TAG}_cli_report_granularity.txt 2>&1
python tools/finite_soak.py 1024 2>/dev/null > $OUT/${TAG}_finite_soak_all_scenes_1024.txt
( echo "# python tools/reference_image_compare.py: hr_render of samplings 1 .. 1000 at 1920x1080 + hr_resolve against tests/golden/reference_rtcamp6_1000x4spp.png"; python tools/reference_image_compare.py 2>/dev/null ) > $OUT/${TAG}_reference_image_compare.txt
python bench.py --precise --spp-per-step 16 --steps 32 --no-cpu-baseline > $OUT/${TAG}_bench_precise.json.log 2>> $OUT/bench_full.err
python bench.py --no-precise --scene spheres --steps 4 --spp-per-step 16 --no-cpu-baseline > $OUT/${TAG}_bench_c2_spheres_fp32_shading.json.log 2>> $OUT/bench_full.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_precise -o ${TAG}_precise -- python bench.py --precise --spp-per-step 16 --steps 16 --no-cpu-baseline --no-counters > /dev/null 2>> $OUT/bench_full.err
DB=$(find $OUT/prof_precise -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" $OUT/${TAG}_bench_precise_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --precise --spp-per-step 16 --steps 16 --no-cpu-baseline --no-counters" > /dev/null
tail -3 $OUT/pytest_gpu.log; cat $OUT/${TAG}_bench_full_unprofiled.json.log | head -c 400; echo; cat $OUT/${TAG}_bench_kernel_stats.md | head -12
