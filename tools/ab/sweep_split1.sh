mkdir -p gpurun_out/s5
{
python tools/ab/split_ab.py --scenes rtcamp6_v3_1,rtcamp6_v2 --counters --profile-only
for o in "wf_adv_den=2" "wf_adv_den=8" "wf_adv_den=0" "wf_trav_wgs=6" "wf_trav_wgs=5" "wf_trav_wgs=4" "wf_trav_wgs=12"; do
  echo "=== $o"; python tools/ab/split_ab.py --scenes rtcamp6_v3_1,rtcamp6_v2 --profile-only --opt $o | grep -v "step  [3-9]\|step 10"
done
} 2>&1 | grep -v libdrm | tee gpurun_out/s5/sweep.txt
