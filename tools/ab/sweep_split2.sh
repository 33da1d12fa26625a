mkdir -p gpurun_out/s6
{
for o in "wf_trav_wgs=8" "wf_trav_wgs=6" "wf_trav_wgs=5" "wf_trav_wgs=4" "wf_trav_wgs=3" "wf_trav_wgs=4 --opt wf_shade_wgs=4" "wf_trav_wgs=4 --opt wf_shade_wgs=2" "wf_trav_wgs=3 --opt wf_shade_wgs=3"; do
  python tools/ab/split_ab.py --scenes rtcamp6_v3_1 --samplings 64 --modes 1 --opt wf_adv_den=2 --opt $o
done
python tools/ab/split_ab.py --scenes rtcamp6_v3_1 --samplings 64 --modes 0
} 2>&1 | grep -v libdrm | tee gpurun_out/s6/sweep.txt
