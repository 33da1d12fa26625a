mkdir -p gpurun_out/s16
{
python tools/nan_probe.py
for mw in 5 6 4; do
  echo "== precise megakernel, min_waves option $mw (5: 128 VGPRs, 6: 96 VGPRs, 4: 168 VGPRs)"
  python tools/ab/split_ab.py --scenes rtcamp6_v3_1,rtcamp6_dodeca,rtcamp6_v2 --samplings 128 --modes 3 --opt min_waves=$mw
done
python tools/ab/split_ab.py --scenes simple,material_examples,cornell_mini --samplings 128 --modes 0,3
} 2>&1 | grep -v libdrm | tee gpurun_out/s16/sweep.txt
