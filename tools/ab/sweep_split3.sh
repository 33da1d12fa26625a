mkdir -p gpurun_out/s13
{
for b in 512 768 1024 1280 1536 2048; do
  python tools/ab/split_ab.py --scenes rtcamp6_v3_1,tbf3 --samplings 128 --modes 2 --opt trace_budget=$b
done
python tools/ab/split_ab.py --scenes rtcamp6_v3_1,tbf3 --samplings 128 --modes 0,2
} 2>&1 | grep -v libdrm | tee gpurun_out/s13/sweep.txt
