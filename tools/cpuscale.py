import sys, time, os
sys.path.insert(0,'hanamaru-renderer_amd/python'); sys.path.insert(0,'oracle')
import hanamaru_amd as ha, oracle_py as orc
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cpu.max", e)
sc=ha.Scene("rtcamp6_v3_1"); o=orc.OracleScene(sc.desc_ptr)
for th in (1,4,16,32,64,128,256):
    t=time.time(); a,_=o.render(960,540,1,2,threads=th); dt=time.time()-t
    print(th, "threads: %.3f Mpaths/s"%(960*540*4/dt/1e6))
