import re,sys
d=[];sm=[];mx=[];bad=0
for l in open(sys.argv[1]):
    m=re.search(r"divergent (\d+) ppm.*same>1e-3 (\d+) ppm.*max (\S+) rays_equal (\S+) .*culls identical (\S+)",l)
    if m:
        d.append(int(m.group(1))); sm.append(int(m.group(2))); mx.append(float(m.group(3)))
        if m.group(4)!="True" or m.group(5)!="True": bad+=1
import numpy as np
d=np.array(d);sm=np.array(sm);mx=np.array(mx)
print("scenes %d  divergent ppm: median %.0f p99 %.0f max %.0f   same-branch>1e-3 ppm: mean %.2f p99 %.0f max %.0f  scenes with any: %d   worst same-branch path: median %.2g p99 %.2g max %.3g   not rays_equal/culls identical: %d"%(len(d),np.median(d),np.percentile(d,99),d.max(),sm.mean(),np.percentile(sm,99),sm.max(),(sm>0).sum(),np.median(mx),np.percentile(mx,99),mx.max(),bad))
