import sys, os
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
import hanamaru_amd as ha
from PIL import Image
ref = np.asarray(Image.open(os.path.join(ROOT,"tests","golden","reference_rtcamp6_1000x4spp.png")).convert("RGB")).astype(np.float64)
r = ha.Renderer(0)
sc = ha.Scene("rtcamp6_v3_1")
r.upload_scene(sc); r.set_resolution(1920,1080)
for prec in (0,1):
    r.set_option("precise_shading", prec)
    r.clear(); r.render(1,1001)
    img = r.resolve(1000).astype(np.float64)
    d = np.abs(img-ref)
    print("precise %d: PSNR %.2f dB, identical %.5f, within 1 LSB %.6f, max %d, differing channels %d" % (prec, 10*np.log10(255.0**2/(d**2).mean()), (d==0).mean(), (d<=1).mean(), d.max(), int((d!=0).sum())), flush=True)
