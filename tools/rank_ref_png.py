import sys, time, glob, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import tinyraytracing_amd as T, oracle_lib as O
from PIL import Image
R='/root/reference/RayTracingOnCPU/example-scenes-cg22/'
def lin8(a): return ((a.astype(np.float64)+0.5)/255.0)**2.2
def blocks(img,b):
    h,w,_=img.shape
    return img[:h//b*b,:w//b*b].reshape(h//b,b,w//b,b,3).mean(axis=(1,3))
name=sys.argv[1]; spp=int(sys.argv[2]); seed={'veach-mis':0x5EED0002,'staircase':T.SEED_STAIRCASE,'back':T.SEED_BACK}[name]
d={'back':'test'}.get(name,name)
files=sorted(glob.glob(R+d+'/image*.png'))
W,H=Image.open(files[0]).size
sc=T.Scene.named(name,W,H)
res={}
for flags in (0,8):
    img,st=O.render(sc.flat,T.make_params(W,H,spp,seed,flags=flags))
    res[flags]=blocks(np.clip(img.astype(np.float64),0,1),16)
for f in files:
    a=np.asarray(Image.open(f).convert('RGB'))
    if a.shape[:2]!=(H,W): continue
    rb=blocks(lin8(a),16)
    out=[]
    for flags in (0,8):
        ob=res[flags]
        rel=np.abs(ob-rb)/(0.02+rb)
        lo=np.log(ob.sum(axis=2)+0.01).ravel(); lr=np.log(rb.sum(axis=2)+0.01).ravel()
        out.append((round(float(np.median(rel)),3),round(float(np.percentile(rel,90)),3),round(float(np.corrcoef(lo,lr)[0,1]),4)))
    print(os.path.basename(f),'parity',out[0],'fixed',out[1])
