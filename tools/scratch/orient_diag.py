import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from util import run_hip, run_oracle, synth, GRAD_SCALE, pyoracle
SC=synth.SceneConfig
dev=torch.device('cuda:0')
def masked(scene, ref):
    W,H=scene['W'],scene['H']
    keep=torch.from_numpy(~ref['border'].astype(bool)).to(torch.float32)
    g=synth.make_upstream_grads(W,H,seed=1,scale=GRAD_SCALE)
    return {k:v*keep.reshape((1,)*(v.dim()-2)+(H,W)) for k,v in g.items()}
cases=[("rot4d_sh3_t1", SC("v", 8000, 200, 120, 3, 1, 0.03, 2.0, True, 4, False), dict(bg=(1.0,1.0,1.0), st_scale=2.0)),
       ("rot4d_sh3_t2", SC("v", 12000, 320, 240, 3, 2, 0.02, 2.0, True, 4, False), dict(random_flow=True, st_scale=3.0)),
       ("rot4d_sh0", SC("v", 6000, 200, 200, 0, 0, 0.03, 1.0, True, 4, True), dict(st_scale=2.0))]
for name,cfg,kw in cases:
  for pose in ("axis","rig1"):
    scene=synth.make_scene(cfg,seed=3,pose=pose,rot_sigma="uniform",**kw)
    ref,_=run_oracle(scene,None,kind='port')
    grads=masked(scene,ref)
    o=pyoracle.Oracle(scene,kind='port'); o.forward()
    args=(grads['grad_color'],grads['grad_depth'],grads['grad_alpha'],grads['grad_flow'])
    refg={k:v.copy() for k,v in o.backward(*args).items()}
    pyoracle.set_accumulation(1); rev={k:v.copy() for k,v in o.backward(*args).items()}
    pyoracle.set_accumulation(2); f64={k:v.copy() for k,v in o.backward(*args).items()}
    pyoracle.set_accumulation(0); o.close()
    hip,hipg=run_hip(scene,dev,grads)
    rad=ref['radii']
    print("==",name,pose)
    for k in ('dL_dmean2D','dL_dmean3D','dL_dcov3D','dL_dscale','dL_dscale_t','dL_drot','dL_drot_r','dL_dts','dL_dopacity'):
        b=refg[k]; a=hipg[k].reshape(b.shape)
        sc=max(1.0,float(np.abs(b).max()))
        d=np.abs(a-b).reshape(b.shape[0],-1).max(1)
        sp=np.abs(rev[k]-b).reshape(b.shape[0],-1).max(1)
        df=np.abs(a-f64[k].reshape(b.shape)).reshape(b.shape[0],-1).max(1)
        rf=np.abs(b-f64[k].reshape(b.shape)).reshape(b.shape[0],-1).max(1)
        i=int(np.argmax(d))
        small=rad<=40
        print("%-12s scale %.1e hip-ref %.2e (at g %d radius %d) ref-ref' %.2e hip-f64 %.2e ref-f64 %.2e | radius<=40: hip-ref %.2e ref-ref' %.2e ; n beyond 1e-4: %d, of them radius<=40: %d"%(
            k,sc,d.max(),i,rad[i],sp.max(),df.max(),rf.max(),d[small].max(),sp[small].max(),int((d>1e-4*sc).sum()),int(((d>1e-4*sc)&small).sum())))
