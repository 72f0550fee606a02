import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from util import run_hip, run_oracle, synth, GRAD_SCALE, pyoracle, oracle_four_modes, gaussian_noise_scale, CHAIN_ACTIVATED_WIDE
SC=synth.SceneConfig
dev=torch.device('cuda:0')
def masked(scene, ref):
    W,H=scene['W'],scene['H']
    keep=torch.from_numpy(~ref['border'].astype(bool)).to(torch.float32)
    g=synth.make_upstream_grads(W,H,seed=1,scale=GRAD_SCALE)
    return {k:v*keep.reshape((1,)*(v.dim()-2)+(H,W)) for k,v in g.items()}
cases=[("rot4d_sh3_t1", SC("v", 8000, 200, 120, 3, 1, 0.03, 2.0, True, 4, False), dict(bg=(1.0,1.0,1.0), st_scale=2.0), "rig1", [6475]),
       ("rot4d_sh3_t2", SC("v", 12000, 320, 240, 3, 2, 0.02, 2.0, True, 4, False), dict(random_flow=True, st_scale=3.0), "axis", [6233])]
for name,cfg,kw,pose,gs in cases:
    scene=synth.make_scene(cfg,seed=3,pose=pose,rot_sigma="uniform",**kw)
    o=pyoracle.Oracle(scene,kind='port'); ref=dict(o.forward())
    grads=masked(scene,ref)
    refg,rev,f64,prb=oracle_four_modes(o,grads)
    o.close()
    nu_o=gaussian_noise_scale([refg,rev],f64,CHAIN_ACTIVATED_WIDE)
    nu_p=gaussian_noise_scale([prb],f64,CHAIN_ACTIVATED_WIDE)
    runs=[run_hip(scene,dev,grads)[1] for _ in range(4)]
    print("==",name,pose)
    for g in gs:
        print("Gaussian",g,"radius",ref['radii'][g],"tiles",ref['tiles_touched'][g],"nu_order %.2e nu_probe %.2e"%(nu_o[g],nu_p[g]), 'conic_op', ref['conic_opacity'][g])
        for k in ('dL_dmean2D','dL_dopacity','dL_dcolor','dL_dcov3D','dL_dmean3D','dL_dts','dL_dscale','dL_dscale_t','dL_drot','dL_drot_r'):
            sc=max(1.0,float(np.abs(refg[k]).max()))
            print("  %-12s scale %.1e f64 %s"%(k,sc,np.array2string(f64[k][g].reshape(-1),precision=5)))
            print("      ref-f64 %s rev-f64 %s probe-f64 %s"%(np.array2string((refg[k][g]-f64[k][g]).reshape(-1),precision=2),np.array2string((rev[k][g]-f64[k][g]).reshape(-1),precision=2),np.array2string((prb[k][g]-f64[k][g]).reshape(-1),precision=2)))
            for r in runs:
                print("      hip-f64 %s"%np.array2string((r[k].reshape(refg[k].shape)[g]-f64[k][g]).reshape(-1),precision=2))
    # global: ratio stats with both nus
    for K in (4.0,):
        for k in CHAIN_ACTIVATED_WIDE:
            sc=max(1.0,float(np.abs(refg[k]).max())); P=refg[k].shape[0]
            worst_o=worst_p=0
            for r in runs:
                e=np.abs(r[k].reshape(refg[k].shape).astype(np.float64)-f64[k]).reshape(P,-1).max(1)
                worst_o=max(worst_o,(e/(1e-4*sc+K*sc*nu_o)).max()); worst_p=max(worst_p,(e/(1e-4*sc+K*sc*np.maximum(nu_o,nu_p))).max())
            print("  %-12s worst ratio with nu_order only %.2f ; with max(nu_order, nu_probe) %.2f"%(k,worst_o,worst_p))
