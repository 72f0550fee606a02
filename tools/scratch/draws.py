import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from util import synth, gaussian_noise_scale
from test_gpu_parity import _timed_path_vs_oracle
dev=torch.device('cuda:0')
dump=[]
N=int(sys.argv[1]) if len(sys.argv)>1 else 40
_timed_path_vs_oracle(synth.CONFIGS["C3"], dev, 1, "C3-axis", 1e-3, tile_cull=True, repeats=N, dump=dump)
ora=dump[1]; draws=[dump[0]]+dump[2:]
chain=("_scaling","_scaling_t","_rotation","_rotation_r")
nu=gaussian_noise_scale([ora['want'],ora['want_rev'],ora['want_probe']],ora['want_f64'],chain)
for n in chain:
    X=np.stack([d[n].reshape(d[n].shape[0],-1) for d in draws]).astype(np.float64)  # [N,P,k]
    f64=ora['want_f64'][n].reshape(X.shape[1],-1)
    sc=max(1.0,float(np.abs(ora['want'][n]).max()))
    med=np.median(X,0)
    dev_=np.abs(X-med)          # deviation from the median over draws
    mad=np.median(dev_,0)+1e-12
    e=np.abs(X-f64)
    need=np.maximum(e.max(2)-1e-4*sc,0)/(sc*nu+1e-30)[None,:]
    print("==",n,"scale %.3g"%sc,"K needed per draw:",np.array2string(need.max(1),precision=2))
    # top outliers: element & draw with the largest deviation from the per-element median
    flat=dev_.max(2)  # [N,P]
    idx=np.dstack(np.unravel_index(np.argsort(-flat,axis=None)[:8],flat.shape))[0]
    for d_,g in idx:
        print("   draw %d gaussian %d: |x-median| %.3e (%.1e of scale), median-f64 %.3e, MAD %.2e, nu_g %.2e, values over draws min %.5e max %.5e median %.5e f64 %.5e"%(
            d_,g,flat[d_,g],flat[d_,g]/sc,np.abs(med[g]-f64[g]).max(),mad[g].max(),nu[g],X[:,g,:].min(),X[:,g,:].max(),med[g].flat[0],f64[g].flat[0]))
    # how often is an element far from the median (in units of scale)?
    for thr in (1e-4,3e-4,1e-3):
        print("   elements x draws deviating from their median-over-draws by > %.0e of scale: %d"%(thr,int((flat>thr*sc).sum())))
np.savez_compressed(os.path.join(ROOT,'gpurun_out','r6_draws_scaling_t.npz'), X=np.stack([d['_scaling_t'].reshape(-1) for d in draws]), f64=ora['want_f64']['_scaling_t'].reshape(-1), ref=ora['want']['_scaling_t'].reshape(-1), nu=nu)
