import sys; sys.path.insert(0,'.')
import numpy as np
from _pkg import load_package, load_synth
from oracle import oracle as O
vsf, synth = load_package(), load_synth()
w,h,win,ml,eps = 752,480,21,3,0.01
frames,_ = synth.sequence(11,w,h,2)
rng = np.random.default_rng(5)
n=600
pts = np.stack([rng.uniform(-5,w+5,n), rng.uniform(-5,h+5,n)],1).astype(np.float32)
pts[:50]=np.round(pts[:50])
for lvl in (0,1,3):
    ctx = vsf.Context(vsf.default_params(width=w,height=h,klt_win=win,klt_max_level=lvl,klt_eps=eps))
    got,gst = ctx.debug_lk(frames[0],frames[1],pts)
    ref,rst,_ = O.calc_optical_flow_pyr_lk(frames[0],frames[1],pts,win,lvl,30,eps,acc_mode=O.ACC_EXACT)
    bad = np.where((gst!=rst) | ((gst==1)&(np.abs(got-ref).max(1)>0)))[0]
    print('maxlevel',lvl,'status mismatches',(gst!=rst).sum(),'pos mismatches',len(bad),'of',n)
    for i in bad[:8]:
        print('  ',i,pts[i],got[i],gst[i],ref[i],rst[i])
    ctx.close()
