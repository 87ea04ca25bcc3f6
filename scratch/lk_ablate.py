import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from _pkg import load_package
import bench
vsf = load_package()
dev = torch.device('cuda',0)
B=256
frames = bench.synth_frames(B, 3, dev, 0)
torch.cuda.synchronize()
W,H = bench.W, bench.H
def run(max_iter, max_level, eps=0.01):
    p = vsf.default_params(grid_rows=4,grid_cols=4,max_features=300,min_distance=10.0,klt_win=21,klt_max_level=max_level,klt_max_iter=max_iter,klt_eps=eps,backtrack_thr=1.0,width=W,height=H,streams=B)
    ctx = vsf.Context(p)
    fs = H*W
    ctx.batch_step(frames.data_ptr(), fs, W, 0)           # detect on frame 0
    ids,xy,age,n,_ = ctx.batch_download()
    res=[]
    for rep in range(3):
        ctx.batch_upload(ids,xy,age,n)
        ctx.reset_tracker()
        ctx.batch_track(frames.data_ptr(), fs, W, 0)
        ctx.profile_enable(True)
        ctx.batch_track(frames.data_ptr()+B*fs, fs, W, 0)
        ctx.profile_enable(False)
        pr = ctx.profile_read()
        res.append(pr['lk'][0])
    _,_,_,n2,st = ctx.batch_download()
    print('max_iter',max_iter,'levels',max_level+1,'lk ms',min(res),'feat',n.mean(),'kept',n2.mean(), 'fwd',st[:,1].mean(),'bwd',st[:,2].mean())
    ctx.close()
for mi in (0,1,2,3,5,30):
    run(mi,3)
for ml in (0,1,2):
    run(30,ml)
run(30,3,eps=0.1)
