import sys, time, numpy as np, torch
sys.path.insert(0,__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from monoport_amd import synthetic as syn, ops
dev="cuda:0"
layers=syn.body_mlp("G",noise=0.05,seed=1); f=syn.body_feat(256,128,128,2)
mlp=ops.PackedMLP.from_layers(dev,layers,1); fh=ops.pack_features(torch.from_numpy(f)[None].to(dev))
from oracle import pifu_oracle as o
cal=torch.from_numpy(o.pifu_calib(*syn.scene_camera(30))).to(dev)
for n in (4913, 65536, 262144, 1048576):
    p=torch.from_numpy(syn.rand_points(n,3,1.0))[None].to(dev)
    for _ in range(2): ops.query(mlp,fh,p,cal,syn.Z_SCALE)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(5): ops.query(mlp,fh,p,cal,syn.Z_SCALE)
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/5
    print("query N=%d: %.3f ms  %.2f Mpts/s  %.1f TFLOP/s"%(n,ms,n/ms/1e3,n*2363906/ms/1e9))
res=[17,33,65,129,257]
vol=torch.empty(257,257,257,device=dev); st=torch.empty(6,dtype=torch.int32,device=dev)
for _ in range(2): ops.recon(mlp,fh,cal,syn.Z_SCALE,[-1]*3,[1]*3,res,volume=vol,status=st)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.recon(mlp,fh,cal,syn.Z_SCALE,[-1]*3,[1]*3,res,volume=vol,status=st)
e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/5
s=st.cpu().numpy(); print("recon 257: %.3f ms, status %s, total pts %d -> %.1f TFLOP/s"%(ms,s,s[1:].sum(),s[1:].sum()*2363906/ms/1e9))
e0.record()
for _ in range(5): x,y,z,nn,c=ops.forward_vertices_raw(vol)
e1.record(); torch.cuda.synchronize(); print("forward_vertices: %.3f ms, n=%d"%(e0.elapsed_time(e1)/5,int(c.item())))
