import sys, torch
sys.path.insert(0, '.')
from lifelong_nnunet_amd import native as nat
DEV='cuda:0'
def timed(fn, iters=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
N=2
for name,C,K,D,H,W in [("enc3.0 128->256 s2 @40x48x40",128,256,40,48,40),("enc2.0 64->128 s2 @80x96x80",64,128,80,96,80),("enc4.0 256->320 s2 @20x24x20",256,320,20,24,20)]:
    x=(torch.randn(N,D,H,W,C,device=DEV)*0.5).half()
    w=torch.randn(K,C,3,3,3,device=DEV)*0.05
    wp=torch.empty(nat.query("lnn_packed_weight_elems",27,K,C),dtype=torch.float16,device=DEV); nat.call("lnn_pack_weights",w,wp,27,K,C,C*27,27,1)
    wd=torch.empty(nat.query("lnn_packed_weight_elems",27,C,K),dtype=torch.float16,device=DEV); nat.call("lnn_pack_weights",w,wd,27,C,K,27,C*27,1)
    b=torch.zeros(K,device=DEV)
    Do,Ho,Wo=D//2,H//2,W//2
    y=torch.empty(N,Do,Ho,Wo,K,dtype=torch.float16,device=DEV)
    dy=(torch.randn(N,Do,Ho,Wo,K,device=DEV)*0.5).half()
    dx=torch.zeros(N,D,H,W,C,dtype=torch.float16,device=DEV)
    sk=torch.empty(64*N*D*H*W*8,device=DEV) if D*H*W*N*C*64 < 2e9 else torch.empty(1<<28,device=DEV)
    gf=2*27*C*K*N*Do*Ho*Wo/1e9
    out=[]
    for mode in (0,1):
        nat.lib().lnn_debug_set_gen_mode(mode)
        tf=timed(lambda: nat.call("lnn_conv3d_fwd",x,C,wp,b,y,K,N,D,H,W,C,K,2))
        td=timed(lambda: nat.call("lnn_conv3d_dgrad_ws",dy,K,wd,dx,C,N,D,H,W,C,K,2,1,sk,sk.numel()))
        out.append((tf,td))
    nat.lib().lnn_debug_set_gen_mode(-1)
    print(f"{name:32s} fwd tile/stream {out[0][0]:7.1f} us ({gf/out[0][0]*1e3:5.0f} TF/s)  gen {out[1][0]:7.1f} us | dgrad(acc) up2 {out[0][1]:7.1f} us  gen {out[1][1]:7.1f} us")
