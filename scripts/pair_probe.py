import sys, os, torch
sys.path.insert(0, os.getcwd())
from attention_based_tbn_amd._lib import call, ptr
def mk(n,h,w,cin,cout):
    x=torch.randn(n,h,w,cin,device="cuda"); wt=torch.randn(cout,3,3,cin,device="cuda")*0.05; b=torch.zeros(cout,device="cuda")
    y=torch.empty(n,h,w,cout,device="cuda"); part=torch.empty((n*h*w//128+8)*2*cout,device="cuda")
    return dict(x=x,wt=wt,b=b,y=y,part=part,n=n,h=h,w=w,cin=cin,cout=cout)
def launch(d, st, flags=4, mt=1, nt=1):
    call("tbn_conv2d_fwd_tile", ptr(d["x"]), d["cin"], ptr(d["wt"]), ptr(d["b"]), ptr(d["y"]), d["cout"], d["n"], d["h"], d["w"], d["cin"], d["cout"], 3,1,1, 1, flags, ptr(d["part"]), mt, nt, st)
s1=torch.cuda.Stream(); s2=torch.cuda.Stream()
for name,(a,b) in {"4c 14x14 (128->160 | 128->160)":((96,14,14,128,160),(96,14,14,128,160)), "5a 7x7 (192->320 | 160->224)":((96,7,7,192,320),(96,7,7,160,224)), "3b 28x28 (64->96 | 64->96)":((96,28,28,64,96),(96,28,28,64,96)), "4a 14x14 (64->96 | 96->128)":((96,14,14,64,96),(96,14,14,96,128))}.items():
    A,B=mk(*a),mk(*b)
    def seq():
        launch(A, torch.cuda.current_stream().cuda_stream); launch(B, torch.cuda.current_stream().cuda_stream)
    def par():
        e=torch.cuda.Event(); e.record()
        s1.wait_event(e); s2.wait_event(e)
        launch(A, s1.cuda_stream); launch(B, s2.cuda_stream)
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    for fn,label in ((seq,"sequential"),(par,"two streams")):
        for _ in range(3): fn()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:34s} {label:12s} {e0.elapsed_time(e1)/20*1e3:7.1f} us")
