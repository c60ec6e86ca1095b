import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/intel-texture-works-plugin_amd')
import numpy as np, torch, itw_amd
from itw_amd import surfaces
from oracle import pyoracle
dev=torch.device('cuda:0'); torch.cuda.set_device(dev)
itw_amd.lib().itwWarmupBC45()
for fmt,prof,h,w in (("bc1",None,256,256),("bc4",None,256,256),("bc7","slow",256,256),("bc7","slow",2048,1024),("bc7","alpha_slow",1024,1024),("bc6h","slow",128,128)):
    img = surfaces.hdr_smooth(h,w) if fmt=="bc6h" else surfaces.ldr_smooth(h,w)
    d=torch.from_numpy(img).to(dev)
    ref=itw_amd.compress(fmt,d,prof); torch.cuda.synchronize(); ref=ref.clone()
    out=torch.zeros_like(ref)
    try:
        g=torch.cuda.CUDAGraph()
        s=torch.cuda.Stream()
        with torch.cuda.stream(s):
            itw_amd.compress(fmt,d,prof,out=out)   # warm-up on the side stream (workspace sized)
        torch.cuda.synchronize()
        out.zero_()
        with torch.cuda.graph(g):
            itw_amd.compress(fmt,d,prof,out=out)
        torch.cuda.synchronize()
        out.zero_(); g.replay(); torch.cuda.synchronize()
        print(fmt,prof,h,w,"captured; replay equals eager:", bool(torch.equal(out,ref)))
        out.zero_(); g.replay(); g.replay(); torch.cuda.synchronize(); print("   twice:", bool(torch.equal(out,ref)))
    except Exception as e:
        print(fmt,prof,h,w,"capture failed:", repr(e)[:300])

# eager launches vs graph replay, device-resident small calls (the plugin's slice is 16 384 blocks = 64 x 4096 texels)
print("call time, eager vs hipGraph replay (HIP events, 50 calls):")
for fmt,prof,h,w in (("bc7","slow",64,4096),("bc7","basic",64,4096),("bc7","slow",256,4096),("bc6h","slow",64,4096),("bc1",None,64,4096),("bc7","slow",1024,4096)):
    img = surfaces.hdr_smooth(h,w) if fmt=="bc6h" else surfaces.ldr_smooth(h,w)
    d=torch.from_numpy(img).to(dev)
    out=itw_amd.compress(fmt,d,prof); torch.cuda.synchronize()
    def timed(fn,n=50):
        fn(); torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b)/n
    def timed_sync(fn,n=50):
        import time
        fn(); torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(n): fn(); torch.cuda.synchronize()
        return (time.perf_counter()-t0)/n*1e3
    eager=lambda: itw_amd.compress(fmt,d,prof,out=out)
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        itw_amd.compress(fmt,d,prof,out=out)
    rep=lambda: g.replay()
    print(f"  {fmt} {prof} {h//4*w//4:7d} blocks: back to back eager {timed(eager):.4f} ms, graph {timed(rep):.4f} ms | one call + sync: eager {timed_sync(eager):.4f} ms, graph {timed_sync(rep):.4f} ms")
