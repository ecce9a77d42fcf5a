import sys, os, json
sys.path.insert(0, 'jetracer-orbslam2_amd')
import torch, numpy as np
import orbfe
from orbfe import synth
lib_path = os.path.abspath(sys.argv[1])
orbfe.LIB_PATH = lib_path
w,h,B=640,480,256
ctx = orbfe.Context(w,h,max_batch=B,levels=8,cell=8,min_arc=9,max_features=2000)
base = synth.frames(w,h,16,first_index=1000,kind='rects',**synth.DENSE)
dev=torch.device('cuda')
frames = torch.from_numpy(base).to(dev)[torch.arange(B,device=dev)%16].contiguous()
s=torch.cuda.current_stream().cuda_stream
ctx.build_pyramid(frames.data_ptr(), w, w*h, B, s)
def timed(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
print(lib_path, 'detect ms', timed(lambda: ctx.detect_batch(B,s)))
