import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
import bench, torch
from x265_amd import api
W,H,D,F=1920,1080,8,60
clip, chroma = bench.make_clip(W,H,F,D,1234)
dy=torch.from_numpy(clip).cuda(); du=torch.from_numpy(np.stack([c[0] for c in chroma])).cuda(); dv=torch.from_numpy(np.stack([c[1] for c in chroma])).cuda()
torch.cuda.synchronize()
ctx = api.Context(W,H,depth=D,num_slots=F, stream=torch.cuda.current_stream().cuda_stream)
n=ctx.geom.cuCount
class AqPic(C.Structure):
    _fields_=[("slot",C.c_int32),("y",C.c_void_p),("u",C.c_void_p),("v",C.c_void_p),("strideY",C.c_ssize_t),("strideC",C.c_ssize_t),("quantOffsets",C.c_void_p),("qpAq",C.c_void_p),("qpCu",C.c_void_p),("inv",C.c_void_p),("wpSum",C.c_uint64*3),("wpSsd",C.c_uint64*3)]
pics=(AqPic*F)()
keep=[]
for i in range(F):
    a,b,c=np.zeros(n),np.zeros(n),np.zeros(n,np.int32); keep+= [a,b,c]
    pics[i].slot=i; pics[i].y=dy.data_ptr()+i*W*H; pics[i].u=du.data_ptr()+i*(W//2)*(H//2); pics[i].v=dv.data_ptr()+i*(W//2)*(H//2)
    pics[i].strideY=W; pics[i].strideC=W//2; pics[i].qpAq=a.ctypes.data; pics[i].qpCu=b.ctypes.data; pics[i].inv=c.ctypes.data
aqp = api.AqParams(2, 1, 0, 0, 1.0, 32, 0, None, None, None, 0, 0, 0.0, None)
for th in ("1","4","16"):
    os.environ["X265GPU_HOST_THREADS"]=th
    for rep in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        rc=ctx.L.x265gpu_aq_frames(ctx.h, F, pics, api.MEM_DEVICE, C.byref(aqp)); assert rc==0
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("threads",th,"aq_frames x60: %.2f ms"%(dt*1e3))
aqp0 = api.AqParams(0, 1, 0, 0, 1.0, 32, 0, None, None, None, 0, 0, 0.0, None)
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    rc=ctx.L.x265gpu_aq_frames(ctx.h, F, pics, api.MEM_DEVICE, C.byref(aqp0)); assert rc==0
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("aqMode 0 (device pass + wp only): %.2f ms"%(dt*1e3))
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
