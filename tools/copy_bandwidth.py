import ctypes, sys, torch
sys.path.insert(0,'.')
from fiveeqscm_amd import _capi
lib=_capi.load()
for n in (1<<27, 1<<28):
    src=torch.empty(n,dtype=torch.float64,device='cuda').normal_(); dst=torch.empty_like(src)
    for name,fn in (("8B",lib.fiveeq_stream_copy_f64),("16B",lib.fiveeq_stream_copy_wide_f64)):
        for _ in range(3): fn(n, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), None)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn(n, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), None)
        e1.record(); e1.synchronize()
        print(n, name, 2*n*8*10/(e0.elapsed_time(e1)*1e-3)/1e9, "GB/s", torch.equal(src,dst))
    # torch's own copy for reference
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    dst.copy_(src); torch.cuda.synchronize(); e0.record()
    for _ in range(10): dst.copy_(src)
    e1.record(); e1.synchronize(); print(n, "torch copy_", 2*n*8*10/(e0.elapsed_time(e1)*1e-3)/1e9)
    del src,dst
