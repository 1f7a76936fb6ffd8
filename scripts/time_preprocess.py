import sys, time, torch
sys.path.insert(0, "/root/repo")
from vima_amd import preprocess
from oracle.preprocess_oracle import synthetic_frames
dev="cuda:0"
fr = {v: synthetic_frames(1, 4, seed=100+i) for i, v in enumerate(("front","top"))}
rgb = {v: torch.from_numpy(fr[v][0]).to(dev) for v in fr}; seg = {v: torch.from_numpy(fr[v][1]).to(dev) for v in fr}; ids = fr["front"][2]
def t(fn, n=50):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("crop_objects one view: %.3f ms" % t(lambda: preprocess.crop_objects(rgb["front"], seg["front"], ids)))
meta = {"n_objects": 4, "obj_id_to_info": {i: {} for i in ids}}
print("prepare_obs 2 views: %.3f ms" % t(lambda: preprocess.prepare_obs(obs={"ee": torch.tensor([0]), "rgb": dict(rgb), "segm": dict(seg)}, rgb_dict=None, meta=meta, device=dev)))
idt = torch.tensor(ids, dtype=torch.int32, device=dev)
import ctypes
from vima_amd import _lib
lib=_lib.load()
crops=torch.empty(1,4,3,32,32,dtype=torch.uint8,device=dev); bb=torch.empty(1,4,4,dtype=torch.int64,device=dev); mk=torch.empty(1,4,dtype=torch.bool,device=dev)
p=lambda x: ctypes.c_void_p(x.data_ptr())
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("raw C call: %.3f ms" % t(lambda: lib.vima_crop_objects(p(rgb["front"]), p(seg["front"]), 1, p(idt), 1, 4, 128, 256, p(crops), p(bb), p(mk), st)))
