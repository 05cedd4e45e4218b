# In-kernel stamps of the layer-tail kernel.  Needs the diagnostic build:
#   bash tools/attn_stamps.sh && TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_stamps.so python tools/mlp_ablate.py
import os, sys, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from titok_video_amd import _lib
lib=_lib.lib(); DEV=torch.device("cuda:0"); S=_lib.stream_ptr(DEV)
L,d,I=36864,256,704
bf=torch.bfloat16
x=(torch.randn(L,d,device=DEV)).to(bf); w12=(torch.randn(2*I,d,device=DEV)*d**-0.5).to(bf); w3=(torch.randn(d,I,device=DEV)*I**-0.5).to(bf)
gain=torch.ones(d,device=DEV); yb=torch.empty(L,d,dtype=bf,device=DEV)
mp=torch.empty(lib.ttv_mlp_pack_bytes(I,0),dtype=torch.uint8,device=DEV)
lib.ttv_mlp_pack(w12.data_ptr(),w3.data_ptr(),None,None,0,I,d,0,mp.data_ptr(),S)
def t(fn,it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/it
for dbg in [0, 1, 2, 4, 16, 4 | 16, 2 | 4 | 16, 0]:      # 1: no stores, 2: no weight DMA after the first panel, 4: no GEGLU arithmetic, 16: no P2 MFMAs
    lib.ttv_debug_set(dbg)
    print(f"debug {dbg:3d}: {t(lambda: lib.ttv_mlp_fused(x.data_ptr(),d,mp.data_ptr(),I,yb.data_ptr(),d,gain.data_ptr(),8.0,1e-5,L,d,0,S)):7.1f} us", flush=True)
lib.ttv_debug_set(0)
# in-kernel stamps of block 0 (wave 0 = producer, wave 4 = consumer)
st=torch.zeros(128,dtype=torch.int64,device=DEV)
lib.ttv_debug_stamps(st.data_ptr())
lib.ttv_mlp_fused(x.data_ptr(),d,mp.data_ptr(),I,yb.data_ptr(),d,gain.data_ptr(),8.0,1e-5,L,d,0,S)
torch.cuda.synchronize(); lib.ttv_debug_stamps(None)
s=st.cpu().tolist()
for role,name in ((0,"P1"),(1,"P2")):
    v=[t for t in s[role*64:role*64+64] if t]
    print(name, "n=",len(v), "deltas:", [v[i+1]-v[i] for i in range(len(v)-1)])
# layer tail (fused front) stamps
ao=(torch.randn(L,d,device=DEV)).to(bf); wo=(torch.randn(d,d,device=DEV)*d**-0.5).to(bf)
lib.ttv_mlp_pack(w12.data_ptr(),w3.data_ptr(),wo.data_ptr(),None,0,I,d,0,mp.data_ptr(),S)
tail=lambda: lib.ttv_layer_tail_fused(ao.data_ptr(),d,gain.data_ptr(),8.0,x.data_ptr(),d,mp.data_ptr(),I,yb.data_ptr(),d,gain.data_ptr(),8.0,1e-5,L,d,0,None,S)
print(f"layer tail: {t(tail):7.1f} us")
st.zero_(); lib.ttv_debug_stamps(st.data_ptr()); tail(); torch.cuda.synchronize(); lib.ttv_debug_stamps(None)
s=st.cpu().tolist()
for role,name in ((0,"A"),(1,"B")):
    v=[q for q in s[role*64:role*64+64] if q]
    print(name, "n=",len(v), "deltas:", [v[i+1]-v[i] for i in range(len(v)-1)])

# layer tail + next qkv stamps
from titok_video_amd.plan import BatchPlan
plan = BatchPlan([(16, 128, 128)] * 32, [128] * 32, (4, 8, 8), DEV)
wq=(torch.randn(768,d,device=DEV)*d**-0.5).to(bf); qkv=torch.empty(L,768,dtype=bf,device=DEV)
mp2=torch.empty(lib.ttv_mlp_pack_bytes(I,768),dtype=torch.uint8,device=DEV)
lib.ttv_mlp_pack(w12.data_ptr(),w3.data_ptr(),wo.data_ptr(),wq.data_ptr(),768,I,d,0,mp2.data_ptr(),S)
nxq=_lib.NextQkv(qkv=qkv.data_ptr(),ld=768,rope_cs=plan.rope_cs.data_ptr(),rows=768,rope_q_end=256,rope_k_begin=512,rope_k_end=640)
tail2=lambda: lib.ttv_layer_tail_fused(ao.data_ptr(),d,gain.data_ptr(),8.0,x.data_ptr(),d,mp2.data_ptr(),I,yb.data_ptr(),d,gain.data_ptr(),8.0,1e-5,L,d,0,C.byref(nxq),S)
print(f"layer tail + qkv: {t(tail2):7.1f} us")
st.zero_(); lib.ttv_debug_stamps(st.data_ptr()); tail2(); torch.cuda.synchronize(); lib.ttv_debug_stamps(None)
s=st.cpu().tolist()
for role,name in ((0,"A"),(1,"B")):
    v=[q for q in s[role*64:role*64+64] if q]
    print(name, "n=",len(v), "deltas:", [v[i+1]-v[i] for i in range(len(v)-1)][-12:])
