import os, sys, torch
sys.path.insert(0, os.getcwd())
from titok_video_amd import _lib
from titok_video_amd.plan import BatchPlan
lib=_lib.lib(); DEV=torch.device("cuda:0"); S=_lib.stream_ptr(DEV)
plan = BatchPlan([(16, 128, 128)] * 32, [128] * 32, (4, 8, 8), DEV)
L=plan.total_rows; bf=torch.bfloat16
qkv=(torch.randn(L,768,device=DEV)).to(bf); ao=torch.empty(L,256,dtype=bf,device=DEV)
tab=plan.attention_table(4,2)
print("table entries", tab.shape[0], "valid", int((tab[:,0]>=0).sum()))
def t(n,it=30):
    fn=lambda: lib.ttv_attention(qkv.data_ptr(),768,ao.data_ptr(),256,plan.cu_dev.data_ptr(),tab.data_ptr(),n,4,2,64,1,0,S)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/it
for n in (256,512,768,896,1024,1152):
    print(n, f"{t(n):.1f} us")
