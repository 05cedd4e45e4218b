"""to_qkv + rotary at the benchmark shape (36 864 rows, width 256) through the kernels that could run it: k_qkv256 (default), k_gemm_k256, and -
ttv_debug_set bit 14 - the general-K kernels with 256 x 256 or 128-feature tiles.  GPU box only."""
import os, sys, torch
sys.path.insert(0, '.')
from titok_video_amd import _lib
from titok_video_amd.plan import BatchPlan
DEV=torch.device("cuda:0"); lib=_lib.lib(); ST=_lib.stream_ptr(DEV); bf=torch.bfloat16; code=_lib.dtype_code(bf)
plan=BatchPlan([(16,128,128)]*32,[128]*32,(4,8,8),DEV)
M,d,gq=plan.total_rows,256,128
g=torch.Generator().manual_seed(0)
x=(torch.randn(M,d,generator=g)).to(bf).to(DEV); w=(torch.randn(2*d+2*gq,d,generator=g)*d**-0.5).to(bf).to(DEV)
y=torch.empty(M,2*d+2*gq,dtype=bf,device=DEV)
flops=2*M*(2*d+2*gq)*d
for bit,tag in ((0,"default (k_qkv256)"),(32768,"k_gemm_k256"),(16384|512,"general-K 256x256"),(16384|1024,"general-K 128/160-token tiles")):
    lib.ttv_debug_set(bit)
    call=lambda: lib.ttv_linear_qkv_rope(x.data_ptr(),d,w.data_ptr(),d,y.data_ptr(),2*d+2*gq,M,d,gq,plan.rope_cs.data_ptr(),code,ST)
    for _ in range(3): _lib.check(call(),"qkv")
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/20
    print(f"{tag:22s} {us:7.1f} us {flops/us/1e6:6.0f} TFLOP/s  checksum {float(y.float().abs().sum()):.4e}")
lib.ttv_debug_set(0)
