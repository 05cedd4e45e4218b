"""MI355X-native TiTok-Video tokenizer hot path (encode -> FSQ -> decode).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); the compute path is
hand-written HIP for gfx950 behind the C-ABI in include/titok_hip.h (libtitok_hip.so, loaded lazily by
`titok_video_amd._lib`).  There is no CPU or eager-PyTorch fallback: ops raise if the library is missing.
"""
__version__ = "0.1.0"
