"""Time sd_unet_finalize (host-side packing + upload) of an SD-1.5-width UNet; SD_PACK_TIMING=1 prints the packer's share.
Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, ctypes as C, torch, os
from sonicdiffusionbayeslab_amd import _lib
from sonicdiffusionbayeslab_amd.unet import _c_config
from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict, param_shapes
lib=_lib.load()
cfg=UNetConfig(sample_size=16)
sd=make_synthetic_state_dict(cfg, seed=5)
h=C.c_void_p()
_lib.check(lib.sd_unet_create(C.byref(_c_config(cfg,"bf16")), C.byref(h)))
for name,shape in param_shapes(cfg):
    t=sd[name].float().contiguous(); _lib.check(lib.sd_unet_load_param(h,name.encode(),t.data_ptr(),t.numel()))
t0=time.time(); rc=lib.sd_unet_finalize(h); print('finalize', round(time.time()-t0,1), rc)
