import sys; sys.path.insert(0,'.')
import torch
from pmesh_amd import backend, _abi
be=backend.get()
def tryplan(name,*a):
    try:
        p=be.fft_create(*a); print('ok  ',name); be.fft_destroy(p)
    except Exception as e: print('FAIL',name,str(e)[:100])
# kind, elsize, n, istride, idist, ostride, odist, batch, scale, inplace
tryplan('r2c 2d batched oop', _abi.PMX_FFT_R2C, 8, [12,10],[12,1],144,[6,1],72,4,1.0,False)
tryplan('c2c col inplace dist1', _abi.PMX_FFT_C2C_FWD, 8, [8],[36],1,[36],1,36,1.0,True)
tryplan('c2c col oop dist1', _abi.PMX_FFT_C2C_FWD, 8, [8],[36],1,[36],1,36,1.0,False)
tryplan('c2c col inplace dist1 512', _abi.PMX_FFT_C2C_FWD, 8, [512],[64*257],1,[64*257],1,64*257,1.0,True)
tryplan('r2c 1d batched (2-d mesh)', _abi.PMX_FFT_R2C, 8, [7],[1],8,[1],4,5,1.0,False)
tryplan('c2c len9 stride4', _abi.PMX_FFT_C2C_FWD, 8, [9],[4],1,[4],1,4,1.0,True)
tryplan('c2r 2d batched oop', _abi.PMX_FFT_C2R, 8, [12,10],[6,1],72,[12,1],144,4,1.0,False)
tryplan('r2c 2d 512 batched', _abi.PMX_FFT_R2C, 8, [512,512],[514,1],512*514,[257,1],512*257,64,1.0/512**3,False)
tryplan('c2c nb=1', _abi.PMX_FFT_C2C_FWD, 8, [8],[1],1,[1],1,1,1.0,True)
