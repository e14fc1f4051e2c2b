"""Probe which rocFFT plan geometries the slab FFT needs are accepted (GPU box only)."""
import sys; sys.path.insert(0,'.')
import torch
from pmesh_amd import backend, _abi
be=backend.get()
def tryplan(name,*a):
    try:
        p=be.fft_create(*a); print('ok  ',name); be.fft_destroy(p)
    except Exception as e: print('FAIL',name)
R2C,C2R=_abi.PMX_FFT_R2C,_abi.PMX_FFT_C2R
for es in (4,8):
  for n in (8,16,32,64,128,256,512):
    nc=n//2+1
    tryplan('c2r 2d [%d,%d] es%d padded oop'%(n,n,es), C2R, es, [n,n],[nc,1],n*nc,[2*nc,1],n*2*nc,3,1.0,False)
    tryplan('c2r 2d [%d,%d] es%d dense oop'%(n,n,es), C2R, es, [n,n],[nc,1],n*nc,[n,1],n*n,3,1.0,False)
    tryplan('r2c 2d [%d,%d] es%d padded oop'%(n,n,es), R2C, es, [n,n],[2*nc,1],n*2*nc,[nc,1],n*nc,3,1.0,False)
