import sys, time; sys.path.insert(0,'python-soxr_amd'); sys.path.insert(0,'.')
import numpy as np, soxr_amd as soxr
from oracle import oracle
import _provider  # noqa: F401  (port mode on the product's bank: arithmetic-order check)
rng=np.random.default_rng(0)
x=(rng.standard_normal(20_000_000)*0.25).astype(np.float32)
t=time.time(); y=soxr.resample(x,48000,44100,'HQ'); dt=time.time()-t
print(len(y), int(len(x)*44100/48000+.5), "%.1f ms"%(dt*1e3))
pl=oracle.plan(48000,44100,'HQ')
for k0 in (0, 15_414_000, 15_414_500, len(y)-1000):   # around the 2^24-frame seam: 16777216*147/160 = 15414067
    want=oracle.resample_channel(pl,x,'port_f32',k0=k0,n_out=1000)
    print(k0, np.array_equal(y[k0:k0+1000],want))
xi=(rng.standard_normal((17_000_000,2))*5000).astype(np.int16)
t=time.time(); yi=soxr.resample(xi,44100,16000,'VHQ'); dt=time.time()-t
print(yi.shape, "%.1f ms"%(dt*1e3))
pl=oracle.plan(44100,16000,'VHQ')
k0=int(16777216*160/441)-500
v=oracle.resample_channel(pl,xi[:,1].astype(np.float32),'port_f32',k0=k0,n_out=1000)
q,_=oracle.quantize(v,np.int16,channel=1,k0=k0)
print(np.array_equal(yi[k0:k0+1000,1],q))
