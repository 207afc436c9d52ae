import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package()
n=1<<24
rng=np.random.default_rng(0)
x=(rng.standard_normal(n)+1j*rng.standard_normal(n)).astype(np.complex64); y=np.empty_like(x)
fft=pkg.clFFT(4096,pkg.CLFFT_FORWARD,np.blackman(4096).astype(np.float32),pkg.DTYPE_COMPLEX,1,2,0,0,0,1,True)
for i in range(4):
    t0=time.perf_counter(); fft.work(n//4096,[x],[y]); print("call %.3f ms"%((time.perf_counter()-t0)*1e3))
# plain memcpy rates of numpy on this host
t0=time.perf_counter(); 
for _ in range(5): np.copyto(y,x)
print("np.copyto 128 MiB: %.1f GB/s (1 thread)"%(5*x.nbytes/(time.perf_counter()-t0)/1e9))
