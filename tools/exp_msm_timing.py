import sys, time, ctypes as C, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
import kzg_rust_amd as kz
from synth import random_blob
L=kz.kzg.lib()
G='tests/golden/'
g1=open(G+'trusted_setup_g1.bin','rb').read(); g2=open(G+'trusted_setup_g2.bin','rb').read()
s=kz.Kzg.load_trusted_setup([g1[48*i:48*i+48] for i in range(4096)],[g2[96*i:96*i+96] for i in range(65)])
s.set_kernel_timing(True)
def run(tag, blobs):
    n=len(blobs)
    t=torch.frombuffer(bytearray(b''.join(blobs)),dtype=torch.uint8).cuda()
    out=C.create_string_buffer(48*n); st=(C.c_int*n)()
    for rep in range(2):
        torch.cuda.synchronize(); t0=time.time()
        rc=L.kzg355_blob_to_kzg_commitment_many_device(out,st,t.data_ptr(),n,s.handle)
        dt=time.time()-t0
        print(tag, 'n',n,'rep',rep,'rc',rc,'%.1f ms'%(dt*1e3), 'bucket %.1f ms'%s.last_kernel_ms('msm_bucket'), 'fin %.2f'%s.last_kernel_ms('msm_finalize'), 'dig %.2f'%s.last_kernel_ms('digits'), flush=True)
n=256
bench=[random_blob(i) for i in range(n)]
run('bench-style(top byte 0)', bench)
rng=np.random.default_rng(1)
full=[]
for i in range(n):
    a=bytearray(rng.integers(0,256,131072,dtype=np.uint8).tobytes())
    a[0::32]=bytes(rng.integers(0,0x73,4096,dtype=np.uint8))
    full.append(bytes(a))
run('full-range', full)
run('bench-style again', bench)
run('n=1', bench[:1]); run('n=16', bench[:16])
