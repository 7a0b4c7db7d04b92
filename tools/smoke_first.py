import sys, time, json
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import kzg_rust_amd as kz
G='tests/golden/'
g1=open(G+'trusted_setup_g1.bin','rb').read(); g2=open(G+'trusted_setup_g2.bin','rb').read()
t=time.time()
s=kz.Kzg.load_trusted_setup([g1[48*i:48*i+48] for i in range(4096)],[g2[96*i:96*i+96] for i in range(65)])
print('setup ok', time.time()-t, flush=True)
V=json.load(open(G+'vectors.json'))['functions']
blobs=[open(G+f'blobs/blob_{i}.bin','rb').read() for i in range(10)]
c=[x for x in V['blob_to_kzg_commitment'] if x['output']][0]
t=time.time(); r=kz.Kzg.blob_to_kzg_commitment(kz.Blob(blobs[c['input']['blob']['blob']]), s); print('commit', time.time()-t, r.to_bytes().hex()==c['output'][2:], flush=True)
c=[x for x in V['verify_blob_kzg_proof_batch'] if x['output'] is True and len(x['input']['blobs'])==3][0]
t=time.time(); r=kz.Kzg.verify_blob_kzg_proof_batch([kz.Blob(blobs[b['blob']]) for b in c['input']['blobs']],[kz.KzgCommitment.from_hex(x) for x in c['input']['commitments']],[kz.KzgProof.from_hex(x) for x in c['input']['proofs']], s); print('verify3', time.time()-t, r, flush=True)
