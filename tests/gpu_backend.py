"""Adapter: the product (kzg_rust_amd.Kzg over the C ABI of libkzg355.so) behind the bytes-in/bytes-out
interface tests/vector_harness.py drives.  Used only by the -m gpu tests."""
from kzg_rust_amd import Blob, Bytes32, Kzg, KzgCommitment, KzgProof


class ProductBackend:
    def blob_to_kzg_commitment(self, blob, s):
        return Kzg.blob_to_kzg_commitment(Blob(blob), s).to_bytes()

    def compute_kzg_proof(self, blob, z, s):
        p, y = Kzg.compute_kzg_proof(Blob(blob), Bytes32(z), s)
        return p.to_bytes(), y.to_bytes()

    def compute_blob_kzg_proof(self, blob, c, s):
        return Kzg.compute_blob_kzg_proof(Blob(blob), KzgCommitment(c), s).to_bytes()

    def verify_kzg_proof(self, c, z, y, p, s):
        return Kzg.verify_kzg_proof(KzgCommitment(c), Bytes32(z), Bytes32(y), KzgProof(p), s)

    def verify_blob_kzg_proof(self, blob, c, p, s):
        return Kzg.verify_blob_kzg_proof(Blob(blob), KzgCommitment(c), KzgProof(p), s)

    def verify_blob_kzg_proof_batch(self, blobs, cs, ps, s):
        return Kzg.verify_blob_kzg_proof_batch([Blob(b) for b in blobs], [KzgCommitment(c) for c in cs], [KzgProof(p) for p in ps], s)
