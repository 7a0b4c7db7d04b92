//! The reference's vector loop (src/lib.rs:30-203, schemas src/test_formats/*.rs) against the engine: every
//! `tests/<function>/small/<case>/data.yaml` of the reference checkout pointed to by KZG_RUST_VECTORS (default ../../reference/tests).
//! Pass rule, the same for all six functions (e.g. src/lib.rs:189-201): an input that fails to PARSE (hex / length) => the expected
//! output must be null; otherwise `Ok(v)` => `v == output`, `Err(_)` => output is null.  `cargo test --release` on a box with
//! rustc, a HIP device and libkzg355.so built.
use kzg_rust::*;
use serde::de::DeserializeOwned;
use serde::Deserialize;
use std::path::PathBuf;

fn vectors_root() -> PathBuf {
    std::env::var("KZG_RUST_VECTORS").map(PathBuf::from).unwrap_or_else(|_| PathBuf::from("../../reference/tests"))
}
fn settings() -> KzgSettings {
    let f = std::env::var("KZG_RUST_TRUSTED_SETUP").unwrap_or_else(|_| "../../reference/trusted_setup.txt".into());
    Kzg::load_trusted_setup_file(f).expect("trusted setup")
}
/// (file, parsed case) for every data.yaml of one function; `expected_cases` guards against an empty glob
fn cases<T: DeserializeOwned>(function: &str, expected_cases: usize) -> Vec<(PathBuf, T)> {
    let pattern = vectors_root().join(function).join("*/*/data.yaml");
    let files: Vec<PathBuf> = glob::glob(pattern.to_str().unwrap()).unwrap().map(|p| p.unwrap()).collect();
    assert_eq!(files.len(), expected_cases, "{}: vector count", function);
    files.into_iter().map(|f| { let t = serde_yaml::from_str(&std::fs::read_to_string(&f).unwrap()).unwrap(); (f, t) }).collect()
}
/// the newtype behind a commitment / proof string: both are `Bytes48` on the wire
fn b48(h: &str) -> Result<Bytes48, Error> { Bytes48::from_hex(h) }

#[derive(Deserialize)]
struct Case<I, O> { input: I, output: Option<O> }

#[derive(Deserialize)]
struct CommitIn { blob: String }

#[test]
fn blob_to_kzg_commitment_vectors() {
    let s = settings();
    for (f, t) in cases::<Case<CommitIn, String>>("blob_to_kzg_commitment", 10) {
        let Ok(blob) = Blob::from_hex(&t.input.blob) else { assert!(t.output.is_none(), "{:?}", f); continue; };
        match Kzg::blob_to_kzg_commitment(&blob, &s) {
            Ok(c) => assert_eq!(c.to_bytes(), *b48(t.output.as_ref().unwrap()).unwrap(), "{:?}", f),
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}

#[derive(Deserialize)]
struct ProofIn { blob: String, z: String }

#[test]
fn compute_kzg_proof_vectors() {
    let s = settings();
    for (f, t) in cases::<Case<ProofIn, (String, String)>>("compute_kzg_proof", 46) {
        let (Ok(blob), Ok(z)) = (Blob::from_hex(&t.input.blob), Bytes32::from_hex(&t.input.z)) else { assert!(t.output.is_none(), "{:?}", f); continue; };
        match Kzg::compute_kzg_proof(&blob, &z, &s) {
            Ok((proof, y)) => {
                let (want_proof, want_y) = t.output.as_ref().unwrap();
                assert_eq!(proof.to_bytes(), *b48(want_proof).unwrap(), "{:?}", f);
                assert_eq!(*y, *Bytes32::from_hex(want_y).unwrap(), "{:?}", f);
            }
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}

#[derive(Deserialize)]
struct BlobProofIn { blob: String, commitment: String }

#[test]
fn compute_blob_kzg_proof_vectors() {
    let s = settings();
    for (f, t) in cases::<Case<BlobProofIn, String>>("compute_blob_kzg_proof", 14) {
        let (Ok(blob), Ok(c)) = (Blob::from_hex(&t.input.blob), b48(&t.input.commitment)) else { assert!(t.output.is_none(), "{:?}", f); continue; };
        match Kzg::compute_blob_kzg_proof(&blob, &KzgCommitment(c), &s) {
            Ok(proof) => assert_eq!(proof.to_bytes(), *b48(t.output.as_ref().unwrap()).unwrap(), "{:?}", f),
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}

#[derive(Deserialize)]
struct VerifyIn { commitment: String, z: String, y: String, proof: String }

#[test]
fn verify_kzg_proof_vectors() {
    let s = settings();
    for (f, t) in cases::<Case<VerifyIn, bool>>("verify_kzg_proof", 92) {
        let (Ok(c), Ok(z), Ok(y), Ok(p)) = (b48(&t.input.commitment), Bytes32::from_hex(&t.input.z), Bytes32::from_hex(&t.input.y), b48(&t.input.proof)) else {
            assert!(t.output.is_none(), "{:?}", f);
            continue;
        };
        match Kzg::verify_kzg_proof(&KzgCommitment(c), &z, &y, &KzgProof(p), &s) {
            Ok(v) => assert_eq!(Some(v), t.output, "{:?}", f),
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}

#[derive(Deserialize)]
struct VerifyBlobIn { blob: String, commitment: String, proof: String }

#[test]
fn verify_blob_kzg_proof_vectors() {
    let s = settings();
    for (f, t) in cases::<Case<VerifyBlobIn, bool>>("verify_blob_kzg_proof", 24) {
        let (Ok(blob), Ok(c), Ok(p)) = (Blob::from_hex(&t.input.blob), b48(&t.input.commitment), b48(&t.input.proof)) else {
            assert!(t.output.is_none(), "{:?}", f);
            continue;
        };
        match Kzg::verify_blob_kzg_proof(&blob, &KzgCommitment(c), &KzgProof(p), &s) {
            Ok(v) => assert_eq!(Some(v), t.output, "{:?}", f),
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}

#[derive(Deserialize)]
struct BatchIn { blobs: Vec<String>, commitments: Vec<String>, proofs: Vec<String> }

#[test]
fn verify_blob_kzg_proof_batch_vectors() {
    let s = settings();
    for (f, t) in cases::<Case<BatchIn, bool>>("verify_blob_kzg_proof_batch", 22) {
        let blobs: Result<Vec<_>, _> = t.input.blobs.iter().map(|h| Blob::from_hex(h)).collect();
        let cs: Result<Vec<_>, _> = t.input.commitments.iter().map(|h| b48(h).map(KzgCommitment)).collect();
        let ps: Result<Vec<_>, _> = t.input.proofs.iter().map(|h| b48(h).map(KzgProof)).collect();
        let (Ok(blobs), Ok(cs), Ok(ps)) = (blobs, cs, ps) else { assert!(t.output.is_none(), "{:?}", f); continue; };
        match Kzg::verify_blob_kzg_proof_batch(&blobs, &cs, &ps, &s) {
            Ok(v) => assert_eq!(Some(v), t.output, "{:?}", f),
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}

/// The `*_many` extensions give, per unit, what the single calls give.
#[test]
fn many_extensions_agree_with_single_calls() {
    let s = settings();
    let valid: Vec<Blob> = cases::<Case<CommitIn, String>>("blob_to_kzg_commitment", 10)
        .into_iter()
        .filter(|(_, t)| t.output.is_some())
        .map(|(_, t)| Blob::from_hex(&t.input.blob).unwrap())
        .collect();
    let cs: Vec<KzgCommitment> = Kzg::blob_to_kzg_commitment_many(&valid, &s).unwrap().into_iter().map(|r| r.unwrap()).collect();
    let ps: Vec<KzgProof> = Kzg::compute_blob_kzg_proof_many(&valid, &cs, &s).unwrap().into_iter().map(|r| r.unwrap()).collect();
    for (i, b) in valid.iter().enumerate() {
        assert_eq!(cs[i], Kzg::blob_to_kzg_commitment(b, &s).unwrap());
        assert_eq!(ps[i], Kzg::compute_blob_kzg_proof(b, &cs[i], &s).unwrap());
    }
    assert!(Kzg::verify_blob_kzg_proof_batch(&valid, &cs, &ps, &s).unwrap());
    let per_blob = Kzg::verify_blob_kzg_proof_batch_many(&valid, &cs, &ps, 1, &s).unwrap();
    assert!(per_blob.into_iter().all(|r| r.unwrap()));
}

/// The single-proof functions through their `*_many` forms: every parsable `verify_kzg_proof` vector in ONE call, every parsable
/// `compute_kzg_proof` vector in ONE call -- a unit's `Err` is that vector's null output (pass rule of src/lib.rs:189-201 per unit).
#[test]
fn single_proof_vectors_through_the_many_forms() {
    let s = settings();
    let (mut cs, mut zs, mut ys, mut ps, mut want) = (vec![], vec![], vec![], vec![], vec![]);
    for (f, t) in cases::<Case<VerifyIn, bool>>("verify_kzg_proof", 92) {
        let (Ok(c), Ok(z), Ok(y), Ok(p)) = (b48(&t.input.commitment), Bytes32::from_hex(&t.input.z), Bytes32::from_hex(&t.input.y), b48(&t.input.proof)) else {
            assert!(t.output.is_none(), "{:?}", f);
            continue;
        };
        cs.push(KzgCommitment(c)); zs.push(z); ys.push(y); ps.push(KzgProof(p)); want.push((f, t.output));
    }
    let got = Kzg::verify_kzg_proof_many(&cs, &zs, &ys, &ps, &s).unwrap();
    assert_eq!(got.len(), want.len());
    for (r, (f, w)) in got.into_iter().zip(want) {
        match r { Ok(v) => assert_eq!(Some(v), w, "{:?}", f), Err(_) => assert!(w.is_none(), "{:?}", f) }
    }
    let (mut blobs, mut zs, mut want) = (vec![], vec![], vec![]);
    for (f, t) in cases::<Case<ProofIn, (String, String)>>("compute_kzg_proof", 46) {
        let (Ok(blob), Ok(z)) = (Blob::from_hex(&t.input.blob), Bytes32::from_hex(&t.input.z)) else { assert!(t.output.is_none(), "{:?}", f); continue; };
        blobs.push(blob); zs.push(z); want.push((f, t.output));
    }
    let got = Kzg::compute_kzg_proof_many(&blobs, &zs, &s).unwrap();
    for (r, (f, w)) in got.into_iter().zip(want) {
        match r {
            Ok((proof, y)) => {
                let (want_proof, want_y) = w.as_ref().unwrap();
                assert_eq!(proof.to_bytes(), *b48(want_proof).unwrap(), "{:?}", f);
                assert_eq!(*y, *Bytes32::from_hex(want_y).unwrap(), "{:?}", f);
            }
            Err(_) => assert!(w.is_none(), "{:?}", f),
        }
    }
}
