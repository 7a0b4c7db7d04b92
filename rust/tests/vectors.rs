//! The reference's vector loop (src/lib.rs:30-203) against the engine: every `tests/<fn>/small/<case>/data.yaml` of the
//! reference checkout pointed to by KZG_RUST_VECTORS (default ../../reference/tests), `null` <=> any `Err`.
//! `cargo test --release` on a box with rustc, a HIP device and libkzg355.so built.
use kzg_rust::*;
use serde::Deserialize;
use std::path::PathBuf;

fn vectors_root() -> PathBuf {
    std::env::var("KZG_RUST_VECTORS").map(PathBuf::from).unwrap_or_else(|_| PathBuf::from("../../reference/tests"))
}
fn settings() -> KzgSettings {
    let f = std::env::var("KZG_RUST_TRUSTED_SETUP").unwrap_or_else(|_| "../../reference/trusted_setup.txt".into());
    Kzg::load_trusted_setup_file(f).expect("trusted setup")
}
fn cases(name: &str) -> Vec<PathBuf> {
    let pat = vectors_root().join(name).join("*/*/data.yaml");
    glob::glob(pat.to_str().unwrap()).unwrap().map(|p| p.unwrap()).collect()
}

#[derive(Deserialize)]
struct CommitIn { blob: String }
#[derive(Deserialize)]
struct CommitCase { input: CommitIn, output: Option<String> }

#[test]
fn blob_to_kzg_commitment_vectors() {
    let s = settings();
    let files = cases("blob_to_kzg_commitment");
    assert!(!files.is_empty());
    for f in files {
        let t: CommitCase = serde_yaml::from_str(&std::fs::read_to_string(&f).unwrap()).unwrap();
        let blob = match Blob::from_hex(&t.input.blob) { Ok(b) => b, Err(_) => { assert!(t.output.is_none()); continue; } };
        match Kzg::blob_to_kzg_commitment(&blob, &s) {
            Ok(c) => assert_eq!(c, KzgCommitment::from_hex(t.output.as_ref().unwrap()).unwrap(), "{:?}", f),
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}

#[derive(Deserialize)]
struct BatchIn { blobs: Vec<String>, commitments: Vec<String>, proofs: Vec<String> }
#[derive(Deserialize)]
struct BatchCase { input: BatchIn, output: Option<bool> }

#[test]
fn verify_blob_kzg_proof_batch_vectors() {
    let s = settings();
    let files = cases("verify_blob_kzg_proof_batch");
    assert!(!files.is_empty());
    for f in files {
        let t: BatchCase = serde_yaml::from_str(&std::fs::read_to_string(&f).unwrap()).unwrap();
        let blobs: Result<Vec<_>, _> = t.input.blobs.iter().map(|h| Blob::from_hex(h)).collect();
        let cs: Result<Vec<_>, _> = t.input.commitments.iter().map(|h| KzgCommitment::from_hex(h)).collect();
        let ps: Result<Vec<_>, _> = t.input.proofs.iter().map(|h| KzgProof::from_hex(h)).collect();
        let (blobs, cs, ps) = match (blobs, cs, ps) { (Ok(b), Ok(c), Ok(p)) => (b, c, p), _ => { assert!(t.output.is_none()); continue; } };
        match Kzg::verify_blob_kzg_proof_batch(&blobs, &cs, &ps, &s) {
            Ok(v) => assert_eq!(Some(v), t.output, "{:?}", f),
            Err(_) => assert!(t.output.is_none(), "{:?}", f),
        }
    }
}
