//! The reference's criterion bench (benches/kzg_benches.rs:25-127) over this crate: the five single-operation benches and the group
//! `verify_blob_kzg_proof_batch/{1,2,4,8,16,32,64}` with `Throughput::Elements(count)` -- blobs per second, the unit of BASELINE.json's
//! metric.  Inputs as in the reference (bench:7-44): random bytes with byte 0 of every field element cleared, honest commitments and
//! proofs made with the library outside the timed region.  Also benches the `*_many` extension over 64 batches per call.
//! KZG_RUST_TRUSTED_SETUP names the trusted setup file (default ../../reference/trusted_setup.txt).
use criterion::{criterion_group, criterion_main, BatchSize, BenchmarkId, Criterion, Throughput};
use kzg_rust::*;
use rand::Rng;

const MAX_COUNT: usize = 64;

fn random_field_element<R: Rng>(rng: &mut R) -> Bytes32 {
    let mut raw = [0u8; BYTES_PER_FIELD_ELEMENT];
    rng.fill(&mut raw[..]);
    raw[0] = 0; // below the BLS modulus
    raw.into()
}

fn random_blob<R: Rng>(rng: &mut R) -> Blob {
    let mut raw = vec![0u8; BYTES_PER_BLOB];
    rng.fill(&mut raw[..]);
    for element in raw.chunks_exact_mut(BYTES_PER_FIELD_ELEMENT) {
        element[0] = 0; // every field element canonical
    }
    Blob::from_bytes(&raw).unwrap()
}

fn benches(c: &mut Criterion) {
    let setup = std::env::var("KZG_RUST_TRUSTED_SETUP").unwrap_or_else(|_| "../../reference/trusted_setup.txt".into());
    let settings = Kzg::load_trusted_setup_file(setup).unwrap();
    let mut rng = rand::thread_rng();
    let blobs: Vec<Blob> = (0..MAX_COUNT).map(|_| random_blob(&mut rng)).collect();
    let commitments: Vec<KzgCommitment> = blobs.iter().map(|b| Kzg::blob_to_kzg_commitment(b, &settings).unwrap()).collect();
    let proofs: Vec<KzgProof> = blobs.iter().zip(&commitments).map(|(b, c)| Kzg::compute_blob_kzg_proof(b, c, &settings).unwrap()).collect();
    let z = random_field_element(&mut rng);
    let (point_proof, y) = Kzg::compute_kzg_proof(&blobs[0], &z, &settings).unwrap();

    c.bench_function("blob_to_kzg_commitment", |b| b.iter(|| Kzg::blob_to_kzg_commitment(&blobs[0], &settings)));
    c.bench_function("compute_kzg_proof", |b| b.iter(|| Kzg::compute_kzg_proof(&blobs[0], &z, &settings)));
    c.bench_function("compute_blob_kzg_proof", |b| b.iter(|| Kzg::compute_blob_kzg_proof(&blobs[0], &commitments[0], &settings)));
    c.bench_function("verify_kzg_proof", |b| b.iter(|| Kzg::verify_kzg_proof(&commitments[0], &z, &y, &point_proof, &settings)));
    c.bench_function("verify_blob_kzg_proof", |b| b.iter(|| Kzg::verify_blob_kzg_proof(&blobs[0], &commitments[0], &proofs[0], &settings)));

    let mut group = c.benchmark_group("verify_blob_kzg_proof_batch");
    for count in [1usize, 2, 4, 8, 16, 32, 64] {
        group.throughput(Throughput::Elements(count as u64));
        group.bench_with_input(BenchmarkId::from_parameter(count), &count, |b, &count| {
            b.iter_batched_ref(
                || (blobs[..count].to_vec(), commitments[..count].to_vec(), proofs[..count].to_vec()),
                |(bs, cs, ps)| assert!(Kzg::verify_blob_kzg_proof_batch(bs, cs, ps, &settings).unwrap()),
                BatchSize::LargeInput,
            );
        });
    }
    group.finish();

    // the throughput extension: 64 batches of 64 blobs per call (the same 64 blobs repeated)
    let mut many = c.benchmark_group("verify_blob_kzg_proof_batch_many");
    let groups = 64usize;
    let (mb, mc, mp): (Vec<Blob>, Vec<KzgCommitment>, Vec<KzgProof>) = (
        (0..groups).flat_map(|_| blobs.iter().cloned()).collect(),
        (0..groups).flat_map(|_| commitments.iter().copied()).collect(),
        (0..groups).flat_map(|_| proofs.iter().copied()).collect(),
    );
    many.throughput(Throughput::Elements((groups * MAX_COUNT) as u64));
    many.bench_function("64x64", |b| {
        b.iter(|| assert!(Kzg::verify_blob_kzg_proof_batch_many(&mb, &mc, &mp, MAX_COUNT, &settings).unwrap().into_iter().all(|r| r.unwrap())))
    });
    many.finish();
}

criterion_group!(kzg, benches);
criterion_main!(kzg);
