//! The public surface of the reference crate (src/kzg.rs:10-22, 28-79, 101-204, 983-1079) with every body forwarded through
//! `ffi` to libkzg355.so.  The byte newtypes raise their length / hex errors on this side, before any FFI call, exactly as the
//! reference's do; the engine decides everything else (Ok vs Err, the returned bytes, the verdicts).
use crate::consts::*;
use crate::ffi;
use std::ffi::CString;
use std::ops::{Deref, DerefMut};
use std::path::Path;

#[derive(Debug)]
pub enum Error {
    /// The supplied data is invalid in some way.
    BadArgs(String),
    /// Internal error - this should never occur (also: no usable HIP device, allocation failure; there is no CPU fallback).
    InternalError,
    /// The provided bytes are of incorrect length.
    InvalidBytesLength(String),
    /// Error when converting from hex to bytes.
    InvalidHexFormat(String),
    /// The provided trusted setup params are invalid.
    InvalidTrustedSetup(String),
}

fn check(rc: i32, what: &str) -> Result<(), Error> {
    match rc {
        ffi::KZG355_OK => Ok(()),
        ffi::KZG355_BADARGS => Err(Error::BadArgs(what.into())),
        ffi::KZG355_INVALID_BYTES_LENGTH => Err(Error::InvalidBytesLength(what.into())),
        ffi::KZG355_INVALID_HEX => Err(Error::InvalidHexFormat(what.into())),
        ffi::KZG355_INVALID_TRUSTED_SETUP => Err(Error::InvalidTrustedSetup(what.into())),
        _ => Err(Error::InternalError), // INTERNAL, NO_DEVICE, NO_MEMORY, DEVICE_ERROR: no reference counterpart (there is no CPU fallback)
    }
}

/// Opaque handle to the device-resident trusted setup; replaces the `Vec`s of the reference's `KzgSettings`.
#[derive(Debug)]
pub struct KzgSettings {
    raw: *mut ffi::kzg355_settings,
}
// The handle is immutable after creation and the engine takes a private workspace + stream per call.
unsafe impl Send for KzgSettings {}
unsafe impl Sync for KzgSettings {}

impl Drop for KzgSettings {
    fn drop(&mut self) {
        unsafe { ffi::kzg355_free_trusted_setup(self.raw) }
    }
}

impl KzgSettings {
    pub fn load_trusted_setup(g1_bytes: Vec<[u8; BYTES_PER_G1]>, g2_bytes: Vec<[u8; BYTES_PER_G2]>) -> Result<Self, Error> {
        let g1: Vec<u8> = g1_bytes.iter().flatten().copied().collect();
        let g2: Vec<u8> = g2_bytes.iter().flatten().copied().collect();
        let mut raw = std::ptr::null_mut();
        check(
            unsafe { ffi::kzg355_load_trusted_setup(g1.as_ptr(), g1_bytes.len(), g2.as_ptr(), g2_bytes.len(), &mut raw) },
            "load_trusted_setup",
        )?;
        Ok(Self { raw })
    }

    /// Extension: the handle spans several GPUs of the node; calls spread their work inside the library.
    pub fn load_trusted_setup_on_devices(g1_bytes: Vec<[u8; BYTES_PER_G1]>, g2_bytes: Vec<[u8; BYTES_PER_G2]>, devices: &[i32]) -> Result<Self, Error> {
        let g1: Vec<u8> = g1_bytes.iter().flatten().copied().collect();
        let g2: Vec<u8> = g2_bytes.iter().flatten().copied().collect();
        let mut raw = std::ptr::null_mut();
        check(
            unsafe {
                ffi::kzg355_load_trusted_setup_devices(g1.as_ptr(), g1_bytes.len(), g2.as_ptr(), g2_bytes.len(), devices.as_ptr(), devices.len(), &mut raw)
            },
            "load_trusted_setup_on_devices",
        )?;
        Ok(Self { raw })
    }

    /// Extension: explicit engine options (`ffi::kzg355_options`, filled from `KzgSettings::default_options()`); no environment
    /// variable is read on this path.
    pub fn load_trusted_setup_with_options(
        g1_bytes: Vec<[u8; BYTES_PER_G1]>,
        g2_bytes: Vec<[u8; BYTES_PER_G2]>,
        devices: &[i32],
        options: &ffi::kzg355_options,
    ) -> Result<Self, Error> {
        let g1: Vec<u8> = g1_bytes.iter().flatten().copied().collect();
        let g2: Vec<u8> = g2_bytes.iter().flatten().copied().collect();
        let mut raw = std::ptr::null_mut();
        let devs = if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() };
        check(
            unsafe { ffi::kzg355_load_trusted_setup_ex(g1.as_ptr(), g1_bytes.len(), g2.as_ptr(), g2_bytes.len(), devs, devices.len(), options, &mut raw) },
            "load_trusted_setup_with_options",
        )?;
        Ok(Self { raw })
    }

    /// Defaults of every engine knob (`kzg355_options_default`).  E.g. `o.verify_only = 1` for a handle that never builds the MSM table,
    /// `o.msm_bits = 16` for the widest one (143.5 GB), `o.msm_eager = 1` to build it inside the load instead of on the first commitment.
    pub fn default_options() -> ffi::kzg355_options {
        let mut o = std::mem::MaybeUninit::<ffi::kzg355_options>::zeroed();
        unsafe {
            ffi::kzg355_options_default(o.as_mut_ptr());
            o.assume_init()
        }
    }

    /// FIELD_ELEMENTS_PER_BLOB of this handle (4096, or 4 for a minimal-preset setup).
    pub fn field_elements_per_blob(&self) -> usize {
        unsafe { ffi::kzg355_settings_field_elements_per_blob(self.raw) as usize }
    }

    pub fn load_trusted_setup_file<P: AsRef<Path>>(trusted_setup_file: P) -> Result<Self, Error> {
        let p = trusted_setup_file.as_ref().to_str().ok_or_else(|| Error::InvalidTrustedSetup("path is not UTF-8".into()))?;
        let c = CString::new(p).map_err(|_| Error::InvalidTrustedSetup("path contains NUL".into()))?;
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::kzg355_load_trusted_setup_file(c.as_ptr(), &mut raw) }, "load_trusted_setup_file")?;
        Ok(Self { raw })
    }
}

pub(crate) fn hex_to_bytes(hex_str: &str) -> Result<Vec<u8>, Error> {
    let trimmed = hex_str.strip_prefix("0x").unwrap_or(hex_str);
    hex::decode(trimmed).map_err(|e| Error::InvalidHexFormat(format!("Failed to decode hex: {}", e)))
}

macro_rules! fixed_bytes {
    ($name:ident, $n:expr, $len_err:expr) => {
        #[derive(Debug, Copy, Clone, PartialEq)]
        pub struct $name {
            pub(crate) bytes: [u8; $n],
        }
        impl $name {
            pub fn from_bytes(b: &[u8]) -> Result<Self, Error> {
                if b.len() != $n {
                    return Err($len_err(format!("Invalid byte length. Expected {} got {}", $n, b.len())));
                }
                let mut bytes = [0u8; $n];
                bytes.copy_from_slice(b);
                Ok(Self { bytes })
            }
            pub fn from_hex(hex_str: &str) -> Result<Self, Error> {
                Self::from_bytes(&hex_to_bytes(hex_str)?)
            }
        }
        impl Default for $name {
            fn default() -> Self {
                Self { bytes: [0u8; $n] }
            }
        }
        impl Deref for $name {
            type Target = [u8; $n];
            fn deref(&self) -> &Self::Target {
                &self.bytes
            }
        }
        impl From<[u8; $n]> for $name {
            fn from(bytes: [u8; $n]) -> Self {
                Self { bytes }
            }
        }
    };
}
fixed_bytes!(Bytes32, 32, Error::BadArgs); // the reference reports a wrong Bytes32 length as BadArgs (kzg.rs:107-112)
fixed_bytes!(Bytes48, 48, Error::InvalidBytesLength);

#[derive(Debug, Clone, PartialEq)]
pub struct Blob {
    bytes: Box<[u8; BYTES_PER_BLOB]>,
}
impl Blob {
    pub fn from_bytes(bytes: &[u8]) -> Result<Self, Error> {
        if bytes.len() != BYTES_PER_BLOB {
            return Err(Error::InvalidBytesLength(format!("Invalid byte length. Expected {} got {}", BYTES_PER_BLOB, bytes.len())));
        }
        let mut v = vec![0u8; BYTES_PER_BLOB].into_boxed_slice();
        v.copy_from_slice(bytes);
        let bytes: Box<[u8; BYTES_PER_BLOB]> = v.try_into().map_err(|_| Error::InternalError)?;
        Ok(Self { bytes })
    }
    pub fn from_hex(hex_str: &str) -> Result<Self, Error> {
        Self::from_bytes(&hex_to_bytes(hex_str)?)
    }
}
impl Deref for Blob {
    type Target = [u8; BYTES_PER_BLOB];
    fn deref(&self) -> &Self::Target {
        &self.bytes
    }
}
// the reference lets callers build and edit blobs in place (kzg.rs:222-228, 262-266: the bench fills arrays and converts them)
impl DerefMut for Blob {
    fn deref_mut(&mut self) -> &mut Self::Target {
        &mut self.bytes
    }
}
impl From<[u8; BYTES_PER_BLOB]> for Blob {
    fn from(value: [u8; BYTES_PER_BLOB]) -> Self {
        Self { bytes: Box::new(value) }
    }
}

macro_rules! g1_newtype {
    ($name:ident, $n:expr) => {
        #[derive(Debug, Copy, Clone, PartialEq)]
        pub struct $name(pub Bytes48);
        impl $name {
            pub fn from_hex(hex_str: &str) -> Result<Self, Error> {
                Ok(Self(Bytes48::from_bytes(&hex_to_bytes(hex_str)?)?))
            }
            pub fn to_bytes(self) -> [u8; $n] {
                self.0.bytes
            }
        }
        impl From<Bytes48> for $name {
            fn from(b: Bytes48) -> Self {
                Self(b)
            }
        }
        impl From<[u8; $n]> for $name {
            fn from(b: [u8; $n]) -> Self {
                Self(Bytes48 { bytes: b })
            }
        }
        impl Deref for $name {
            type Target = [u8; $n];
            fn deref(&self) -> &Self::Target {
                &self.0.bytes
            }
        }
    };
}
g1_newtype!(KzgCommitment, BYTES_PER_COMMITMENT);
g1_newtype!(KzgProof, BYTES_PER_PROOF);

pub struct Kzg;

impl Kzg {
    pub fn load_trusted_setup_file<P: AsRef<Path>>(trusted_setup_file: P) -> Result<KzgSettings, Error> {
        KzgSettings::load_trusted_setup_file(trusted_setup_file)
    }

    pub fn load_trusted_setup(g1_bytes: Vec<[u8; BYTES_PER_G1]>, g2_bytes: Vec<[u8; BYTES_PER_G2]>) -> Result<KzgSettings, Error> {
        KzgSettings::load_trusted_setup(g1_bytes, g2_bytes)
    }

    pub fn blob_to_kzg_commitment(blob: &Blob, s: &KzgSettings) -> Result<KzgCommitment, Error> {
        let mut out = [0u8; BYTES_PER_COMMITMENT];
        check(unsafe { ffi::kzg355_blob_to_kzg_commitment(out.as_mut_ptr(), blob.as_ptr(), s.raw) }, "blob_to_kzg_commitment")?;
        Ok(out.into())
    }

    pub fn compute_kzg_proof(blob: &Blob, z_bytes: &Bytes32, s: &KzgSettings) -> Result<(KzgProof, Bytes32), Error> {
        let (mut proof, mut y) = ([0u8; BYTES_PER_PROOF], [0u8; 32]);
        check(
            unsafe { ffi::kzg355_compute_kzg_proof(proof.as_mut_ptr(), y.as_mut_ptr(), blob.as_ptr(), z_bytes.as_ptr(), s.raw) },
            "compute_kzg_proof",
        )?;
        Ok((proof.into(), y.into()))
    }

    pub fn compute_blob_kzg_proof(blob: &Blob, commitment_bytes: &KzgCommitment, s: &KzgSettings) -> Result<KzgProof, Error> {
        let mut proof = [0u8; BYTES_PER_PROOF];
        check(
            unsafe { ffi::kzg355_compute_blob_kzg_proof(proof.as_mut_ptr(), blob.as_ptr(), commitment_bytes.as_ptr(), s.raw) },
            "compute_blob_kzg_proof",
        )?;
        Ok(proof.into())
    }

    pub fn verify_kzg_proof(
        commitment_bytes: &KzgCommitment,
        z_bytes: &Bytes32,
        y_bytes: &Bytes32,
        proof_bytes: &KzgProof,
        s: &KzgSettings,
    ) -> Result<bool, Error> {
        let mut ok = false;
        check(
            unsafe { ffi::kzg355_verify_kzg_proof(&mut ok, commitment_bytes.as_ptr(), z_bytes.as_ptr(), y_bytes.as_ptr(), proof_bytes.as_ptr(), s.raw) },
            "verify_kzg_proof",
        )?;
        Ok(ok)
    }

    pub fn verify_blob_kzg_proof(blob: &Blob, commitment_bytes: &KzgCommitment, proof_bytes: &KzgProof, s: &KzgSettings) -> Result<bool, Error> {
        let mut ok = false;
        check(
            unsafe { ffi::kzg355_verify_blob_kzg_proof(&mut ok, blob.as_ptr(), commitment_bytes.as_ptr(), proof_bytes.as_ptr(), s.raw) },
            "verify_blob_kzg_proof",
        )?;
        Ok(ok)
    }

    pub fn verify_blob_kzg_proof_batch(blobs: &[Blob], commitment_bytes: &[KzgCommitment], proof_bytes: &[KzgProof], s: &KzgSettings) -> Result<bool, Error> {
        let staged = stage_blobs(blobs);
        let c: Vec<u8> = commitment_bytes.iter().flat_map(|x| x.to_bytes()).collect();
        let p: Vec<u8> = proof_bytes.iter().flat_map(|x| x.to_bytes()).collect();
        let mut ok = false;
        // all three lengths cross the boundary: the mismatch check of the reference (kzg.rs:644-651) is the library's
        check(
            unsafe {
                ffi::kzg355_verify_blob_kzg_proof_batch(&mut ok, staged.as_ptr(), blobs.len(), c.as_ptr(), commitment_bytes.len(), p.as_ptr(), proof_bytes.len(), s.raw)
            },
            "verify_blob_kzg_proof_batch",
        )?;
        Ok(ok)
    }

    // ---- throughput extensions (no reference counterpart): many independent units per call, one set of kernel launches ----

    /// `blobs.len()` independent `blob_to_kzg_commitment` calls; one `Result` per blob.
    pub fn blob_to_kzg_commitment_many(blobs: &[Blob], s: &KzgSettings) -> Result<Vec<Result<KzgCommitment, Error>>, Error> {
        let n = blobs.len();
        let staged = stage_blobs(blobs);
        let mut out = vec![0u8; BYTES_PER_COMMITMENT * n.max(1)];
        let mut st = vec![0i32; n.max(1)];
        let rc = unsafe { ffi::kzg355_blob_to_kzg_commitment_many(out.as_mut_ptr(), st.as_mut_ptr(), staged.as_ptr(), n, s.raw) };
        whole_call(rc, &st[..n], "blob_to_kzg_commitment_many")?;
        Ok((0..n)
            .map(|i| check(st[i], "commit").map(|_| KzgCommitment::from(<[u8; BYTES_PER_COMMITMENT]>::try_from(&out[48 * i..48 * i + 48]).unwrap())))
            .collect())
    }

    /// `blobs.len()` independent `compute_blob_kzg_proof` calls.
    pub fn compute_blob_kzg_proof_many(blobs: &[Blob], commitments: &[KzgCommitment], s: &KzgSettings) -> Result<Vec<Result<KzgProof, Error>>, Error> {
        if blobs.len() != commitments.len() {
            return Err(Error::BadArgs("length mismatch".into()));
        }
        let n = blobs.len();
        let staged = stage_blobs(blobs);
        let c: Vec<u8> = commitments.iter().flat_map(|x| x.to_bytes()).collect();
        let mut out = vec![0u8; BYTES_PER_PROOF * n.max(1)];
        let mut st = vec![0i32; n.max(1)];
        let rc = unsafe { ffi::kzg355_compute_blob_kzg_proof_many(out.as_mut_ptr(), st.as_mut_ptr(), staged.as_ptr(), c.as_ptr(), n, s.raw) };
        whole_call(rc, &st[..n], "compute_blob_kzg_proof_many")?;
        Ok((0..n)
            .map(|i| check(st[i], "proof").map(|_| KzgProof::from(<[u8; BYTES_PER_PROOF]>::try_from(&out[48 * i..48 * i + 48]).unwrap())))
            .collect())
    }

    /// `groups` independent `verify_blob_kzg_proof_batch` calls of `n_per_group` blobs each (inputs group-major).
    pub fn verify_blob_kzg_proof_batch_many(
        blobs: &[Blob],
        commitments: &[KzgCommitment],
        proofs: &[KzgProof],
        n_per_group: usize,
        s: &KzgSettings,
    ) -> Result<Vec<Result<bool, Error>>, Error> {
        if blobs.len() != commitments.len() || blobs.len() != proofs.len() || n_per_group == 0 || blobs.len() % n_per_group != 0 {
            return Err(Error::BadArgs("length mismatch".into()));
        }
        let groups = blobs.len() / n_per_group;
        let staged = stage_blobs(blobs);
        let c: Vec<u8> = commitments.iter().flat_map(|x| x.to_bytes()).collect();
        let p: Vec<u8> = proofs.iter().flat_map(|x| x.to_bytes()).collect();
        let mut ok = vec![false; groups.max(1)];
        let mut st = vec![0i32; groups.max(1)];
        let rc = unsafe {
            ffi::kzg355_verify_blob_kzg_proof_batch_many(ok.as_mut_ptr(), st.as_mut_ptr(), staged.as_ptr(), c.as_ptr(), p.as_ptr(), n_per_group, groups, s.raw)
        };
        whole_call(rc, &st[..groups], "verify_blob_kzg_proof_batch_many")?;
        Ok((0..groups).map(|g| check(st[g], "verify").map(|_| ok[g])).collect())
    }

    /// `commitments.len()` independent `verify_kzg_proof` checks (one proof per call is what benches/kzg_benches.rs:70-81 times).
    pub fn verify_kzg_proof_many(
        commitments: &[KzgCommitment],
        zs: &[Bytes32],
        ys: &[Bytes32],
        proofs: &[KzgProof],
        s: &KzgSettings,
    ) -> Result<Vec<Result<bool, Error>>, Error> {
        let n = commitments.len();
        if zs.len() != n || ys.len() != n || proofs.len() != n {
            return Err(Error::BadArgs("length mismatch".into()));
        }
        let c: Vec<u8> = commitments.iter().flat_map(|x| x.to_bytes()).collect();
        let z: Vec<u8> = zs.iter().flat_map(|x| x.bytes).collect();
        let y: Vec<u8> = ys.iter().flat_map(|x| x.bytes).collect();
        let p: Vec<u8> = proofs.iter().flat_map(|x| x.to_bytes()).collect();
        let mut ok = vec![false; n.max(1)];
        let mut st = vec![0i32; n.max(1)];
        let rc = unsafe { ffi::kzg355_verify_kzg_proof_many(ok.as_mut_ptr(), st.as_mut_ptr(), c.as_ptr(), z.as_ptr(), y.as_ptr(), p.as_ptr(), n, s.raw) };
        whole_call(rc, &st[..n], "verify_kzg_proof_many")?;
        Ok((0..n).map(|i| check(st[i], "verify").map(|_| ok[i])).collect())
    }

    /// `blobs.len()` independent `verify_blob_kzg_proof` checks.
    pub fn verify_blob_kzg_proof_many(blobs: &[Blob], commitments: &[KzgCommitment], proofs: &[KzgProof], s: &KzgSettings) -> Result<Vec<Result<bool, Error>>, Error> {
        let n = blobs.len();
        if commitments.len() != n || proofs.len() != n {
            return Err(Error::BadArgs("length mismatch".into()));
        }
        let staged = stage_blobs(blobs);
        let c: Vec<u8> = commitments.iter().flat_map(|x| x.to_bytes()).collect();
        let p: Vec<u8> = proofs.iter().flat_map(|x| x.to_bytes()).collect();
        let mut ok = vec![false; n.max(1)];
        let mut st = vec![0i32; n.max(1)];
        let rc = unsafe { ffi::kzg355_verify_blob_kzg_proof_many(ok.as_mut_ptr(), st.as_mut_ptr(), staged.as_ptr(), c.as_ptr(), p.as_ptr(), n, s.raw) };
        whole_call(rc, &st[..n], "verify_blob_kzg_proof_many")?;
        Ok((0..n).map(|i| check(st[i], "verify").map(|_| ok[i])).collect())
    }

    /// `blobs.len()` independent `compute_kzg_proof` calls: proof and y = p(z) per blob at the caller's points.
    pub fn compute_kzg_proof_many(blobs: &[Blob], zs: &[Bytes32], s: &KzgSettings) -> Result<Vec<Result<(KzgProof, Bytes32), Error>>, Error> {
        let n = blobs.len();
        if zs.len() != n {
            return Err(Error::BadArgs("length mismatch".into()));
        }
        let staged = stage_blobs(blobs);
        let z: Vec<u8> = zs.iter().flat_map(|x| x.bytes).collect();
        let mut out = vec![0u8; BYTES_PER_PROOF * n.max(1)];
        let mut ys = vec![0u8; 32 * n.max(1)];
        let mut st = vec![0i32; n.max(1)];
        let rc = unsafe { ffi::kzg355_compute_kzg_proof_many(out.as_mut_ptr(), ys.as_mut_ptr(), st.as_mut_ptr(), staged.as_ptr(), z.as_ptr(), n, s.raw) };
        whole_call(rc, &st[..n], "compute_kzg_proof_many")?;
        Ok((0..n)
            .map(|i| {
                check(st[i], "proof").map(|_| {
                    (
                        KzgProof::from(<[u8; BYTES_PER_PROOF]>::try_from(&out[48 * i..48 * i + 48]).unwrap()),
                        Bytes32::from(<[u8; 32]>::try_from(&ys[32 * i..32 * i + 32]).unwrap()),
                    )
                })
            })
            .collect())
    }
}

/// `&[Blob]` is a slice of boxes: the blobs are not contiguous in memory, the ABI wants one buffer.
fn stage_blobs(blobs: &[Blob]) -> Vec<u8> {
    let mut staged = Vec::with_capacity(blobs.len() * BYTES_PER_BLOB);
    for b in blobs {
        staged.extend_from_slice(&b[..]);
    }
    staged
}

/// A `*_many` call that failed as a whole (device, allocation or library failure -- or a non-zero return with no unit carrying a
/// status) is an `Err` of the call; per-unit statuses are then not to be trusted.
fn whole_call(rc: i32, st: &[i32], what: &str) -> Result<(), Error> {
    let whole = matches!(rc, ffi::KZG355_INTERNAL | ffi::KZG355_NO_DEVICE | ffi::KZG355_NO_MEMORY | ffi::KZG355_DEVICE_ERROR);
    if whole || (rc != ffi::KZG355_OK && st.iter().all(|&x| x == 0)) {
        return check(rc, what);
    }
    Ok(())
}
