//! kzg_rust with the MI355X engine behind `Kzg` (see README.md in this directory).
mod consts;
pub mod ffi; // public: `KzgSettings::load_trusted_setup_with_options` takes `ffi::kzg355_options`
mod kzg;
mod trusted_setup;

pub use consts::{
    BYTES_PER_BLOB, BYTES_PER_COMMITMENT, BYTES_PER_FIELD_ELEMENT, BYTES_PER_G1, BYTES_PER_G2, BYTES_PER_PROOF, FIELD_ELEMENTS_PER_BLOB,
};
pub use kzg::{Blob, Bytes32, Bytes48, Error, Kzg, KzgCommitment, KzgProof, KzgSettings};
pub use trusted_setup::TrustedSetup;
