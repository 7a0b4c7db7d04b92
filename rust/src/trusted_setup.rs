//! JSON trusted-setup helper (reference src/trusted_setup.rs): `setup_G1_lagrange` / `setup_G2` hex arrays -> the byte vectors
//! `Kzg::load_trusted_setup` takes.  Unlike the reference, a minimal-preset build does not truncate the 4096 Lagrange points
//! (that is not a basis of the size-4 domain): it derives the size-4 Lagrange setup from the monomial `setup_G1` entries with
//! kzg355_lagrange_setup_from_monomial.
use crate::consts::*;
use crate::ffi;
use crate::kzg::{hex_to_bytes, Error};
use serde::Deserialize;

#[derive(Debug, Clone, Deserialize)]
pub struct TrustedSetup {
    #[serde(rename = "setup_G1_lagrange")]
    g1_lagrange: Vec<String>,
    #[serde(rename = "setup_G1", default)]
    g1_monomial: Vec<String>,
    #[serde(rename = "setup_G2")]
    g2: Vec<String>,
}

impl TrustedSetup {
    pub fn from_json(text: &str) -> Result<Self, Error> {
        serde_json::from_str(text).map_err(|e| Error::InvalidTrustedSetup(format!("{}", e)))
    }

    pub fn g1_points(&self) -> Result<Vec<[u8; BYTES_PER_G1]>, Error> {
        if FIELD_ELEMENTS_PER_BLOB == 4096 {
            return self.g1_lagrange.iter().map(|h| to_array::<BYTES_PER_G1>(h)).collect();
        }
        let mono: Vec<u8> = self
            .g1_monomial
            .iter()
            .take(FIELD_ELEMENTS_PER_BLOB)
            .map(|h| hex_to_bytes(h))
            .collect::<Result<Vec<_>, _>>()?
            .concat();
        if mono.len() != FIELD_ELEMENTS_PER_BLOB * BYTES_PER_G1 {
            return Err(Error::InvalidTrustedSetup("setup_G1 (monomial form) is needed for the minimal preset".into()));
        }
        let mut out = vec![0u8; mono.len()];
        let rc = unsafe { ffi::kzg355_lagrange_setup_from_monomial(out.as_mut_ptr(), mono.as_ptr(), FIELD_ELEMENTS_PER_BLOB) };
        if rc != ffi::KZG355_OK {
            return Err(Error::InvalidTrustedSetup(format!("lagrange_setup_from_monomial: status {}", rc)));
        }
        Ok(out.chunks_exact(BYTES_PER_G1).map(|c| c.try_into().unwrap()).collect())
    }

    pub fn g2_points(&self) -> Result<Vec<[u8; BYTES_PER_G2]>, Error> {
        self.g2.iter().map(|h| to_array::<BYTES_PER_G2>(h)).collect()
    }
}

fn to_array<const N: usize>(h: &str) -> Result<[u8; N], Error> {
    let b = hex_to_bytes(h)?;
    b.try_into().map_err(|_| Error::InvalidBytesLength(format!("expected {} bytes", N)))
}
