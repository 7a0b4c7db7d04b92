//! JSON trusted-setup helper (reference src/trusted_setup.rs:1-161): the `setup_G1_lagrange` / `setup_G2` hex arrays of the consensus-
//! specs trusted setup files <-> the byte vectors `Kzg::load_trusted_setup` takes; `Serialize` writes the same format back
//! (reference :21-29, 47-65), `g1_len` / `g2_len` as reference :40-45.
//! Difference from the reference, on purpose: it truncates the Lagrange points to FIELD_ELEMENTS_PER_BLOB after parsing (:138-153),
//! which for the minimal preset yields four points that load but are NOT a Lagrange basis of the size-4 domain.  A minimal-preset
//! build of this crate derives the size-4 Lagrange setup from the monomial `setup_G1` entries instead
//! (kzg355_lagrange_setup_from_monomial); the mainnet build keeps all 4096 points, as the reference does.
use crate::consts::*;
use crate::ffi;
use crate::kzg::Error;
use serde::de::{self, Deserializer};
use serde::ser::Serializer;
use serde::{Deserialize, Serialize};

/// One compressed point as `N` raw bytes; (de)serialised as a hex string, with or without the `0x` prefix on input.
#[derive(Debug, Clone, PartialEq)]
struct HexPoint<const N: usize>([u8; N]);

impl<const N: usize> Serialize for HexPoint<N> {
    fn serialize<S: Serializer>(&self, serializer: S) -> Result<S::Ok, S::Error> {
        serializer.serialize_str(&hex::encode(self.0))
    }
}
impl<'de, const N: usize> Deserialize<'de> for HexPoint<N> {
    fn deserialize<D: Deserializer<'de>>(deserializer: D) -> Result<Self, D::Error> {
        let text = String::deserialize(deserializer)?;
        let digits = text.strip_prefix("0x").unwrap_or(&text);
        let raw = hex::decode(digits).map_err(|e| de::Error::custom(format!("Failed to decode a {}-byte point: {}", N, e)))?;
        let arr: [u8; N] = raw
            .try_into()
            .map_err(|v: Vec<u8>| de::Error::custom(format!("point has invalid length. Expected {} got {}", N, v.len())))?;
        Ok(HexPoint(arr))
    }
}

#[derive(Debug, Clone, PartialEq, Serialize, Deserialize)]
pub struct TrustedSetup {
    #[serde(rename = "setup_G1_lagrange")]
    g1_points: Vec<HexPoint<BYTES_PER_G1>>,
    /// monomial-form points: present in the ceremony files, read only by a minimal-preset build
    #[serde(rename = "setup_G1", default, skip_serializing_if = "Vec::is_empty")]
    g1_monomial: Vec<HexPoint<BYTES_PER_G1>>,
    #[serde(rename = "setup_G2")]
    g2_points: Vec<HexPoint<BYTES_PER_G2>>,
}

impl TrustedSetup {
    pub fn from_json(text: &str) -> Result<Self, Error> {
        serde_json::from_str(text).map_err(|e| Error::InvalidTrustedSetup(format!("{}", e)))
    }
    pub fn to_json(&self) -> Result<String, Error> {
        serde_json::to_string(self).map_err(|e| Error::InvalidTrustedSetup(format!("{}", e)))
    }

    /// The G1 points `Kzg::load_trusted_setup` takes for this build's preset.
    pub fn g1_points(&self) -> Result<Vec<[u8; BYTES_PER_G1]>, Error> {
        if FIELD_ELEMENTS_PER_BLOB == 4096 {
            return Ok(self.g1_points.iter().map(|p| p.0).collect());
        }
        if self.g1_monomial.len() < FIELD_ELEMENTS_PER_BLOB {
            return Err(Error::InvalidTrustedSetup("setup_G1 (monomial form) is needed for the minimal preset".into()));
        }
        let mono: Vec<u8> = self.g1_monomial.iter().take(FIELD_ELEMENTS_PER_BLOB).flat_map(|p| p.0).collect();
        let mut out = vec![0u8; mono.len()];
        let rc = unsafe { ffi::kzg355_lagrange_setup_from_monomial(out.as_mut_ptr(), mono.as_ptr(), FIELD_ELEMENTS_PER_BLOB) };
        if rc != ffi::KZG355_OK {
            return Err(Error::InvalidTrustedSetup(format!("lagrange_setup_from_monomial: status {}", rc)));
        }
        Ok(out.chunks_exact(BYTES_PER_G1).map(|c| c.try_into().unwrap()).collect())
    }

    pub fn g2_points(&self) -> Vec<[u8; BYTES_PER_G2]> {
        self.g2_points.iter().map(|p| p.0).collect()
    }

    pub fn g1_len(&self) -> usize {
        self.g1_points.len()
    }
    pub fn g2_len(&self) -> usize {
        self.g2_points.len()
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn json_round_trip_and_prefixes() {
        let g1 = format!("\"0x{}\"", "ab".repeat(BYTES_PER_G1));
        let g2 = format!("\"{}\"", "cd".repeat(BYTES_PER_G2));
        let text = format!("{{\"setup_G1_lagrange\": [{}], \"setup_G2\": [{}, {}]}}", g1, g2, g2);
        let ts = TrustedSetup::from_json(&text).unwrap();
        assert_eq!((ts.g1_len(), ts.g2_len()), (1, 2));
        assert_eq!(ts.g2_points()[1], [0xcd; BYTES_PER_G2]);
        assert_eq!(TrustedSetup::from_json(&ts.to_json().unwrap()).unwrap(), ts);
        assert!(TrustedSetup::from_json(&text.replace("abab", "ab")).is_err());       // 47 bytes
        assert!(TrustedSetup::from_json(&text.replace("cdcd", "zzcd")).is_err());     // not hex
    }
}
