//! Sizes of the wire types (reference src/consts.rs:5-37).  The blst limb constants of the reference's consts.rs have no
//! counterpart here: no field or curve arithmetic happens on this side of the boundary.
pub const BYTES_PER_FIELD_ELEMENT: usize = 32;
pub const BYTES_PER_COMMITMENT: usize = 48;
pub const BYTES_PER_PROOF: usize = 48;
#[cfg(not(feature = "minimal"))]
pub const FIELD_ELEMENTS_PER_BLOB: usize = 4096;
#[cfg(feature = "minimal")]
pub const FIELD_ELEMENTS_PER_BLOB: usize = 4;
pub const BYTES_PER_BLOB: usize = FIELD_ELEMENTS_PER_BLOB * BYTES_PER_FIELD_ELEMENT;
pub const BYTES_PER_G1: usize = 48;
pub const BYTES_PER_G2: usize = 96;
pub const TRUSTED_SETUP_NUM_G2_POINTS: usize = 65;
