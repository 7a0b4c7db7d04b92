"""kzg_rust_amd -- MI355X-native (HIP / gfx950) engine for the KZG-4844 hot path of pawanjay176/kzg_rust.

`from kzg_rust_amd import Kzg, Blob, Bytes32, Bytes48, KzgCommitment, KzgProof` mirrors the reference crate's
re-exports (src/lib.rs:7-12).  The compute lives in libkzg355.so (C ABI: include/kzg355.h); importing this
package never falls back to a CPU implementation.
"""
from .kzg import (BYTES_PER_BLOB, BYTES_PER_COMMITMENT, BYTES_PER_FIELD_ELEMENT, BYTES_PER_G1, BYTES_PER_G2, BYTES_PER_PROOF,
                  FIELD_ELEMENTS_PER_BLOB, TRUSTED_SETUP_NUM_G2_POINTS, BadArgs, Blob, Bytes32, Bytes48, Error, InternalError,
                  InvalidBytesLength, InvalidHexFormat, InvalidTrustedSetup, Kzg, KzgCommitment, KzgProof, KzgSettings, NoDevice, NoMemory, DeviceError,
                  hex_to_bytes)

from .trusted_setup import TrustedSetup

__all__ = ["TrustedSetup", "Kzg", "KzgSettings", "Blob", "Bytes32", "Bytes48", "KzgCommitment", "KzgProof", "Error", "BadArgs", "InternalError",
           "InvalidBytesLength", "InvalidHexFormat", "InvalidTrustedSetup", "NoDevice", "NoMemory", "DeviceError", "hex_to_bytes", "BYTES_PER_BLOB",
           "BYTES_PER_COMMITMENT", "BYTES_PER_FIELD_ELEMENT", "BYTES_PER_G1", "BYTES_PER_G2", "BYTES_PER_PROOF",
           "FIELD_ELEMENTS_PER_BLOB", "TRUSTED_SETUP_NUM_G2_POINTS"]
