"""`kzg_mainnet` of the reference's README (README.md:8-9): FIELD_ELEMENTS_PER_BLOB = 4096.  Same engine, same `Kzg`; the
preset of a call is the preset of the settings handle it is given."""
from .kzg import (BYTES_PER_BLOB, BYTES_PER_COMMITMENT, BYTES_PER_FIELD_ELEMENT, BYTES_PER_PROOF, FIELD_ELEMENTS_PER_BLOB, Blob, Bytes32, Bytes48,  # noqa: F401
                  Error, Kzg, KzgCommitment, KzgProof, KzgSettings)
