"""JSON trusted-setup helper: the counterpart of the reference's `TrustedSetup` (src/trusted_setup.rs:21-44, 138-153).

Same wire format as the consensus-specs `testing_trusted_setups.json`: an object with `setup_G1_lagrange` (hex strings of
48-byte compressed G1 points, with or without 0x) and `setup_G2` (96-byte compressed G2 points); other keys (`setup_G1`,
`roots_of_unity`) are ignored.  As in the reference, the G1 list is truncated to FIELD_ELEMENTS_PER_BLOB after parsing
(trusted_setup.rs:138-153).  The parsed points feed `Kzg.load_trusted_setup` (kzg.rs:1005), which does all the checking
on the device."""
import json

from .kzg import BYTES_PER_G1, BYTES_PER_G2, FIELD_ELEMENTS_PER_BLOB, InvalidBytesLength, InvalidHexFormat, hex_to_bytes


class TrustedSetup:
    def __init__(self, g1_points, g2_points):
        self._g1 = list(g1_points)
        self._g2 = list(g2_points)

    @staticmethod
    def _point(s, size):
        if not isinstance(s, str):
            raise InvalidHexFormat("trusted setup points must be hex strings")
        b = hex_to_bytes(s)                                  # strip_prefix + hex::decode (trusted_setup.rs:155-161)
        if len(b) != size:
            raise InvalidBytesLength(f"Invalid byte length. Expected {size} got {len(b)}")
        return b

    @classmethod
    def from_json(cls, text):
        d = json.loads(text)
        g1 = [cls._point(x, BYTES_PER_G1) for x in d["setup_G1_lagrange"]]
        g2 = [cls._point(x, BYTES_PER_G2) for x in d["setup_G2"]]
        return cls(g1[:FIELD_ELEMENTS_PER_BLOB], g2)         # truncate (trusted_setup.rs:151)

    @classmethod
    def from_file(cls, path):
        with open(path) as f:
            return cls.from_json(f.read())

    def to_json(self):
        return json.dumps({"setup_G1_lagrange": ["0x" + p.hex() for p in self._g1], "setup_G2": ["0x" + p.hex() for p in self._g2]})

    def g1_points(self):
        return list(self._g1)

    def g2_points(self):
        return list(self._g2)

    def g1_len(self):
        return len(self._g1)

    def g2_len(self):
        return len(self._g2)
