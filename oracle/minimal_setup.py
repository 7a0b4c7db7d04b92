"""Size-n Lagrange trusted setup from the first n MONOMIAL points [tau^k]G1 (TEST INFRASTRUCTURE, like the rest of oracle/).

The reference's JSON helper truncates the 4096-point Lagrange setup for the minimal preset (src/trusted_setup.rs:144-151),
which passes the loader's checks but is not a Lagrange basis of the size-4 domain.  The basis is derived instead:

    L_j(tau) G1 = (1/n) sum_k w^(-jk) [tau^k]G1,      w = 7^((r-1)/n)   (consts.rs:163-168),   j = 0..n-1 in natural order

(the loader then applies the bit-reversal permutation, kzg.rs:895-896).  Two independent evaluations: the C oracle's generic
lincomb, and oracle/pyref.py's big-integer group law."""
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def lagrange_scalars(n):
    w = pow(7, (R - 1) // n, R)
    winv, ninv = pow(w, -1, R), pow(n, -1, R)
    return [[pow(winv, j * k, R) * ninv % R for k in range(n)] for j in range(n)]


def lagrange_from_monomial(oracle, mono):
    """mono: list of n compressed monomial points -> list of n compressed Lagrange points (natural order), via the C oracle."""
    n = len(mono)
    return [oracle.g1_lincomb(mono, [s.to_bytes(32, "big") for s in row], fast=False) for row in lagrange_scalars(n)]


def lagrange_from_monomial_pyref(mono):
    from . import pyref
    pts = [pyref.g1_uncompress(b) for b in mono]
    return [pyref.g1_compress(pyref.g1_lincomb(pts, row)) for row in lagrange_scalars(len(mono))]
