"""Pure-Python big-int BLS12-381 + EIP-4844 KZG reference.  TEST INFRASTRUCTURE ONLY.

This file is an independent, slow, from-the-math restatement used for two things:
  1. deriving / cross-checking constants and formulas (tools/gen_constants.py imports it);
  2. spot-checking the C oracle (oracle/kzg_oracle.c) on small cases in tests/.
Nothing under kzg_rust_amd/ may import it.  It follows the control flow of the
reference where that matters (file:line cited per function, relative to /root/reference):

  src/kzg.rs:282-291   blob_to_polynomial        src/kzg.rs:298-339  compute_challenge
  src/kzg.rs:346-389   evaluate_polynomial_...   src/kzg.rs:461-528  compute_kzg_proof_impl
  src/kzg.rs:409-426   verify_kzg_proof_impl     src/kzg.rs:579-627  verify_kzg_proof_batch
  src/utils.rs:189-214 pairings_verify           src/utils.rs:282-310 validate_kzg_g1
  src/utils.rs:426-474 compute_r_powers

The arithmetic below the blst line (blst 0.3.11 is not in /root/reference) is restated from the
BLS12-381 definition: p, r, E: y^2=x^3+4, E': y^2=x^3+4(1+u), ZCash serialisation, optimal ate
pairing with x = -0xd201000000010000.
"""
import hashlib

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_ABS = 0xD201000000010000  # |x|; the BLS parameter x is negative

G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)

FIELD_ELEMENTS_PER_BLOB = 4096
BYTES_PER_BLOB = 4096 * 32


def set_preset(field_elements_per_blob):
    """The reference's FIELD_ELEMENTS_PER_BLOB is a compile-time constant (src/consts.rs:13): 4096 (mainnet) or 4 (minimal)."""
    global FIELD_ELEMENTS_PER_BLOB, BYTES_PER_BLOB
    FIELD_ELEMENTS_PER_BLOB = field_elements_per_blob
    BYTES_PER_BLOB = 32 * field_elements_per_blob
FIAT_SHAMIR_PROTOCOL_DOMAIN = b"FSBLOBVERIFY_V1_"
RANDOM_CHALLENGE_KZG_BATCH_DOMAIN = b"RCKZGBATCH___V1_"


class KzgError(Exception):
    pass


# ----------------------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_conj(a): return (a[0], (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return f2_mul(a, a)
def f2_muls(a, s): return (a[0] * s % P, a[1] * s % P)
def f2_mul_xi(a): return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)  # * (1+u)
def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)
F2_ZERO = (0, 0)
F2_ONE = (1, 0)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


def f2_sqrt(a):
    """Square root in Fp2 (p = 3 mod 4), or None.  Algorithm 9 of eprint 2012/685."""
    if a == F2_ZERO:
        return F2_ZERO
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(f2_sqr(a1), a)
    x0 = f2_mul(a1, a)
    if alpha == (P - 1, 0):
        r = (-x0[1] % P, x0[0])  # u * x0
    else:
        b = f2_pow(f2_add(F2_ONE, alpha), (P - 1) // 2)
        r = f2_mul(b, x0)
    return r if f2_sqr(r) == a else None


# ----------------------------------------------------------------------------- Fp6 = Fp2[v]/(v^3-xi)
def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)
def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0, t1, t2 = f2_mul(a0, b0), f2_mul(a1, b1), f2_mul(a2, b2)
    c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a1, a2), f2_add(b1, b2)), t1), t2)))
    c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a0, a1), f2_add(b0, b1)), t0), t1), f2_mul_xi(t2))
    c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a0, a2), f2_add(b0, b2)), t0), t2), t1)
    return (c0, c1, c2)
def f6_mul_v(a): return (f2_mul_xi(a[2]), a[0], a[1])
def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    d = f2_add(f2_mul(a0, t0), f2_mul_xi(f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    di = f2_inv(d)
    return (f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di))
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


# ----------------------------------------------------------------------------- Fp12 = Fp6[w]/(w^2-v)
def f12_mul(a, b):
    t0, t1 = f6_mul(a[0], b[0]), f6_mul(a[1], b[1])
    c0 = f6_add(t0, f6_mul_v(t1))
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), t0), t1)
    return (c0, c1)
def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return (a[0], f6_neg(a[1]))
def f12_inv(a):
    d = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mul_v(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], d), f6_neg(f6_mul(a[1], d)))
F12_ONE = (F6_ONE, F6_ZERO)


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_sqr(a)
        e >>= 1
    return r


# Frobenius constants: v^p = xi^((p-1)/3) v ; w^p = xi^((p-1)/6) w
XI = (1, 1)
FROB_V1 = f2_pow(XI, (P - 1) // 3)
FROB_V2 = f2_pow(XI, 2 * (P - 1) // 3)
FROB_W = f2_pow(XI, (P - 1) // 6)


def f6_frob(a):
    return (f2_conj(a[0]), f2_mul(f2_conj(a[1]), FROB_V1), f2_mul(f2_conj(a[2]), FROB_V2))


def f12_frob(a):
    d1 = f6_frob(a[1])
    return (f6_frob(a[0]), tuple(f2_mul(c, FROB_W) for c in d1))


# ----------------------------------------------------------------------------- G1 (affine, None = infinity)
def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - 4) % P == 0


def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x3 = (lam * lam - a[0] - b[0]) % P
    return (x3, (lam * (a[0] - x3) - a[1]) % P)


def g1_neg(a): return None if a is None else (a[0], (-a[1]) % P)


def g1_mul(a, k):
    r = None
    while k:
        if k & 1:
            r = g1_add(r, a)
        a = g1_add(a, a)
        k >>= 1
    return r


def g1_compress(pt):
    """ZCash 48-byte compressed form (what blst_p1_compress emits; utils.rs:221-227)."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    flags = 0x80 | (0x20 if y > (P - 1) // 2 else 0)
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= flags
    return bytes(b)


def g1_uncompress(b):
    """blst_p1_uncompress semantics (utils.rs:290): KzgError on bad encoding / not on curve."""
    if len(b) != 48:
        raise KzgError("length")
    if not b[0] & 0x80:
        raise KzgError("uncompressed form not accepted")
    if b[0] & 0x40:
        if (b[0] & 0x3F) or any(b[1:]):
            raise KzgError("bad infinity encoding")
        return None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    if x >= P:
        raise KzgError("x >= p")
    y2 = (x * x * x + 4) % P
    y = pow(y2, (P + 1) // 4, P)
    if y * y % P != y2:
        raise KzgError("not on curve")
    if (y > (P - 1) // 2) != bool(b[0] & 0x20):
        y = P - y
    return (x, y)


def validate_kzg_g1(b):
    """utils.rs:282-310: uncompress, infinity accepted, subgroup check."""
    pt = g1_uncompress(b)
    if pt is None:
        return None
    if g1_mul(pt, R) is not None:
        raise KzgError("not in G1")
    return pt


# ----------------------------------------------------------------------------- G2 (affine over Fp2)
B2 = (4, 4)


def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO


def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if f2_add(a[1], b[1]) == F2_ZERO:
            return None
        lam = f2_mul(f2_muls(f2_sqr(a[0]), 3), f2_inv(f2_muls(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))


def g2_neg(a): return None if a is None else (a[0], f2_neg(a[1]))


def g2_mul(a, k):
    r = None
    while k:
        if k & 1:
            r = g2_add(r, a)
        a = g2_add(a, a)
        k >>= 1
    return r


def g2_uncompress(b):
    """96-byte ZCash compressed G2: x.c1 (with flags) || x.c0 (blst_p2_uncompress, kzg.rs:877)."""
    if len(b) != 96:
        raise KzgError("length")
    if not b[0] & 0x80:
        raise KzgError("uncompressed form not accepted")
    if b[0] & 0x40:
        if (b[0] & 0x3F) or any(b[1:]):
            raise KzgError("bad infinity encoding")
        return None
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:], "big")
    if x0 >= P or x1 >= P:
        raise KzgError("x >= p")
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
    if y is None:
        raise KzgError("not on curve")
    big = y[1] > (P - 1) // 2 if y[1] != 0 else y[0] > (P - 1) // 2
    if big != bool(b[0] & 0x20):
        y = f2_neg(y)
    return (x, y)


# ----------------------------------------------------------------------------- pairing
def _line(T, Q, Pt):
    """Line through untwisted T,Q (T==Q: tangent) evaluated at Pt=(xP,yP), scaled by w^3 and an
    Fp2 factor (both die in the final exponentiation):  c + (-lam*xP) v + yP v w  with
    c = lam*x_T - y_T  ('line functions').  Returns (fp12 element, T+Q)."""
    xP, yP = Pt
    if T[0] == Q[0] and T[1] == Q[1]:
        lam = f2_mul(f2_muls(f2_sqr(T[0]), 3), f2_inv(f2_muls(T[1], 2)))
    else:
        lam = f2_mul(f2_sub(Q[1], T[1]), f2_inv(f2_sub(Q[0], T[0])))
    c = f2_sub(f2_mul(lam, T[0]), T[1])
    l = ((c, f2_muls(f2_neg(lam), xP), F2_ZERO), (F2_ZERO, (yP, 0), F2_ZERO))
    x3 = f2_sub(f2_sub(f2_sqr(lam), T[0]), Q[0])
    y3 = f2_sub(f2_mul(lam, f2_sub(T[0], x3)), T[1])
    return l, (x3, y3)


def miller_loop(Q, Pt):
    """f_{|x|,Q}(P) conjugated (x<0).  Q in E'(Fp2) affine, Pt in E(Fp) affine; either None -> 1."""
    if Q is None or Pt is None:
        return F12_ONE
    f = F12_ONE
    T = Q
    for i in range(X_ABS.bit_length() - 2, -1, -1):
        l, T = _line(T, T, Pt)
        f = f12_mul(f12_sqr(f), l)
        if (X_ABS >> i) & 1:
            l, T = _line(T, Q, Pt)
            f = f12_mul(f, l)
    return f12_conj(f)


def _cyc_exp_x(a):
    """a^x for a in the cyclotomic subgroup (x negative: conjugate = inverse there)."""
    return f12_conj(f12_pow(a, X_ABS))


def final_exp_is_one(f):
    """f^((p^12-1)/r) == 1, computed as easy part then the 3*hard-part chain
    3(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3."""
    f = f12_mul(f12_conj(f), f12_inv(f))              # ^(p^6-1)
    f = f12_mul(f12_frob(f12_frob(f)), f)             # ^(p^2+1)
    a = f12_mul(_cyc_exp_x(f), f12_conj(f))           # f^(x-1)
    a = f12_mul(_cyc_exp_x(a), f12_conj(a))           # ^(x-1)
    b = f12_mul(_cyc_exp_x(a), f12_frob(a))           # ^(x+p)
    c = f12_mul(f12_mul(_cyc_exp_x(_cyc_exp_x(b)), f12_frob(f12_frob(b))), f12_conj(b))  # ^(x^2+p^2-1)
    res = f12_mul(c, f12_mul(f12_sqr(f), f))
    return res == F12_ONE


def pairings_verify(a1, a2, b1, b2):
    """utils.rs:189-214: e(a1,a2) == e(b1,b2) via ML(a2,-a1)*ML(b2,b1) -> final exp -> is_one."""
    f = f12_mul(miller_loop(a2, g1_neg(a1)), miller_loop(b2, b1))
    return final_exp_is_one(f)


# ----------------------------------------------------------------------------- Fr helpers / KZG
def reverse_bits(n, order):  # kzg.rs:700-710
    r = 0
    for _ in range(order.bit_length() - 1):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def bit_reversal_permutation(v):  # kzg.rs:717-731
    n = len(v)
    return [v[reverse_bits(i, n)] for i in range(n)]


def compute_roots_of_unity(n=FIELD_ELEMENTS_PER_BLOB):  # kzg.rs:764-799 (7^((r-1)/n), consts.rs:163-168)
    root = pow(7, (R - 1) // n, R)
    out, cur = [], 1
    for _ in range(n):
        out.append(cur)
        cur = cur * root % R
    assert cur == 1
    return bit_reversal_permutation(out)


class Settings:
    """kzg.rs:28-40 KzgSettings; load_trusted_setup kzg.rs:833-899."""

    def __init__(self, g1_bytes, g2_bytes, check_lagrange=True):
        if len(g1_bytes) != FIELD_ELEMENTS_PER_BLOB or len(g2_bytes) != 65:
            raise KzgError("bad counts")
        g1 = [g1_uncompress(b) for b in g1_bytes]
        g2 = [g2_uncompress(b) for b in g2_bytes]
        if check_lagrange and pairings_verify(g1[1], g2[0], g1[0], g2[1]):  # kzg.rs:802-830
            raise KzgError("monomial form")
        self.roots = compute_roots_of_unity(FIELD_ELEMENTS_PER_BLOB)
        self.g1 = bit_reversal_permutation(g1)
        self.g2 = g2


def bytes_to_bls_field(b):  # utils.rs:262-275
    if len(b) != 32:
        raise KzgError("len")
    v = int.from_bytes(b, "big")
    if v >= R:
        raise KzgError("non-canonical")
    return v


def blob_to_polynomial(blob):  # kzg.rs:282-291
    if len(blob) != BYTES_PER_BLOB:
        raise KzgError("blob length")
    return [bytes_to_bls_field(blob[i * 32:(i + 1) * 32]) for i in range(FIELD_ELEMENTS_PER_BLOB)]


def compute_challenge(blob, commitment_bytes):  # kzg.rs:298-339
    validate_kzg_g1(commitment_bytes)
    msg = FIAT_SHAMIR_PROTOCOL_DOMAIN + (0).to_bytes(8, "big") + FIELD_ELEMENTS_PER_BLOB.to_bytes(8, "big")
    msg += bytes(blob) + bytes(commitment_bytes)
    return int.from_bytes(hashlib.sha256(msg).digest(), "big") % R


def evaluate_polynomial_in_evaluation_form(poly, z, s):  # kzg.rs:346-389
    n = FIELD_ELEMENTS_PER_BLOB
    for i in range(n):
        if z == s.roots[i]:
            return poly[i]
    acc = 0
    for i in range(n):
        acc = (acc + pow(z - s.roots[i], -1, R) * s.roots[i] % R * poly[i]) % R
    acc = acc * pow(n, -1, R) % R
    return acc * (pow(z, n, R) - 1) % R


def g1_lincomb(points, scalars):  # utils.rs:329-342 / 367-410 (same group element either way)
    acc = None
    for pt, k in zip(points, scalars):
        acc = g1_add(acc, g1_mul(pt, k % R))
    return acc


def blob_to_kzg_commitment(blob, s):  # kzg.rs:401-406
    return g1_compress(g1_lincomb(s.g1, blob_to_polynomial(blob)))


def compute_kzg_proof_impl(poly, z, s):  # kzg.rs:461-528
    n = FIELD_ELEMENTS_PER_BLOB
    y = evaluate_polynomial_in_evaluation_form(poly, z, s)
    q = [0] * n
    m = None
    for i in range(n):
        if z == s.roots[i]:
            m = i
            continue
        q[i] = (poly[i] - y) * pow(s.roots[i] - z, -1, R) % R
    if m is not None:
        acc = 0
        for i in range(n):
            if i == m:
                continue
            acc = (acc + (poly[i] - y) * s.roots[i] % R * pow(z * (z - s.roots[i]) % R, -1, R)) % R
        q[m] = acc
    return g1_compress(g1_lincomb(s.g1, q)), y


def compute_kzg_proof(blob, z_bytes, s):  # kzg.rs:446-457
    poly = blob_to_polynomial(blob)
    z = bytes_to_bls_field(z_bytes)
    proof, y = compute_kzg_proof_impl(poly, z, s)
    return proof, y.to_bytes(32, "big")


def compute_blob_kzg_proof(blob, commitment_bytes, s):  # kzg.rs:533-544
    poly = blob_to_polynomial(blob)
    z = compute_challenge(blob, commitment_bytes)
    return compute_kzg_proof_impl(poly, z, s)[0]


def verify_kzg_proof_impl(c, z, y, proof, s):  # kzg.rs:409-426
    x_minus_z = g2_add(s.g2[1], g2_neg(g2_mul(G2_GEN, z)))
    p_minus_y = g1_add(c, g1_neg(g1_mul(G1_GEN, y)))
    return pairings_verify(p_minus_y, G2_GEN, proof, x_minus_z)


def verify_kzg_proof(cb, zb, yb, pb, s):  # kzg.rs:429-443
    c = validate_kzg_g1(cb)
    z = bytes_to_bls_field(zb)
    y = bytes_to_bls_field(yb)
    pr = validate_kzg_g1(pb)
    return verify_kzg_proof_impl(c, z, y, pr, s)


def verify_blob_kzg_proof(blob, cb, pb, s):  # kzg.rs:547-569
    poly = blob_to_polynomial(blob)
    c = validate_kzg_g1(cb)
    pr = validate_kzg_g1(pb)
    z = compute_challenge(blob, cb)
    y = evaluate_polynomial_in_evaluation_form(poly, z, s)
    return verify_kzg_proof_impl(c, z, y, pr, s)


def compute_r_powers(cs, zs, ys, prs):  # utils.rs:426-474
    n = len(cs)
    msg = RANDOM_CHALLENGE_KZG_BATCH_DOMAIN + FIELD_ELEMENTS_PER_BLOB.to_bytes(8, "big") + n.to_bytes(8, "big")
    for i in range(n):
        msg += g1_compress(cs[i]) + zs[i].to_bytes(32, "big") + ys[i].to_bytes(32, "big") + g1_compress(prs[i])
    r = int.from_bytes(hashlib.sha256(msg).digest(), "big") % R
    return [pow(r, i, R) for i in range(n)]


def verify_kzg_proof_batch(cs, zs, ys, prs, s):  # kzg.rs:579-627
    n = len(cs)
    if n == 0:
        raise KzgError("empty")
    rp = compute_r_powers(cs, zs, ys, prs)
    proof_lincomb = g1_lincomb(prs, rp)
    c_minus_y = [g1_add(cs[i], g1_neg(g1_mul(G1_GEN, ys[i]))) for i in range(n)]
    r_times_z = [rp[i] * zs[i] % R for i in range(n)]
    rhs = g1_add(g1_lincomb(c_minus_y, rp), g1_lincomb(prs, r_times_z))
    return pairings_verify(proof_lincomb, s.g2[1], rhs, G2_GEN)


def verify_blob_kzg_proof_batch(blobs, cbs, pbs, s):  # kzg.rs:637-693
    n = len(blobs)
    if len(cbs) != n or len(pbs) != n:
        raise KzgError("length mismatch")
    if n == 0:
        return True
    if n == 1:
        return verify_blob_kzg_proof(blobs[0], cbs[0], pbs[0], s)
    cs, zs, ys, prs = [], [], [], []
    for i in range(n):
        cs.append(validate_kzg_g1(cbs[i]))
        poly = blob_to_polynomial(blobs[i])
        zs.append(compute_challenge(blobs[i], cbs[i]))
        ys.append(evaluate_polynomial_in_evaluation_form(poly, zs[-1], s))
        prs.append(validate_kzg_g1(pbs[i]))
    return verify_kzg_proof_batch(cs, zs, ys, prs, s)
