// GPU unit test of the DPP-quad G1 routines of k_g1.hip (g1_dbl_quad, g1_add_quad) against the plain lane routines of g1.h: 16 quads
// of one wave, each with its own operands, including the special cases of the complete addition (an operand at infinity, P = Q,
// P = -Q).  Prints one line per case; exit code 1 on any mismatch.  Built by __graft_entry__.build(), run by tests/test_gpu_quad_ops.py.
#define KZG_MID_INLINE 1
#include <hip/hip_runtime.h>
#include "../../kzg_rust_amd/csrc/g1_quad.h"
#include <cstdio>
using namespace kzg;

__device__ bool same_point(const G1Jac &a, const G1Jac &b) {
    G1Jac ca, cb; g1_canon_lazy(ca, a); g1_canon_lazy(cb, b);
    G1Affine x, y; g1_to_affine(x, ca); g1_to_affine(y, cb);
    return fp_eq(x.x, y.x) && fp_eq(x.y, y.y);
}
// kind: 0 generic, 1 b = infinity, 2 a = infinity, 3 a = b, 4 a = -b, 5 doubling chain, 6 sums of lazy results (operands straight from a previous quad addition)
__global__ void __launch_bounds__(64) k_quad_test(int kind, int *bad) {
    const int lane = threadIdx.x, role = lane & 3, quad = lane >> 2;
    const uint32_t gx[NFP] = G1_GEN_X_INIT, gy[NFP] = G1_GEN_Y_INIT;
    G1Affine g; for (int q = 0; q < NFP; q++) { g.x.l[q] = gx[q]; g.y.l[q] = gy[q]; }
    uint32_t ka[8] = {(uint32_t)(0x1234567u * (quad + 3)), 0x9abcdefu + quad, 7u * quad + 1, 0, 0, 0, 0, 0}, kb[8] = {(uint32_t)(0x7654321u * (quad + 11)), 0x1357u + 13 * quad, 5, 0, 0, 0, 0, 0};
    G1Jac a, b;
    g1_mul_words(a, g, ka, 3); g1_mul_words(b, g, kb, 3);        // Jacobian points with non-trivial z
    if (kind == 1) b = g1_inf();
    if (kind == 2) a = g1_inf();
    if (kind == 3) b = a;
    if (kind == 4) { b = a; fp_neg(b.y, b.y); }
    bool ok = true;
    if (kind == 5) {
        G1Jac r1 = a, r2 = a;
        for (int i = 0; i < 7; i++) { g1_dbl(r1, r1); g1_dbl_quad(r2, role); }
        ok = same_point(r1, r2);
    } else if (kind == 6) {
        G1Jac s, t, u1, u2;
        g1_add_quad(s, a, b, role); g1_add_quad(t, b, a, role);  // lazy outputs as operands
        g1_add_quad(u1, s, a, role);
        G1Jac w; g1_add(w, a, b); g1_add(u2, w, a);
        ok = same_point(u1, u2) && same_point(s, t);
        G1Jac d1, d2; g1_add_quad(d1, s, t, role); g1_dbl(d2, w);           // equal points in different representations: the doubling case
        ok = ok && same_point(d1, d2);
    } else {
        G1Jac r1, r2;
        g1_add(r1, a, b);
        g1_add_quad(r2, a, b, role);
        ok = same_point(r1, r2);
        G1Jac r3 = a; g1_add_quad(r3, r3, b, role);              // aliased destination, as the kernels call it
        ok = ok && same_point(r1, r3);
    }
    if (!ok) atomicOr(bad, 1 << kind);
}

__device__ bool fp_same(const Fp &a, const Fp &b) { Fp x, y; fp_canon64(x, a); fp_canon64(y, b); return fp_eq(x, y); }
__device__ int add_quad_stage_check(const G1Jac &a, const G1Jac &b, int role) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m8[NFP] = FP_MOD8_INIT;
    Fp u, v, w, t;
    fp_select(u, role == 1, a.z, b.z); fp_select(u, role == 2, u, a.y); fp_select(u, role == 3, u, b.y);
    fp_select(v, role == 1 || role == 2, a.z, b.z);
    fp_mul_lz(w, u, v);
    const Fp Z1Z1 = fp_quad_bcast<0>(w), Z2Z2 = fp_quad_bcast<1>(w), T1 = fp_quad_bcast<2>(w), T2 = fp_quad_bcast<3>(w);
    Fp e;
    fp_mul_lz(e, a.z, a.z); if (!fp_same(e, Z1Z1)) return 11;
    fp_mul_lz(e, b.z, b.z); if (!fp_same(e, Z2Z2)) return 12;
    fp_mul_lz(e, a.y, b.z); if (!fp_same(e, T1)) return 13;
    fp_mul_lz(e, b.y, a.z); if (!fp_same(e, T2)) return 14;
    fp_select(u, role == 1, a.x, b.x); fp_select(u, role == 2, u, T1); fp_select(u, role == 3, u, T2);
    fp_select(v, role == 1 || role == 3, Z2Z2, Z1Z1);
    fp_mul_lz(w, u, v);
    const Fp U1 = fp_quad_bcast<0>(w), U2 = fp_quad_bcast<1>(w), S1 = fp_quad_bcast<2>(w), S2 = fp_quad_bcast<3>(w);
    fp_mul_lz(e, a.x, Z2Z2); if (!fp_same(e, U1)) return 21;
    fp_mul_lz(e, b.x, Z1Z1); if (!fp_same(e, U2)) return 22;
    fp_mul_lz(e, T1, Z2Z2); if (!fp_same(e, S1)) return 23;
    fp_mul_lz(e, T2, Z1Z1); if (!fp_same(e, S2)) return 24;
    Fp H, R;
    fp_sub_lz(H, U2, U1, m2);
    fp_sub_lz(R, S2, S1, m2);
    fp_select(u, role == 1, H, R); fp_select(u, role >= 2, u, a.z);
    fp_select(v, role == 1, H, R); fp_select(v, role >= 2, v, b.z);
    fp_mul_lz(w, u, v);
    const Fp HH = fp_quad_bcast<0>(w), RR = fp_quad_bcast<1>(w), ZZ = fp_quad_bcast<2>(w);
    fp_mul_lz(e, H, H); if (!fp_same(e, HH)) return 31;
    fp_mul_lz(e, R, R); if (!fp_same(e, RR)) return 32;
    fp_mul_lz(e, a.z, b.z); if (!fp_same(e, ZZ)) return 33;
    fp_select(u, role == 1, H, U1); fp_select(u, role >= 2, u, ZZ);
    fp_select(v, role >= 2, HH, H);
    fp_mul_lz(w, u, v);
    const Fp HHH = fp_quad_bcast<0>(w), V = fp_quad_bcast<1>(w), Z3 = fp_quad_bcast<2>(w);
    fp_mul_lz(e, H, HH); if (!fp_same(e, HHH)) return 41;
    fp_mul_lz(e, U1, HH); if (!fp_same(e, V)) return 42;
    fp_mul_lz(e, ZZ, H); if (!fp_same(e, Z3)) return 43;
    Fp X3;
    fp_sub_lz(t, RR, HHH, m2); fp_sub_lz(u, t, V, m2); fp_sub_lz(X3, u, V, m2);
    fp_sub_lz(t, V, X3, m8);
    fp_select(u, role == 0, S1, R);
    fp_select(v, role == 0, HHH, t);
    fp_mul_lz(w, u, v);
    const Fp A = fp_quad_bcast<0>(w), B = fp_quad_bcast<1>(w);
    fp_mul_lz(e, R, t); if (!fp_same(e, A)) return 51;
    fp_mul_lz(e, S1, HHH); if (!fp_same(e, B)) return 52;
    G1Jac L; g1_add_lazy2(L, a, b);
    Fp Y3; fp_sub_lz(Y3, A, B, m2);
    if (!fp_same(L.x, X3)) return 61;
    if (!fp_same(L.y, Y3)) return 62;
    if (!fp_same(L.z, Z3)) return 63;
    G1Jac q; g1_add_quad(q, a, b, role);                          // the real routine (with its rare-path branch)
    if (!fp_same(L.x, q.x)) return 71;
    if (!fp_same(L.y, q.y)) return 72;
    if (!fp_same(L.z, q.z)) return 73;
    G1Jac c1; g1_add(c1, a, b);
    if (!same_point(c1, L)) return 81;
    if (!same_point(c1, q)) return 82;
    return 0;
}
__global__ void __launch_bounds__(64) k_quad_stage(int *out) {
    const int lane = threadIdx.x, role = lane & 3, quad = lane >> 2;
    const uint32_t gx[NFP] = G1_GEN_X_INIT, gy[NFP] = G1_GEN_Y_INIT;
    G1Affine g; for (int q = 0; q < NFP; q++) { g.x.l[q] = gx[q]; g.y.l[q] = gy[q]; }
    uint32_t ka[8] = {(uint32_t)(0x1234567u * (quad + 3)), 0x9abcdefu + quad, 7u * quad + 1, 0, 0, 0, 0, 0}, kb[8] = {(uint32_t)(0x7654321u * (quad + 11)), 0x1357u + 13 * quad, 5, 0, 0, 0, 0, 0};
    G1Jac a, b;
    g1_mul_words(a, g, ka, 3); g1_mul_words(b, g, kb, 3);
    out[lane] = add_quad_stage_check(a, b, role);
}
int main() {
    int *d_bad, h_bad = 0;
    if (hipMalloc(&d_bad, sizeof(int)) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemset(d_bad, 0, sizeof(int));
    for (int kind = 0; kind <= 6; kind++) hipLaunchKernelGGL(k_quad_test, dim3(1), dim3(64), 0, 0, kind, d_bad);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
    {
        int *d_out, h_out[64];
        (void)hipMalloc(&d_out, 64 * sizeof(int));
        hipLaunchKernelGGL(k_quad_stage, dim3(1), dim3(64), 0, 0, d_out);
        (void)hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
        printf("first mismatching stage per lane:"); for (int i = 0; i < 64; i++) printf(" %d", h_out[i]); printf("\n");
    }
    const char *names[7] = {"generic addition", "b at infinity", "a at infinity", "a = b", "a = -b", "doubling chain", "lazy operands / doubling case"};
    for (int kind = 0; kind <= 6; kind++) printf("%-32s %s\n", names[kind], (h_bad >> kind) & 1 ? "MISMATCH" : "ok");
    return h_bad ? 1 : 0;
}
