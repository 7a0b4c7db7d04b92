// k_verify.hip -- HIP kernels of the verify_blob_kzg_proof_batch path (reference src/kzg.rs:637-693 and
// 579-627, src/utils.rs:250-342, 426-474), gfx950.
//
// Stage 1 (independent per blob -- "HOT LOOP A", kzg.rs:671-683):
//   k_validate_points   validate_kzg_g1 on C_i and proof_i          (utils.rs:282-310)   1 lane / point
//   k_challenge         z_i = SHA-256 Fiat-Shamir challenge          (kzg.rs:298-339)     1 lane / blob
//   k_eval              blob -> Fr (canonical check) and y_i=p_i(z_i) (kzg.rs:282-291, 346-389) 1 wave / blob
// Stage 2 (per batch of n records -- verify_kzg_proof_batch, kzg.rs:579-627):
//   k_rpowers           r = hash of the transcript, r^i, r^i z_i, sum r^i y_i   (utils.rs:426-474)
//   k_lincomb           sum r^i proof_i  and  sum r^i C_i + sum r^i z_i proof_i - [sum r^i y_i]G
//   k_pairing           e(proof_lincomb, [tau]G2) == e(rhs, G2)                 (utils.rs:189-214)
// The rhs is algebraically the reference's  sum r^i (C_i - [y_i]G) + sum r^i z_i proof_i  (kzg.rs:603-622): same
// group element, so the same boolean.
#define KZG_FP_MUL_NOINLINE 1
#include "kernels.h"
#include "fr_block.h"

namespace kzg {


// ------------------------------------------------------------------------------------------------ challenge
// z_i = SHA-256( "FSBLOBVERIFY_V1_" | u64be(0) | u64be(4096) | blob | commitment ) mod r   (kzg.rs:298-339; 131,152 bytes,
// consts.rs:19-22) = 2050 compressions that are strictly sequential per blob (Merkle-Damgard), one blob per lane.
// A lone wave issues one instruction per ~5 cycles, so the only lever on this chain is its instruction count.  The
// message schedule does not depend on the chaining state, so it is moved to a second wave: in each 128-thread
// workgroup wave 1 (producer) loads block b+1, expands W[0..63] and stores W[t]+K[t] to LDS while wave 0 (consumer)
// runs the 64 rounds of block b from LDS (ds_read_b128, [t/4][lane][4] layout: conflict-free).  One barrier per block.
// Rounds use v_alignbit (rotates), v_bitop3 (3-input xor / ch / maj) and v_add3.  Also assembles the record's
// C / z / proof fields.
__device__ __forceinline__ uint32_t ror(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t ch3(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca); }   // e ? f : g
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }

constexpr int CH_BLOCKS = 2050;
// The hashed message puts the blob at offset 32, so compression b covers blob[64b-32, 64b+32): every block straddles two
// aligned 64-byte sectors of the blob.  Reading "the block's 64 bytes" touches each sector twice, two blocks apart in time, and
// even aligned 64-byte reads leave the other half of each 128-byte line to a later block: at full-card sizes those second
// touches miss (FETCH_SIZE: 281 KB, then 191 KB, per 131 KB blob).  The reader below loads each 128-byte LINE once (every
// second block, one block ahead), uses the lower half of the current sector now and carries its upper half to the next block.
struct ChallengeReader {
    const uint4 *blob; const uint8_t *cm;
    uint4 line[8];           // blob[128 l, 128 l + 128) for l = b >> 1, loaded by prefetch(b) with b even
    uint32_t carry[8];       // upper half of sector b-1, big-endian words
    __device__ __forceinline__ void prefetch(int b) {
        if (b < 2048 && !(b & 1)) {
            const uint4 *p = blob + 4 * b;
#pragma unroll
            for (int q = 0; q < 8; q++) line[q] = p[q];
        }
    }
    // words of compression b; prefetch(b) must have run (and no later prefetch).  Leaves carry ready for b + 1.
    __device__ __forceinline__ void words(uint32_t w[16], int b) {
        if (b == 0) {                                   // domain | u64be(0) | u64be(4096)
            w[0] = 0x4653424cu; w[1] = 0x4f425645u; w[2] = 0x52494659u; w[3] = 0x5f56315fu;   // "FSBLOBVERIFY_V1_"
            w[4] = 0; w[5] = 0; w[6] = 0; w[7] = (uint32_t)N_FE;
        } else if (b <= 2048) {
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = carry[k];
        }
        if (b < 2048) {
            uint4 s0, s1, s2, s3;                       // sector b = half (b & 1) of the line
            if (b & 1) { s0 = line[4]; s1 = line[5]; s2 = line[6]; s3 = line[7]; }
            else { s0 = line[0]; s1 = line[1]; s2 = line[2]; s3 = line[3]; }
            w[8] = bswap32(s0.x); w[9] = bswap32(s0.y); w[10] = bswap32(s0.z); w[11] = bswap32(s0.w);
            w[12] = bswap32(s1.x); w[13] = bswap32(s1.y); w[14] = bswap32(s1.z); w[15] = bswap32(s1.w);
            carry[0] = bswap32(s2.x); carry[1] = bswap32(s2.y); carry[2] = bswap32(s2.z); carry[3] = bswap32(s2.w);
            carry[4] = bswap32(s3.x); carry[5] = bswap32(s3.y); carry[6] = bswap32(s3.z); carry[7] = bswap32(s3.w);
        } else if (b == 2048) {                         // ... | commitment[0..32)
            for (int k = 0; k < 8; k++) w[8 + k] = load_be32(cm + 4 * k);
        } else {                                        // commitment[32..48) | 0x80 | zeros | bit length
            for (int k = 0; k < 4; k++) w[k] = load_be32(cm + 32 + 4 * k);
            w[4] = 0x80000000u;
            for (int k = 5; k < 15; k++) w[k] = 0;
            w[15] = (uint32_t)((32 + BLOB_BYTES + 48) * 8);
        }
    }
};

__device__ __forceinline__ void challenge_finish(const uint32_t hh[8], int i, const uint8_t *cm, const uint8_t *proofs, Fr *z_out, Fr *zpow_out,
        uint8_t *records) {
    // hash_to_bls_field (utils.rs:250-258): big-endian integer reduced mod r
    const uint32_t dw[8] = {hh[7], hh[6], hh[5], hh[4], hh[3], hh[2], hh[1], hh[0]};
    Fr z; fr_from_words(z, dw);
    z_out[i] = z;
    if (zpow_out) {                                   // z^2, z^4, ..., z^4096 for the levels of k_eval's tree: twelve squarings on the lane that
        Fr p = z;                                     // has z anyway, instead of twelve on every wave of the evaluation
#pragma unroll 1
        for (int k = 0; k < EVAL_ZPOWERS; k++) { fr_sqr(p, p); zpow_out[EVAL_ZPOWERS * (size_t)i + k] = p; }
    }
    if (!records) return;
    uint8_t *rec = records + (size_t)RECORD_BYTES * i;
    for (int k = 0; k < 48; k++) rec[k] = cm[k];
    uint8_t zb[32]; fr_to_be32(zb, z);
    for (int k = 0; k < 32; k++) rec[48 + k] = zb[k];
    const uint8_t *pr = proofs + 48 * (size_t)i;
    for (int k = 0; k < 48; k++) rec[112 + k] = pr[k];
}

__global__ void __launch_bounds__(128) k_challenge(const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, int n_total,
                                                    Fr *z_out, Fr *zpow_out, uint8_t *records) {
    __shared__ uint4 wk[2][16][64];                 // [buffer][t/4][lane] -> W[t..t+3] + K[t..t+3]
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int i_raw = blockIdx.x * 64 + lane;
    const int i = i_raw < n_total ? i_raw : n_total - 1;           // tail lanes redo the last blob (no out-of-bounds loads)
    const uint4 *blob = reinterpret_cast<const uint4 *>(blobs + (size_t)BLOB_BYTES * i);
    const uint8_t *cm = commitments + 48 * (size_t)i;
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    ChallengeReader rd; rd.blob = blob; rd.cm = cm;
    auto produce = [&](int b) {
        uint32_t w[16];
        rd.words(w, b);
        rd.prefetch(b + 1);                          // flies during the schedule expansion below
        uint4 *dst = &wk[b & 1][0][lane];
#pragma unroll
        for (int t = 0; t < 64; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int tt = t + u;
                if (tt >= 16) {
                    const uint32_t w15 = w[(tt + 1) & 15], w2 = w[(tt + 14) & 15];
                    const uint32_t s0 = xor3(ror(w15, 7), ror(w15, 18), w15 >> 3);
                    const uint32_t s1 = xor3(ror(w2, 17), ror(w2, 19), w2 >> 10);
                    w[tt & 15] = w[tt & 15] + s0 + w[(tt + 9) & 15] + s1;
                }
            }
            dst[(t >> 2) * 64] = make_uint4(w[t & 15] + K[t], w[(t + 1) & 15] + K[t + 1], w[(t + 2) & 15] + K[t + 2], w[(t + 3) & 15] + K[t + 3]);
        }
    };
    uint32_t h0 = 0x6a09e667u, h1 = 0xbb67ae85u, h2 = 0x3c6ef372u, h3 = 0xa54ff53au, h4 = 0x510e527fu, h5 = 0x9b05688cu, h6 = 0x1f83d9abu, h7 = 0x5be0cd19u;
    if (role == 1) { rd.prefetch(0); produce(0); }
    __syncthreads();
    for (int b = 0; b < CH_BLOCKS; b++) {
        if (role == 0) {
            uint32_t a = h0, bb = h1, c = h2, d = h3, e = h4, f = h5, g = h6, h = h7;
            const uint4 *src = &wk[b & 1][0][lane];
#pragma unroll
            for (int t = 0; t < 64; t += 4) {
                const uint4 q = src[(t >> 2) * 64];
                const uint32_t wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t1 = h + xor3(ror(e, 6), ror(e, 11), ror(e, 25)) + ch3(e, f, g) + wv[u];
                    const uint32_t t2 = xor3(ror(a, 2), ror(a, 13), ror(a, 22)) + maj3(a, bb, c);
                    h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
                }
            }
            h0 += a; h1 += bb; h2 += c; h3 += d; h4 += e; h5 += f; h6 += g; h7 += h;
        } else if (b + 1 < CH_BLOCKS) {
            produce(b + 1);
        }
        __syncthreads();
    }
    if (role != 0 || i_raw >= n_total) return;
    const uint32_t hh[8] = {h0, h1, h2, h3, h4, h5, h6, h7};
    challenge_finish(hh, i, cm, proofs, z_out, zpow_out, records);
}

// The same hash with schedule and rounds in ONE wave.  The two-wave form above halves the dependent chain per block, which is
// what counts while every wave has a SIMD to itself (<= 512 workgroups on 1024 SIMDs); past that its producer and consumer
// share SIMDs, each wave issues at half rate, and the lighter single wave (~1430 instead of 930 + 500 instructions per
// block, no LDS hand-off, no barrier) finishes sooner.
constexpr int CH1W_THREADS = 256;       // four waves per workgroup: one per SIMD of the CU it lands on
__global__ void __launch_bounds__(CH1W_THREADS) k_challenge_1w(const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, int n_total,
                                                      Fr *z_out, Fr *zpow_out, uint8_t *records) {
    const int i_raw = blockIdx.x * CH1W_THREADS + threadIdx.x;
    const int i = i_raw < n_total ? i_raw : n_total - 1;
    const uint4 *blob = reinterpret_cast<const uint4 *>(blobs + (size_t)BLOB_BYTES * i);
    const uint8_t *cm = commitments + 48 * (size_t)i;
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t hh[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    ChallengeReader rd; rd.blob = blob; rd.cm = cm;
    rd.prefetch(0);
#pragma unroll 1
    for (int b = 0; b < CH_BLOCKS; b++) {
        uint32_t w[16];
        rd.words(w, b);
        rd.prefetch(b + 1);                          // next sector's loads fly during the rounds
        uint32_t a = hh[0], bb = hh[1], c = hh[2], d = hh[3], e = hh[4], f = hh[5], g = hh[6], h = hh[7];
#pragma unroll
        for (int t = 0; t < 64; t++) {
            if (t >= 16) {
                const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                const uint32_t s0 = xor3(ror(w15, 7), ror(w15, 18), w15 >> 3);
                const uint32_t s1 = xor3(ror(w2, 17), ror(w2, 19), w2 >> 10);
                w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
            }
            const uint32_t t1 = h + xor3(ror(e, 6), ror(e, 11), ror(e, 25)) + ch3(e, f, g) + (w[t & 15] + K[t]);
            const uint32_t t2 = xor3(ror(a, 2), ror(a, 13), ror(a, 22)) + maj3(a, bb, c);
            h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        hh[0] += a; hh[1] += bb; hh[2] += c; hh[3] += d; hh[4] += e; hh[5] += f; hh[6] += g; hh[7] += h;
    }
    if (i_raw >= n_total) return;
    challenge_finish(hh, i, cm, proofs, z_out, zpow_out, records);
}

// Host-hashed form (host_sha256.h): the digests of the challenge transcripts arrive from the host (32 bytes each, big-endian as
// SHA-256 emits them); what is left is hash_to_bls_field (utils.rs:250-258) and the record's C / z / proof fields.
__global__ void __launch_bounds__(64) k_challenge_from_digest(const uint8_t *digests, const uint8_t *commitments, const uint8_t *proofs, int n_total,
                                                              Fr *z_out, Fr *zpow_out, uint8_t *records) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_total) return;
    uint32_t hh[8];
#pragma unroll
    for (int k = 0; k < 8; k++) hh[k] = load_be32(digests + 32 * (size_t)i + 4 * k);
    challenge_finish(hh, i, commitments + 48 * (size_t)i, proofs, z_out, zpow_out, records);
}

// ------------------------------------------------------------------------------------------------ evaluation
// y = p(z) for a polynomial given by its 4096 evaluations at the (bit-reversed) roots of unity (kzg.rs:346-389).
// The reference evaluates  y = (z^N - 1)/N * sum_i p_i w_i / (z - w_i)  with a 4096-long batch inversion and special-
// cases z == w_i.  eval_core.h turns that into  y = (z Ntop - (z^N - 1) sum_i p_i) / N  with Ntop the numerator of sum_i p_i / (z - w_i)
// over the common denominator: NO inversion and no special case (a polynomial identity; for z = w_m it gives p_m, which is what
// kzg.rs:360-362 returns).  Ntop comes up a binary tree over the domain, two limb products and ONE Montgomery reduction per node; the
// lanes take the nodes three at a time: a group joins four neighbouring children to their grandparent (eval_core.h).
// One wave per blob.  Step it = 0..15 takes the 64 groups 64 it .. 64 it + 63 (8 KiB of the blob), one level-1 group per lane.
// The four children of a level-2 group are four NEIGHBOURING lanes of one step, so a lane parks its result in LDS and after every
// fourth step the 256 parked values are dealt out again, four consecutive ones per lane: 64 level-2 groups, one per lane.  The same
// exchange after the loop gives every lane one level-3 group (its four level-2 results go through the same buffer), and levels
// 4-6 (16, 4, 1 groups) run on wave 0 for the workgroup's four blobs.  y comes out as the canonical integer with no conversion.
// Loads: a lane needs the 128 contiguous bytes of its group, but a load instruction whose lanes are 128 bytes apart touches 64
// cache lines for 1 KiB.  The wave instead moves its 8 KiB per step with 8 fully coalesced global -> LDS loads
// (global_load_lds_dwordx4: no VGPR staging, so the next step's data is in flight during this step's ~1,250 instructions
// without costing registers -- staging it in VGPRs spilled to scratch, and waiting for a scratch reload waits for every older
// load too: 59 % of the wave cycles were s_waitcnt).  The LDS side of such a load is linear (lane L of instruction q lands in
// slot 64 q + L), so the bank-spreading XOR is applied on the global side: slot 8 g + s receives chunk 8 g + (s ^ (g & 7)) of
// the tile (still the same 1 KiB per instruction), and lane g reads its part j back from slot 8 g + (j ^ (g & 7)).
// The exchange buffer holds 256 values of 9 limbs, entry e at words 9 e .. 9 e + 8: a lane's writes are 9 words apart (odd:
// conflict-free), its four children are 36 consecutive words read as nine b128 (lanes 36 words apart: 16 lanes cover all banks).
__device__ __forceinline__ void eval_issue_tile_loads(const uint4 *blob_step, uint4 *tile, int lane) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int g = 8 * q + (lane >> 3), s = lane & 7;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(blob_step + 8 * g + (s ^ (g & 7))),
                                         (void __attribute__((address_space(3))) *)(tile + 64 * q), 16, 0, 0);
    }
}
// the pieces of group e (eval_core.h: 64 consecutive groups side by side, one 1 KiB row per load for a wave that takes 64 consecutive groups)
__device__ __forceinline__ void eval_tab_load(EvalPiece p[7], const EvalPiece *tab, int e) {
    const uint4 *src = reinterpret_cast<const uint4 *>(tab);
#pragma unroll
    for (int q = 0; q < 7; q++) { const uint4 v = src[eval_tab_piece(e, q)]; p[q].w[0] = v.x; p[q].w[1] = v.y; p[q].w[2] = v.z; p[q].w[3] = v.w; }
}
__device__ __forceinline__ void eval_park(uint32_t *hx, int entry, const Fr &h) {
#pragma unroll
    for (int k = 0; k < NFR; k++) hx[NFR * entry + k] = h.l[k];
}
// the four consecutive entries 4 node .. 4 node + 3, after every lane's eval_park has landed
__device__ __forceinline__ void eval_children(Fr c[4], const uint32_t *hx, int node) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint4 *src = reinterpret_cast<const uint4 *>(hx + 4 * NFR * node);
    uint32_t w[4 * NFR];
#pragma unroll
    for (int q = 0; q < NFR; q++) { const uint4 v = src[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int k = 0; k < NFR; k++) c[e].l[k] = w[NFR * e + k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();       // all reads done before the buffer is written again
}
// EVAL_WAVES blobs per workgroup, one wave each, with its own tile and exchange buffer; the waves meet once, after level 3.
__global__ void __launch_bounds__(64 * EVAL_WAVES, 2) k_eval(const uint8_t *blobs, const Fr *z_in, const Fr *zpow, const EvalPiece *tab, int n_total,
                                                          int n_per_group, Fr *y_out, uint8_t *records, int *err) {
    __shared__ uint4 tiles[EVAL_WAVES][512];
    __shared__ __attribute__((aligned(16))) uint32_t hxs[EVAL_WAVES][256 * NFR];
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int blob_raw = blockIdx.x * EVAL_WAVES + wid;
    const int blob_i = blob_raw < n_total ? blob_raw : n_total - 1;          // a tail wave redoes the last blob (no out-of-bounds loads) and reports nothing
    uint4 *tile = tiles[wid];
    uint32_t *hx = hxs[wid];
    const uint4 *blob = reinterpret_cast<const uint4 *>(blobs + (size_t)BLOB_BYTES * blob_i);
    eval_issue_tile_loads(blob, tile, lane);
    const Fr z = z_in[blob_i];
    const Fr *zp = zpow + EVAL_ZPOWERS * (size_t)blob_i;          // zp[k - 1] = z^(2^k), k = 1 .. 12 (k_challenge*)
    const Fr z2 = zp[0];
    // z^4, z^8 (level 2, after every fourth step) and z^16, z^32 (level 3) wait in LDS: 18 more live registers spilled, and a global load at the
    // point of use is a full memory latency in front of every level-2 group
    __shared__ uint32_t zls[EVAL_WAVES][4 * NFR];
    const uint32_t zl_word = reinterpret_cast<const uint32_t *>(zp + 1)[lane < 4 * NFR ? lane : 0];       // parked behind the first step's wait for its tile
    Fr h2[4], Sp = fr_zero();
    bool bad = false;
    constexpr int STEPS = N_FE / 4 / 64;
    EvalPiece g_next[7];
    eval_tab_load(g_next, tab, EVAL_TAB_L1 + lane);
#pragma unroll 1
    for (int it = 0; it < STEPS; it++) {
        __builtin_amdgcn_s_waitcnt(0);                            // this step's tile (and table entry) have landed
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (it == 0 && lane < 4 * NFR) zls[wid][lane] = zl_word;
        uint4 cur[8];
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = tile[8 * lane + (j ^ (lane & 7))];
        EvalGroup g1; eval_group_unpack(g1, g_next);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();   // every lane has its 128 bytes
        if (it + 1 < STEPS) eval_issue_tile_loads(blob + 512 * (it + 1), tile, lane);       // next step's loads fly during this one's products
        // the roots of the NEXT piece of work -- the level-2 group after every fourth step, else the next step's group -- fly during this one's
        // (unconditional: a load under a branch made the compiler wait for it on the spot to merge the two paths' registers)
        const int e_l1 = EVAL_TAB_L1 + ((it + 1) & (STEPS - 1)) * 64 + lane, e_l2 = EVAL_TAB_L2 + 64 * (it >> 2) + lane;
        eval_tab_load(g_next, tab, (it & 3) == 3 ? e_l2 : e_l1);
        uint32_t pw[4][8];
#pragma unroll
        for (int e = 0; e < 4; e++) {                             // big-endian 32 bytes -> 8 little-endian words
            const uint4 a = cur[2 * e], b = cur[2 * e + 1];
            pw[e][7] = bswap32(a.x); pw[e][6] = bswap32(a.y); pw[e][5] = bswap32(a.z); pw[e][4] = bswap32(a.w);
            pw[e][3] = bswap32(b.x); pw[e][2] = bswap32(b.y); pw[e][1] = bswap32(b.z); pw[e][0] = bswap32(b.w);
        }
        // bytes_to_bls_field (utils.rs:267-271): value < r.  The top word settles it unless it EQUALS r's top word.
        const uint32_t top = max(max(pw[0][7], pw[1][7]), max(pw[2][7], pw[3][7]));
        if (top >= FR_MOD_TOP_WORD) {
#pragma unroll 1
            for (int e = 0; e < 4; e++) bad = bad || !fr_words_canonical(pw[e]);
        }
        Fr h;
        eval_group_leaves(h, Sp, pw, z, z2, g1);
        eval_park(hx, 64 * (it & 3) + lane, h);                   // group 64 it + lane = entry 64 (it & 3) + lane of this batch of four steps
        if ((it & 3) == 3) {                                      // level 2: group 64 (it >> 2) + lane = entries 4 lane .. 4 lane + 3
            Fr c[4];
            eval_children(c, hx, lane);
            EvalGroup g2; eval_group_unpack(g2, g_next);
            eval_tab_load(g_next, tab, it + 1 < STEPS ? e_l1 : EVAL_TAB_L3 + lane);        // the next step's group, or level 3's behind the last step
            Fr z4, z8;
#pragma unroll
            for (int k = 0; k < NFR; k++) { z4.l[k] = zls[wid][k]; z8.l[k] = zls[wid][NFR + k]; }
            Fr r2; eval_group(r2, c, z4, z8, g2);
#pragma unroll
            for (int a = 0; a < 4; a++) if ((it >> 2) == a) h2[a] = r2;
        }
    }
    if (bad && blob_raw < n_total) atomicOr(&err[blob_i / n_per_group], ERR_NONCANONICAL_FR);
    Fr c[4], h;
    // level 3: group n = lane, children the level-2 groups 4 lane .. 4 lane + 3 (group 64 a + l sits in lane l's h2[a])
#pragma unroll
    for (int a = 0; a < 4; a++) eval_park(hx, 64 * a + lane, h2[a]);
    eval_children(c, hx, lane);
    EvalGroup g3; eval_group_unpack(g3, g_next);
    Fr z16, z32;
#pragma unroll
    for (int k = 0; k < NFR; k++) { z16.l[k] = zls[wid][2 * NFR + k]; z32.l[k] = zls[wid][3 * NFR + k]; }
    eval_group(h, c, z16, z32, g3);
    // What wave 0 will need for levels 4-6 of the workgroup's four blobs (blob b = lane / 16): requested here, by every wave (no branch around a
    // load: k_eval's header), so that the answers arrive under the sum below and the barrier instead of in front of each level
    const int b = lane >> 4, u = lane & 15;                       // blob b of the workgroup, level-4 node u
    const int mine_raw = blockIdx.x * EVAL_WAVES + b, mine = mine_raw < n_total ? mine_raw : n_total - 1;
    const Fr *zq = zpow + EVAL_ZPOWERS * (size_t)mine;
    Fr zt[7];
#pragma unroll
    for (int k = 0; k < 7; k++) zt[k] = zq[5 + k];                // z^64 .. z^4096
    const Fr zmine = z_in[mine];
    EvalPiece gp4[7], gp5[7], gp6[7];
    eval_tab_load(gp4, tab, EVAL_TAB_L4 + u); eval_tab_load(gp5, tab, EVAL_TAB_L5 + (lane & 3)); eval_tab_load(gp6, tab, EVAL_TAB_L6);
    // the blob's sum of values: every lane's 64 folded below 3.1 r, then added across the wave (< 200 r)
    eval_fold(Sp);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const Fr o = fr_shfl_down(Sp, off); fr_add_lazy(Sp, Sp, o); }
    // Levels 4, 5, 6 have 16, 4 and 1 nodes per blob: the workgroup's four blobs share ONE wave for them, 16 lanes per blob (a wave
    // on its own would run each of them on all 64 lanes for 16, 4 and 1 distinct results).
    eval_park(hx, lane, h);                                       // this blob's 64 level-3 results: entries 0..63 of its own buffer
    if (lane == 0) eval_park(hx, 64, Sp);                         // and its sum of values: entry 64
    __syncthreads();
    if (wid != 0) return;
    Fr Sb;
#pragma unroll
    for (int k = 0; k < NFR; k++) Sb.l[k] = hxs[b][NFR * 64 + k];
    eval_children(c, hxs[b], u);
    EvalGroup gt; eval_group_unpack(gt, gp4);
    eval_group(h, c, zt[0], zt[1], gt);
    uint32_t *hx0 = hxs[0];                                       // from here on wave 0's buffer, read only by wave 0: entry 16 b + u, then 4 b + v
    eval_park(hx0, lane, h);
    eval_children(c, hx0, 4 * b + (lane & 3));
    eval_group_unpack(gt, gp5);
    eval_group(h, c, zt[2], zt[3], gt);
    if (u < 4) eval_park(hx0, 4 * b + u, h);
    eval_children(c, hx0, b);
    eval_group_unpack(gt, gp6);
    eval_group(h, c, zt[4], zt[5], gt);
    Fr y; eval_finish(y, h, Sb, zmine, zt[6]);                    // canonical integer value of y
    if (u == 0 && mine_raw < n_total) {
        if (records) {
            uint32_t yw[8]; limbs_to_words<NFR, 8>(yw, y.l);
            uint8_t *rec = records + (size_t)RECORD_BYTES * mine + 80;
#pragma unroll
            for (int i = 0; i < 8; i++) {                         // big-endian bytes, straight from registers
                const uint32_t v = yw[7 - i];
                rec[4 * i] = (uint8_t)(v >> 24); rec[4 * i + 1] = (uint8_t)(v >> 16); rec[4 * i + 2] = (uint8_t)(v >> 8); rec[4 * i + 3] = (uint8_t)v;
            }
        }
        if (y_out) {                                              // Montgomery form for callers that want an Fr
            const uint32_t r2[NFR] = FR_R2_INIT;
            Fr R2; for (int i = 0; i < NFR; i++) R2.l[i] = r2[i];
            Fr ym; fr_mul(ym, y, R2);
            y_out[mine] = ym;
        }
    }
}

// ------------------------------------------------------------------------------------------------ r powers
// One wave per batch.  Transcript: "RCKZGBATCH___V1_" | u64be(4096) | u64be(n) | n records  (utils.rs:439-463).
// Emits, as plain 256-bit integers (8 LE words): a_i = r^i, b_i = r^i z_i, and c = sum r^i y_i.
// The hash is one serial chain per batch, but the message schedule of a block does not depend on the chaining state: the 64
// lanes expand 64 blocks at once (W[t] + K[t] to LDS), then lane 0 runs the rounds of those blocks (~930 instructions each,
// the irreducible chain).  The powers are spread over the lanes: lane l starts at r^l and steps by r^64.
__constant__ uint32_t SHA_K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
// Many batches: the same transcript hash with one LANE per batch (schedule and rounds in the lane's registers), 64 batches per
// wave.  The one-wave-per-batch form above keeps 63 lanes idle during the rounds: fine while every wave has a SIMD to itself,
// 64x the issue slots once the batches outnumber the SIMDs.  Writes the digest words (as fr_from_words takes them) to
// digests[g][8]; k_rpowers then starts from those.
__global__ void __launch_bounds__(64) k_rhash_lanes(const uint8_t *records, int n, int groups, uint32_t *digests, int n_fe) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= groups) return;
    const uint8_t *rec = records + (size_t)RECORD_BYTES * n * g;
    const uint32_t total_words = 8u + 40u * (uint32_t)n;
    const uint32_t nblocks = (total_words * 4u + 9u + 63u) / 64u;
    const uint64_t bits = (uint64_t)total_words * 32u;
    uint32_t hs[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
#pragma unroll 1
    for (uint32_t b = 0; b < nblocks; b++) {
        uint32_t w[16];
        if (b >= 1u && 16u * b + 16u <= total_words) {                  // a block of record bytes only: four 16-byte loads
            const uint4 *p = reinterpret_cast<const uint4 *>(rec + 64u * (size_t)b - 32u);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = p[q];
                w[4 * q] = bswap32(v.x); w[4 * q + 1] = bswap32(v.y); w[4 * q + 2] = bswap32(v.z); w[4 * q + 3] = bswap32(v.w);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const uint32_t idx = 16u * b + (uint32_t)t;
                uint32_t v;
                if (idx < 8u) {
                    const uint32_t hdr[8] = {0x52434b5au, 0x47424154u, 0x43485f5fu, 0x5f56315fu, 0u, (uint32_t)n_fe, 0u, (uint32_t)n};
                    v = hdr[idx];
                } else if (idx < total_words) v = bswap32(reinterpret_cast<const uint32_t *>(rec)[idx - 8u]);
                else if (idx == total_words) v = 0x80000000u;
                else if (idx == 16u * nblocks - 2u) v = (uint32_t)(bits >> 32);
                else if (idx == 16u * nblocks - 1u) v = (uint32_t)bits;
                else v = 0u;
                w[t] = v;
            }
        }
        uint32_t a = hs[0], bb = hs[1], c = hs[2], d = hs[3], e = hs[4], f = hs[5], gg = hs[6], h = hs[7];
#pragma unroll
        for (int t = 0; t < 64; t++) {
            if (t >= 16) {
                const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                const uint32_t s0 = xor3(ror(w15, 7), ror(w15, 18), w15 >> 3);
                const uint32_t s1 = xor3(ror(w2, 17), ror(w2, 19), w2 >> 10);
                w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
            }
            const uint32_t t1 = h + xor3(ror(e, 6), ror(e, 11), ror(e, 25)) + ch3(e, f, gg) + w[t & 15] + SHA_K[t];
            const uint32_t t2 = xor3(ror(a, 2), ror(a, 13), ror(a, 22)) + maj3(a, bb, c);
            h = gg; gg = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        hs[0] += a; hs[1] += bb; hs[2] += c; hs[3] += d; hs[4] += e; hs[5] += f; hs[6] += gg; hs[7] += h;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) digests[8 * (size_t)g + k] = hs[7 - k];
}

// Four batches (waves) per workgroup, one per SIMD of the CU it lands on (one-wave workgroups of long chains are placed
// unevenly); the waves share nothing: each keeps to its own LDS slice behind wave-local fences.
#define RP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
__global__ void __launch_bounds__(256) k_rpowers(const uint8_t *records, int n, int groups, int check_zy, uint32_t *scal_a, uint32_t *scal_b,
                                                  uint32_t *scal_c, int *err, int n_fe, int have_digest) {
    __shared__ uint32_t wk_all[4][64][64];           // per wave: [t][block of the chunk]
    __shared__ uint32_t digest_all[4][8];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g_raw = blockIdx.x * (int)(blockDim.x >> 6) + wid;
    if (g_raw >= groups) return;                                  // (the waves only synchronise with themselves)
    const int g = g_raw;
    const bool live = true;
    uint32_t (*wk)[64] = wk_all[wid];
    uint32_t *digest = digest_all[wid];
    const uint8_t *rec = records + (size_t)RECORD_BYTES * n * g;
    Fr r = fr_one();
    if (n > 1 && have_digest) {                                        // k_rhash_lanes left the digest where c goes (read here, overwritten at the end)
        uint32_t dw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) dw[k] = scal_c[8 * (size_t)g + k];
        fr_from_words(r, dw);
    } else if (n > 1) {   // for n == 1 only r^0 = 1 is used (the reference takes the single-proof path, kzg.rs:658-660)
        const uint32_t total_words = 8u + 40u * (uint32_t)n;            // message length / 4
        const uint32_t nblocks = (total_words * 4u + 9u + 63u) / 64u;
        const uint64_t bits = (uint64_t)total_words * 32u;
        uint32_t h0 = 0x6a09e667u, h1 = 0xbb67ae85u, h2 = 0x3c6ef372u, h3 = 0xa54ff53au, h4 = 0x510e527fu, h5 = 0x9b05688cu, h6 = 0x1f83d9abu, h7 = 0x5be0cd19u;
        for (uint32_t b0 = 0; b0 < nblocks; b0 += 64) {
            const uint32_t b = b0 + (uint32_t)lane;
            if (b < nblocks) {                                          // this lane expands block b
                uint32_t w[16];
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const uint32_t idx = 16u * b + (uint32_t)t;
                    uint32_t v;
                    if (idx < 8u) {
                        const uint32_t hdr[8] = {0x52434b5au, 0x47424154u, 0x43485f5fu, 0x5f56315fu, 0u, (uint32_t)n_fe, 0u, (uint32_t)n};
                        v = hdr[idx];                                   // "RCKZGBATCH___V1_" | 4096 | n
                    } else if (idx < total_words) v = bswap32(reinterpret_cast<const uint32_t *>(rec)[idx - 8u]);   // records are 4-byte aligned
                    else if (idx == total_words) v = 0x80000000u;
                    else if (idx == 16u * nblocks - 2u) v = (uint32_t)(bits >> 32);
                    else if (idx == 16u * nblocks - 1u) v = (uint32_t)bits;
                    else v = 0u;
                    w[t] = v;
                }
#pragma unroll
                for (int t = 0; t < 64; t++) {
                    if (t >= 16) {
                        const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                        const uint32_t s0 = xor3(ror(w15, 7), ror(w15, 18), w15 >> 3);
                        const uint32_t s1 = xor3(ror(w2, 17), ror(w2, 19), w2 >> 10);
                        w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
                    }
                    wk[t][lane] = w[t & 15] + SHA_K[t];
                }
            }
            RP_WAVE_SYNC();
            if (lane == 0) {
                const uint32_t cnt = nblocks - b0 < 64u ? nblocks - b0 : 64u;
#pragma unroll 1
                for (uint32_t q = 0; q < cnt; q++) {
                    uint32_t a = h0, bb = h1, c = h2, d = h3, e = h4, f = h5, gg = h6, h = h7;
#pragma unroll
                    for (int t = 0; t < 64; t++) {
                        const uint32_t t1 = h + xor3(ror(e, 6), ror(e, 11), ror(e, 25)) + ch3(e, f, gg) + wk[t][q];
                        const uint32_t t2 = xor3(ror(a, 2), ror(a, 13), ror(a, 22)) + maj3(a, bb, c);
                        h = gg; gg = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
                    }
                    h0 += a; h1 += bb; h2 += c; h3 += d; h4 += e; h5 += f; h6 += gg; h7 += h;
                }
            }
            RP_WAVE_SYNC();
        }
        if (lane == 0) { digest[0] = h7; digest[1] = h6; digest[2] = h5; digest[3] = h4; digest[4] = h3; digest[5] = h2; digest[6] = h1; digest[7] = h0; }
        RP_WAVE_SYNC();
        uint32_t dw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) dw[k] = digest[k];
        fr_from_words(r, dw);                                           // hash_to_bls_field (utils.rs:472)
    }
    // lane l: r^l by square-and-multiply over its 6 index bits; step r^64
    Fr pw = fr_one(), r64 = r;
#pragma unroll 1
    for (int bit = 5; bit >= 0; bit--) {
        Fr t;
        fr_sqr(pw, pw);
        fr_mul(t, pw, r);
        fr_select(pw, (lane >> bit) & 1, pw, t);
        fr_sqr(r64, r64);
    }
    Fr csum = fr_zero();
    bool bad = false;
#pragma unroll 1
    for (int i = lane; i < n; i += 64) {
        uint32_t zw[8], yw[8];
        be32_to_words(zw, rec + (size_t)RECORD_BYTES * i + 48);
        be32_to_words(yw, rec + (size_t)RECORD_BYTES * i + 80);
        if (check_zy) bad = bad || !fr_words_canonical(zw) || !fr_words_canonical(yw);
        Fr z, y, t; fr_from_words(z, zw); fr_from_words(y, yw);
        uint32_t *pa = scal_a + 8 * ((size_t)g * n + i), *pb = scal_b + 8 * ((size_t)g * n + i);
        uint32_t ow[8];
        fr_to_words(ow, pw); if (live) for (int k = 0; k < 8; k++) pa[k] = ow[k];
        fr_mul(t, pw, z); fr_to_words(ow, t); if (live) for (int k = 0; k < 8; k++) pb[k] = ow[k];
        fr_mul(t, pw, y); fr_add(csum, csum, t);
        fr_mul(pw, pw, r64);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { Fr o = fr_shfl_down(csum, off); fr_add(csum, csum, o); }
    if (lane == 0 && live) {
        uint32_t ow[8]; fr_to_words(ow, csum);
        for (int k = 0; k < 8; k++) scal_c[8 * (size_t)g + k] = ow[k];
    }
    if (bad && live) atomicOr(&err[g], ERR_NONCANONICAL_FR);
}

// ------------------------------------------------------------------------------------------------ pairing
// One lane per batch:  ML([tau]G2, -proof_lincomb) * ML(G2, rhs)  ->  final exponentiation  ->  == 1 ?
__global__ void __launch_bounds__(64) k_pairing(const PairPt *pair_pts, const LineCoeff *lines, const int *lines_inf, int groups, int *ok) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    // (this A/B kernel works on affine points)
    G1Affine p1, p2; pairpt_to_affine(p1, pair_pts[2 * (size_t)g]); pairpt_to_affine(p2, pair_pts[2 * (size_t)g + 1]);
    if (lines_inf[2]) p1 = g1a_inf();          // e(P, infinity) = 1
    if (lines_inf[0]) p2 = g1a_inf();
    Fp12 f;
    miller_loop_pair(f, lines + 2 * N_LINES, p1, lines, p2);     // lines[2] = setup g2[1] = [tau]G2 ; lines[0] = G2 generator
    ok[g] = final_exp_is_one(f) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_challenges(const uint8_t *d_blobs, const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, Fr *d_z, Fr *d_zpow, uint8_t *d_records,
                       hipStream_t st, int form) {
    if (n_total <= 0) return;
    const int wgs = (n_total + 63) / 64;
    if (form == 2 || (form == 0 && wgs <= 512)) hipLaunchKernelGGL(k_challenge, dim3(wgs), dim3(128), 0, st, d_blobs, d_commitments, d_proofs, n_total, d_z,
            d_zpow, d_records);
    else hipLaunchKernelGGL(k_challenge_1w, dim3((n_total + CH1W_THREADS - 1) / CH1W_THREADS), dim3(CH1W_THREADS), 0, st, d_blobs, d_commitments, d_proofs,
            n_total, d_z, d_zpow, d_records);
}
void launch_challenges_from_digests(const uint8_t *d_digests, const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, Fr *d_z, Fr *d_zpow,
                                    uint8_t *d_records, hipStream_t st) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_challenge_from_digest, dim3((n_total + 63) / 64), dim3(64), 0, st, d_digests, d_commitments, d_proofs, n_total, d_z, d_zpow,
            d_records);
}
void launch_eval(const uint8_t *d_blobs, const Fr *d_z, const Fr *d_zpow, DeviceTables t, int n_total, int n_per_group, Fr *d_y, uint8_t *d_records, int *d_err,
                 hipStream_t st) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_eval, dim3((n_total + EVAL_WAVES - 1) / EVAL_WAVES), dim3(64 * EVAL_WAVES), 0, st, d_blobs, d_z, d_zpow, t.eval_tab, n_total,
            n_per_group,
                       d_y, d_records, d_err);
}
void launch_rpowers(const uint8_t *d_records, int n_per_group, int groups, int check_zy, uint32_t *d_scal_a, uint32_t *d_scal_b,
                    uint32_t *d_scal_c, int *d_err, hipStream_t st, int n_fe, int lanes_from, int have_digest) {
    if (groups <= 0) return;
    // four waves per workgroup once there are more batches than CUs can take one each (even placement); below that one wave per
    // workgroup: four of these LDS-latency-bound single-lane chains on one CU slow each other down (512-blob batches: 3x)
    const int wpw = groups > 512 ? 4 : 1;
    // from one wave per SIMD on (lanes_from, default 1024 batches), hash with a lane per batch first (k_rhash_lanes)
    const int lanes = (!have_digest && n_per_group > 1 && groups >= lanes_from) ? 1 : 0;
    if (lanes) hipLaunchKernelGGL(k_rhash_lanes, dim3((groups + 63) / 64), dim3(64), 0, st, d_records, n_per_group, groups, d_scal_c, n_fe);
    hipLaunchKernelGGL(k_rpowers, dim3((groups + wpw - 1) / wpw), dim3(64 * wpw), 0, st, d_records, n_per_group, groups, check_zy, d_scal_a, d_scal_b,
            d_scal_c, d_err, n_fe, lanes || have_digest);
}
void launch_pairing_lane(const PairPt *d_pair_pts, DeviceTables t, int groups, int *d_ok, hipStream_t st) {
    if (groups <= 0) return;
    hipLaunchKernelGGL(k_pairing, dim3((groups + 63) / 64), dim3(64), 0, st, d_pair_pts, t.lines, t.lines_inf, groups, d_ok);
}

}  // namespace kzg
