// host_sha256.h -- SHA-256 on the HOST for the Fiat-Shamir challenges of host-buffer calls (SURVEY 8f-4, section 7 hard part 4).
//
// compute_challenge (reference src/kzg.rs:298-339) hashes 131,152 bytes per blob: 2050 compressions that are strictly sequential.
// On the GPU that chain is 3.7 ms for a lone batch whatever the card does besides (one lane per blob); a host core with the SHA
// extensions walks it in ~60 us, and the blobs of a host-buffer call start out in host memory anyway.  So small host-buffer calls
// hash on a few host threads WHILE the H2D copy and the point kernels run, and upload 32-byte digests; device-resident and large
// calls keep the device kernels (k_challenge*, k_verify.hip).
//
// Written from FIPS 180-4 (and the Intel SHA extensions programming reference for the sha256rnds2 / sha256msg1 / sha256msg2
// data flow); it shares nothing with oracle/ (test infrastructure) or with the device header sha256.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace kzg_host {

enum Sha256Impl { SHA256_AUTO = 0, SHA256_PORTABLE = 1, SHA256_SHANI = 2 };

bool sha256_have_shani();                                  // CPUID leaf 7: SHA + SSSE3 / SSE4.1
// one-shot digest; impl = SHA256_SHANI on a CPU without the extensions falls back to the portable form and returns false
bool sha256(uint8_t out[32], const uint8_t *msg, size_t len, int impl = SHA256_AUTO);
// digests of the challenge transcripts  "FSBLOBVERIFY_V1_" | u64be(0) | u64be(n_fe) | blob | commitment  (kzg.rs:298-339;
// consts.rs:19-22) of `count` blobs, read in place (no 131 KB staging copy); pairs of blobs are hashed interleaved, which hides
// the latency of the dependent round instructions.  blobs: count * blob_bytes contiguous; commitments: count * 48; out: count * 32.
void challenge_digests(uint8_t *out, const uint8_t *blobs, size_t blob_bytes, const uint8_t *commitments, size_t count, uint64_t n_fe,
                       int impl = SHA256_AUTO);

}  // namespace kzg_host
