// kzg355.hpp -- header-only C++ mirror of the reference's public surface over the C ABI (include/kzg355.h).
//
// The reference is compiled code (Rust): `pub struct Kzg` with eight associated functions plus the byte newtypes
// (pawanjay176/kzg_rust src/kzg.rs:10-22, 88-279, 983-1079).  This header restates that surface for C++ callers with
// the same names, argument meaning and error behaviour, so a test written against the reference reads the same here:
//
//     auto s = kzg355::Kzg::load_trusted_setup_file("trusted_setup.txt");          // Result<KzgSettings>
//     auto c = kzg355::Kzg::blob_to_kzg_commitment(blob, s.value());                // Result<KzgCommitment>
//     auto ok = kzg355::Kzg::verify_blob_kzg_proof_batch(blobs, commitments, proofs, s.value());   // Result<bool>
//
// `Result<T>` is Ok(T) or Err(Error) like Rust's; all arithmetic happens in libkzg355.so on the GPU (no CPU fallback).
#pragma once
#include <cstring>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "kzg355.h"

namespace kzg355 {

constexpr size_t BYTES_PER_FIELD_ELEMENT = KZG355_BYTES_PER_FIELD_ELEMENT;   // consts.rs:5
constexpr size_t BYTES_PER_COMMITMENT = KZG355_BYTES_PER_COMMITMENT;         // consts.rs:8
constexpr size_t BYTES_PER_PROOF = KZG355_BYTES_PER_PROOF;                   // consts.rs:11
constexpr size_t FIELD_ELEMENTS_PER_BLOB = KZG355_FIELD_ELEMENTS_PER_BLOB;   // consts.rs:13
constexpr size_t BYTES_PER_BLOB = KZG355_BYTES_PER_BLOB;                     // consts.rs:16

// enum Error (kzg.rs:10-22)
struct Error {
    enum Kind { BadArgs = 1, InternalError = 2, InvalidBytesLength = 3, InvalidHexFormat = 4, InvalidTrustedSetup = 5, NoDevice = 6, NoMemory = 7, DeviceError = 8 } kind;
    std::string message;
};

template <class T> class Result {
    bool ok_;
    T value_{};
    Error err_{Error::InternalError, ""};
public:
    Result(T v) : ok_(true), value_(std::move(v)) {}
    Result(Error e) : ok_(false), err_(std::move(e)) {}
    bool is_ok() const { return ok_; }
    bool is_err() const { return !ok_; }
    T &value() { return value_; }
    const T &value() const { return value_; }
    const Error &error() const { return err_; }
    T unwrap() && { if (!ok_) throw std::runtime_error("unwrap on Err: " + err_.message); return std::move(value_); }
};

inline Error from_status(int rc, const char *what) {
    Error::Kind k = (rc >= 1 && rc <= 8) ? (Error::Kind)rc : Error::InternalError;
    return Error{k, std::string(what) + ": status " + std::to_string(rc)};
}

// hex_to_bytes (kzg.rs:82-86): with or without the 0x prefix
inline Result<std::vector<uint8_t>> hex_to_bytes(const std::string &hex_str) {
    size_t off = hex_str.rfind("0x", 0) == 0 ? 2 : 0;
    size_t n = hex_str.size() - off;
    if (n % 2) return Error{Error::InvalidHexFormat, "Failed to decode hex: odd length"};
    std::vector<uint8_t> out(n / 2);
    auto val = [](char ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; };
    for (size_t i = 0; i < n / 2; i++) {
        int hi = val(hex_str[off + 2 * i]), lo = val(hex_str[off + 2 * i + 1]);
        if (hi < 0 || lo < 0) return Error{Error::InvalidHexFormat, "Failed to decode hex: invalid character"};
        out[i] = (uint8_t)(hi * 16 + lo);
    }
    return out;
}

template <size_t N, Error::Kind LENGTH_ERROR> struct FixedBytes {
    std::vector<uint8_t> bytes = std::vector<uint8_t>(N);
    static Result<FixedBytes> from_bytes(const uint8_t *b, size_t len) {
        if (len != N) return Error{LENGTH_ERROR, "Invalid byte length. Expected " + std::to_string(N) + " got " + std::to_string(len)};
        FixedBytes r; std::memcpy(r.bytes.data(), b, N); return r;
    }
    static Result<FixedBytes> from_bytes(const std::vector<uint8_t> &b) { return from_bytes(b.data(), b.size()); }
    static Result<FixedBytes> from_hex(const std::string &h) {
        auto b = hex_to_bytes(h);
        if (b.is_err()) return b.error();
        return from_bytes(b.value());
    }
    const uint8_t *data() const { return bytes.data(); }
    bool operator==(const FixedBytes &o) const { return bytes == o.bytes; }
};
using Bytes32 = FixedBytes<32, Error::BadArgs>;                         // kzg.rs:101-122 (length error is BadArgs)
using Bytes48 = FixedBytes<48, Error::InvalidBytesLength>;              // kzg.rs:124-152
using Blob = FixedBytes<BYTES_PER_BLOB, Error::InvalidBytesLength>;     // kzg.rs:154-178
using KzgCommitment = Bytes48;                                          // kzg.rs:180-191
using KzgProof = Bytes48;                                               // kzg.rs:193-204

// KzgSettings (kzg.rs:28-40): opaque device-resident handle, Drop frees it, shareable between threads
class KzgSettings {
    std::shared_ptr<kzg355_settings> h_;
public:
    KzgSettings() = default;
    explicit KzgSettings(kzg355_settings *raw) : h_(raw, [](kzg355_settings *p) { kzg355_free_trusted_setup(p); }) {}
    const kzg355_settings *raw() const { return h_.get(); }
    // KzgSettings::load_trusted_setup (kzg.rs:45-78)
    static Result<KzgSettings> load_trusted_setup(const std::vector<std::vector<uint8_t>> &g1, const std::vector<std::vector<uint8_t>> &g2) {
        std::vector<uint8_t> a, b;
        for (auto &x : g1) { if (x.size() != 48) return Error{Error::InvalidBytesLength, "g1 point length"}; a.insert(a.end(), x.begin(), x.end()); }
        for (auto &x : g2) { if (x.size() != 96) return Error{Error::InvalidBytesLength, "g2 point length"}; b.insert(b.end(), x.begin(), x.end()); }
        kzg355_settings *raw = nullptr;
        int rc = kzg355_load_trusted_setup(a.data(), g1.size(), b.data(), g2.size(), &raw);
        if (rc) return from_status(rc, "load_trusted_setup");
        return KzgSettings(raw);
    }
};

// pub struct Kzg (kzg.rs:983-1079)
struct Kzg {
    static Result<KzgSettings> load_trusted_setup_file(const std::string &path) {                         // kzg.rs:995
        kzg355_settings *raw = nullptr;
        int rc = kzg355_load_trusted_setup_file(path.c_str(), &raw);
        if (rc) return from_status(rc, "load_trusted_setup_file");
        return KzgSettings(raw);
    }
    static Result<KzgSettings> load_trusted_setup(const std::vector<std::vector<uint8_t>> &g1, const std::vector<std::vector<uint8_t>> &g2) {
        return KzgSettings::load_trusted_setup(g1, g2);                                                   // kzg.rs:1005
    }
    static Result<KzgCommitment> blob_to_kzg_commitment(const Blob &blob, const KzgSettings &s) {         // kzg.rs:1013
        KzgCommitment out;
        int rc = kzg355_blob_to_kzg_commitment(out.bytes.data(), blob.data(), s.raw());
        if (rc) return from_status(rc, "blob_to_kzg_commitment");
        return out;
    }
    static Result<std::pair<KzgProof, Bytes32>> compute_kzg_proof(const Blob &blob, const Bytes32 &z, const KzgSettings &s) {   // kzg.rs:1021
        KzgProof p; Bytes32 y;
        int rc = kzg355_compute_kzg_proof(p.bytes.data(), y.bytes.data(), blob.data(), z.data(), s.raw());
        if (rc) return from_status(rc, "compute_kzg_proof");
        return std::make_pair(p, y);
    }
    static Result<KzgProof> compute_blob_kzg_proof(const Blob &blob, const KzgCommitment &c, const KzgSettings &s) {            // kzg.rs:1030
        KzgProof p;
        int rc = kzg355_compute_blob_kzg_proof(p.bytes.data(), blob.data(), c.data(), s.raw());
        if (rc) return from_status(rc, "compute_blob_kzg_proof");
        return p;
    }
    static Result<bool> verify_kzg_proof(const KzgCommitment &c, const Bytes32 &z, const Bytes32 &y, const KzgProof &p, const KzgSettings &s) {   // kzg.rs:1039
        bool ok = false;
        int rc = kzg355_verify_kzg_proof(&ok, c.data(), z.data(), y.data(), p.data(), s.raw());
        if (rc) return from_status(rc, "verify_kzg_proof");
        return ok;
    }
    static Result<bool> verify_blob_kzg_proof(const Blob &blob, const KzgCommitment &c, const KzgProof &p, const KzgSettings &s) {              // kzg.rs:1050
        bool ok = false;
        int rc = kzg355_verify_blob_kzg_proof(&ok, blob.data(), c.data(), p.data(), s.raw());
        if (rc) return from_status(rc, "verify_blob_kzg_proof");
        return ok;
    }
    static Result<bool> verify_blob_kzg_proof_batch(const std::vector<Blob> &blobs, const std::vector<KzgCommitment> &cs,
                                                    const std::vector<KzgProof> &ps, const KzgSettings &s) {                                    // kzg.rs:1066
        // `&[Blob]` is a slice of separately boxed blobs (kzg.rs:155-157): gather into one staging buffer
        std::vector<uint8_t> b(blobs.size() * BYTES_PER_BLOB), c(cs.size() * 48), p(ps.size() * 48);
        for (size_t i = 0; i < blobs.size(); i++) std::memcpy(&b[i * BYTES_PER_BLOB], blobs[i].data(), BYTES_PER_BLOB);
        for (size_t i = 0; i < cs.size(); i++) std::memcpy(&c[i * 48], cs[i].data(), 48);
        for (size_t i = 0; i < ps.size(); i++) std::memcpy(&p[i * 48], ps[i].data(), 48);
        bool ok = false;
        int rc = kzg355_verify_blob_kzg_proof_batch(&ok, b.data(), blobs.size(), c.data(), cs.size(), p.data(), ps.size(), s.raw());
        if (rc) return from_status(rc, "verify_blob_kzg_proof_batch");
        return ok;
    }

    // ---- throughput extensions (no reference counterpart): many independent single-proof units per call; one Result per unit, an Err of the
    // call itself only for whole-call failures (no device, out of memory, a length mismatch)
    static bool whole_call_failed(int rc, const std::vector<int> &st) {
        if (rc == KZG355_INTERNAL || rc == KZG355_NO_DEVICE || rc == KZG355_NO_MEMORY || rc == KZG355_DEVICE_ERROR) return true;
        if (rc == KZG355_OK) return false;
        for (int x : st) if (x) return false;
        return true;
    }
    static Result<std::vector<Result<bool>>> verify_kzg_proof_many(const std::vector<KzgCommitment> &cs, const std::vector<Bytes32> &zs,
                                                                   const std::vector<Bytes32> &ys, const std::vector<KzgProof> &ps, const KzgSettings &s) {
        const size_t n = cs.size();
        if (zs.size() != n || ys.size() != n || ps.size() != n) return Error{Error::BadArgs, "length mismatch"};
        std::vector<uint8_t> c(n * 48 + 1), z(n * 32 + 1), y(n * 32 + 1), p(n * 48 + 1);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&c[i * 48], cs[i].data(), 48); std::memcpy(&z[i * 32], zs[i].data(), 32);
            std::memcpy(&y[i * 32], ys[i].data(), 32); std::memcpy(&p[i * 48], ps[i].data(), 48);
        }
        std::unique_ptr<bool[]> ok(new bool[n + 1]());
        std::vector<int> st(n, 0);
        int rc = kzg355_verify_kzg_proof_many(ok.get(), st.data(), c.data(), z.data(), y.data(), p.data(), n, s.raw());
        if (whole_call_failed(rc, st)) return from_status(rc, "verify_kzg_proof_many");
        std::vector<Result<bool>> out;
        for (size_t i = 0; i < n; i++) { if (st[i]) out.emplace_back(from_status(st[i], "verify")); else out.emplace_back(ok[i]); }
        return out;
    }
    static Result<std::vector<Result<bool>>> verify_blob_kzg_proof_many(const std::vector<Blob> &blobs, const std::vector<KzgCommitment> &cs,
                                                                        const std::vector<KzgProof> &ps, const KzgSettings &s) {
        const size_t n = blobs.size();
        if (cs.size() != n || ps.size() != n) return Error{Error::BadArgs, "length mismatch"};
        std::vector<uint8_t> b(n * BYTES_PER_BLOB + 1), c(n * 48 + 1), p(n * 48 + 1);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&b[i * BYTES_PER_BLOB], blobs[i].data(), BYTES_PER_BLOB); std::memcpy(&c[i * 48], cs[i].data(), 48);
            std::memcpy(&p[i * 48], ps[i].data(), 48);
        }
        std::unique_ptr<bool[]> ok(new bool[n + 1]());
        std::vector<int> st(n, 0);
        int rc = kzg355_verify_blob_kzg_proof_many(ok.get(), st.data(), b.data(), c.data(), p.data(), n, s.raw());
        if (whole_call_failed(rc, st)) return from_status(rc, "verify_blob_kzg_proof_many");
        std::vector<Result<bool>> out;
        for (size_t i = 0; i < n; i++) { if (st[i]) out.emplace_back(from_status(st[i], "verify")); else out.emplace_back(ok[i]); }
        return out;
    }
    static Result<std::vector<Result<std::pair<KzgProof, Bytes32>>>> compute_kzg_proof_many(const std::vector<Blob> &blobs, const std::vector<Bytes32> &zs,
                                                                                           const KzgSettings &s) {
        const size_t n = blobs.size();
        if (zs.size() != n) return Error{Error::BadArgs, "length mismatch"};
        std::vector<uint8_t> b(n * BYTES_PER_BLOB + 1), z(n * 32 + 1), pr(n * 48 + 1), ys(n * 32 + 1);
        for (size_t i = 0; i < n; i++) { std::memcpy(&b[i * BYTES_PER_BLOB], blobs[i].data(), BYTES_PER_BLOB); std::memcpy(&z[i * 32], zs[i].data(), 32); }
        std::vector<int> st(n, 0);
        int rc = kzg355_compute_kzg_proof_many(pr.data(), ys.data(), st.data(), b.data(), z.data(), n, s.raw());
        if (whole_call_failed(rc, st)) return from_status(rc, "compute_kzg_proof_many");
        std::vector<Result<std::pair<KzgProof, Bytes32>>> out;
        for (size_t i = 0; i < n; i++) {
            if (st[i]) { out.emplace_back(from_status(st[i], "proof")); continue; }
            KzgProof p; Bytes32 y;
            std::memcpy(p.bytes.data(), &pr[i * 48], 48); std::memcpy(y.bytes.data(), &ys[i * 32], 32);
            out.emplace_back(std::make_pair(p, y));
        }
        return out;
    }
};

}  // namespace kzg355
