// cpp_vector_runner -- drives the C++ mirror (include/kzg355.hpp) with test cases read from stdin, one per line, so that
// tests/test_gpu_cpp_mirror.py can replay the reference's vector loop (src/lib.rs:30-203) through the C++ surface.
//   line:   <function> <arg> <arg> ...      blob args are paths of files holding the raw bytes (or "hex:<hex>"),
//                                           48/32-byte args are hex strings, list args are comma-separated
//   answer: "ok <hex|true|false>[ <hex>]"  |  "err <kind>"  |  "parse"   (an input failed the newtype's own checks)
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include "../../include/kzg355.hpp"

using namespace kzg355;

static std::string hex(const std::vector<uint8_t> &b) {
    static const char *d = "0123456789abcdef"; std::string s;
    for (uint8_t x : b) { s.push_back(d[x >> 4]); s.push_back(d[x & 15]); }
    return s;
}
static Result<Blob> load_blob(const std::string &arg) {
    if (arg.rfind("hex:", 0) == 0) return Blob::from_hex(arg.substr(4));
    std::ifstream f(arg, std::ios::binary);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return Blob::from_bytes(b);
}
static std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out; std::string cur; std::stringstream ss(s);
    while (std::getline(ss, cur, sep)) out.push_back(cur);
    if (s.empty()) out.clear();
    return out;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "usage: cpp_vector_runner <trusted_setup.txt>\n"; return 2; }
    auto rs = Kzg::load_trusted_setup_file(argv[1]);
    if (rs.is_err()) { std::cout << "setup-err " << rs.error().kind << std::endl; return 1; }
    KzgSettings s = rs.value();
    std::cout << "ready" << std::endl;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::stringstream ss(line); std::string fn; ss >> fn;
        std::vector<std::string> a; std::string t; while (ss >> t) a.push_back(t == "-" ? "" : t);
        auto err = [&](const Error &e) { std::cout << "err " << e.kind << std::endl; };
        if (fn == "blob_to_kzg_commitment") {
            auto b = load_blob(a[0]); if (b.is_err()) { std::cout << "parse" << std::endl; continue; }
            auto r = Kzg::blob_to_kzg_commitment(b.value(), s);
            if (r.is_err()) err(r.error()); else std::cout << "ok " << hex(r.value().bytes) << std::endl;
        } else if (fn == "compute_kzg_proof") {
            auto b = load_blob(a[0]); auto z = Bytes32::from_hex(a[1]);
            if (b.is_err() || z.is_err()) { std::cout << "parse" << std::endl; continue; }
            auto r = Kzg::compute_kzg_proof(b.value(), z.value(), s);
            if (r.is_err()) err(r.error()); else std::cout << "ok " << hex(r.value().first.bytes) << " " << hex(r.value().second.bytes) << std::endl;
        } else if (fn == "compute_blob_kzg_proof") {
            auto b = load_blob(a[0]); auto c = KzgCommitment::from_hex(a[1]);
            if (b.is_err() || c.is_err()) { std::cout << "parse" << std::endl; continue; }
            auto r = Kzg::compute_blob_kzg_proof(b.value(), c.value(), s);
            if (r.is_err()) err(r.error()); else std::cout << "ok " << hex(r.value().bytes) << std::endl;
        } else if (fn == "verify_kzg_proof") {
            auto c = KzgCommitment::from_hex(a[0]); auto z = Bytes32::from_hex(a[1]); auto y = Bytes32::from_hex(a[2]); auto p = KzgProof::from_hex(a[3]);
            if (c.is_err() || z.is_err() || y.is_err() || p.is_err()) { std::cout << "parse" << std::endl; continue; }
            auto r = Kzg::verify_kzg_proof(c.value(), z.value(), y.value(), p.value(), s);
            if (r.is_err()) err(r.error()); else std::cout << "ok " << (r.value() ? "true" : "false") << std::endl;
        } else if (fn == "verify_blob_kzg_proof") {
            auto b = load_blob(a[0]); auto c = KzgCommitment::from_hex(a[1]); auto p = KzgProof::from_hex(a[2]);
            if (b.is_err() || c.is_err() || p.is_err()) { std::cout << "parse" << std::endl; continue; }
            auto r = Kzg::verify_blob_kzg_proof(b.value(), c.value(), p.value(), s);
            if (r.is_err()) err(r.error()); else std::cout << "ok " << (r.value() ? "true" : "false") << std::endl;
        } else if (fn == "verify_blob_kzg_proof_batch") {
            std::vector<Blob> bl; std::vector<KzgCommitment> cs; std::vector<KzgProof> ps; bool bad = false;
            for (auto &x : split(a[0], ',')) { auto b = load_blob(x); if (b.is_err()) bad = true; else bl.push_back(b.value()); }
            for (auto &x : split(a[1], ',')) { auto c = KzgCommitment::from_hex(x); if (c.is_err()) bad = true; else cs.push_back(c.value()); }
            for (auto &x : split(a[2], ',')) { auto p = KzgProof::from_hex(x); if (p.is_err()) bad = true; else ps.push_back(p.value()); }
            if (bad) { std::cout << "parse" << std::endl; continue; }
            auto r = Kzg::verify_blob_kzg_proof_batch(bl, cs, ps, s);
            if (r.is_err()) err(r.error()); else std::cout << "ok " << (r.value() ? "true" : "false") << std::endl;
        } else if (fn == "verify_kzg_proof_many" || fn == "compute_kzg_proof_many") {
            // the *_many forms of the single-proof functions: comma-separated lists, one answer per unit: "many t f e ..." / "many <proof>:<y> e ..."
            if (fn == "verify_kzg_proof_many") {
                std::vector<KzgCommitment> cs; std::vector<Bytes32> zs, ys; std::vector<KzgProof> ps; bool bad = false;
                for (auto &x : split(a[0], ',')) { auto c = KzgCommitment::from_hex(x); if (c.is_err()) bad = true; else cs.push_back(c.value()); }
                for (auto &x : split(a[1], ',')) { auto z = Bytes32::from_hex(x); if (z.is_err()) bad = true; else zs.push_back(z.value()); }
                for (auto &x : split(a[2], ',')) { auto y = Bytes32::from_hex(x); if (y.is_err()) bad = true; else ys.push_back(y.value()); }
                for (auto &x : split(a[3], ',')) { auto q = KzgProof::from_hex(x); if (q.is_err()) bad = true; else ps.push_back(q.value()); }
                if (bad) { std::cout << "parse" << std::endl; continue; }
                auto r = Kzg::verify_kzg_proof_many(cs, zs, ys, ps, s);
                if (r.is_err()) { err(r.error()); continue; }
                std::cout << "many";
                for (auto &u : r.value()) std::cout << " " << (u.is_err() ? "e" : u.value() ? "t" : "f");
                std::cout << std::endl;
            } else {
                std::vector<Blob> bl; std::vector<Bytes32> zs; bool bad = false;
                for (auto &x : split(a[0], ',')) { auto b = load_blob(x); if (b.is_err()) bad = true; else bl.push_back(b.value()); }
                for (auto &x : split(a[1], ',')) { auto z = Bytes32::from_hex(x); if (z.is_err()) bad = true; else zs.push_back(z.value()); }
                if (bad) { std::cout << "parse" << std::endl; continue; }
                auto r = Kzg::compute_kzg_proof_many(bl, zs, s);
                if (r.is_err()) { err(r.error()); continue; }
                std::cout << "many";
                for (auto &u : r.value()) { if (u.is_err()) std::cout << " e"; else std::cout << " " << hex(u.value().first.bytes) << ":" << hex(u.value().second.bytes); }
                std::cout << std::endl;
            }
        } else std::cout << "unknown" << std::endl;
    }
    return 0;
}
