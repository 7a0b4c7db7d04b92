// Links libkzg355.so (built by `make -C kzg_rust_amd/csrc` at the repository root, or installed system-wide).
// KZG355_LIB_DIR overrides the search path; the rpath is embedded so that `cargo test` finds the library without LD_LIBRARY_PATH.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("KZG355_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
        manifest.parent().unwrap().join("kzg_rust_amd")
    });
    println!("cargo:rerun-if-env-changed=KZG355_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/kzg355.h");
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=kzg355");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
