// Points rustc at libiris_hip.so.  IRIS_HIP_LIB_DIR = the directory that holds it (iris_amd/lib after
// `python -m iris_amd.build`); the rpath makes `cargo run` find it without LD_LIBRARY_PATH.
// IRIS_HIP_LIB_NAME = iris_hip (default) or iris_hip_relaxed (the opt-in relaxed-arithmetic build of the same ABI).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("IRIS_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../iris_amd/lib"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    let name = env::var("IRIS_HIP_LIB_NAME").unwrap_or_else(|_| "iris_hip".to_string());
    println!("cargo:rustc-link-lib=dylib={}", name);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=IRIS_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=IRIS_HIP_LIB_NAME");
}
