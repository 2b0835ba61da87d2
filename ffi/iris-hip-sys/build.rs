// Points rustc at libiris_hip.so.  IRIS_HIP_LIB_DIR = the directory that holds it (iris_amd/lib after
// `python -m iris_amd.build`); the rpath makes `cargo run` find it without LD_LIBRARY_PATH.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("IRIS_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../iris_amd/lib"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=iris_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=IRIS_HIP_LIB_DIR");
}
