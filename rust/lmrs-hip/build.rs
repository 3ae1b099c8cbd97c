// Links liblmrs_hip.so.  LMRS_HIP_LIB_DIR = the directory that holds it (default: ../../lm.rs_amd relative to this crate,
// i.e. the in-tree build made by `python -c "import __graft_entry__ as g; g.build()"`).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("LMRS_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../lm.rs_amd")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=lmrs_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=LMRS_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
