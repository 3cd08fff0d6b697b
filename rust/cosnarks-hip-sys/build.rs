// Links libcosnarks_hip.so (built by `python co-snarks_amd/build.py`). COSNARKS_HIP_LIB_DIR overrides the in-tree location.
fn main() {
    let dir = std::env::var("COSNARKS_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../co-snarks_amd/lib").display().to_string()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=cosnarks_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=COSNARKS_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/cosnarks_hip.h");
}
