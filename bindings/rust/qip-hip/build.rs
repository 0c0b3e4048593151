// Link against libqip_hip.so.  QIP_HIP_LIB_DIR points at <repo>/rustqip_amd/lib (where
// `python -c "import __graft_entry__ as g; g.build()"` leaves the library); the HIP runtime comes from ROCm.
fn main() {
    let dir = std::env::var("QIP_HIP_LIB_DIR").unwrap_or_else(|_| "../../../rustqip_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=qip_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=QIP_HIP_LIB_DIR");
}
