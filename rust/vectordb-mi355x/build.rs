// Links libvdbmi.so (built by `make -C vectordb_amd/csrc`); VDBMI_LIB_DIR overrides the in-tree location.
fn main() {
    let dir = std::env::var("VDBMI_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../../vectordb_amd/csrc", std::env::var("CARGO_MANIFEST_DIR").unwrap())
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=vdbmi");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=VDBMI_LIB_DIR");
}
