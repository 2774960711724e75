// Points rustc at liberased_cells_hip.so (set EC_HIP_LIB_DIR to the directory that holds it).
fn main() {
    let dir = std::env::var("EC_HIP_LIB_DIR").unwrap_or_else(|_| "../..".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=erased_cells_hip");
    println!("cargo:rerun-if-env-changed=EC_HIP_LIB_DIR");
}
