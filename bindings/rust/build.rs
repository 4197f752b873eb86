// NEVER COMPILED in the build container - see README.md
fn main() {
    // directory that holds liblasso_hip.so (halo2-lasso_amd/ in the repository)
    let dir = std::env::var("LASSO_HIP_LIB_DIR").unwrap_or_else(|_| "../../halo2-lasso_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=lasso_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=LASSO_HIP_LIB_DIR");
}
