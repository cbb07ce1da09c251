// SOURCE ONLY (uncompiled here).  Points the linker at libtyplonk_hip.so.
//   TYPLONK_LIB_DIR=/path/to/repo/typlonk_amd cargo build
fn main() {
    let dir = std::env::var("TYPLONK_LIB_DIR").unwrap_or_else(|_| "../../../typlonk_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=TYPLONK_LIB_DIR");
}
