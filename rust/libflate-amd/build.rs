// Links liblfx.so (built by `python -c "import __graft_entry__ as g; g.build()"` → libflate_amd/liblfx.so).
// LFX_LIB_DIR overrides the search path.
fn main() {
    let dir = std::env::var("LFX_LIB_DIR").unwrap_or_else(|_| "../../libflate_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=lfx");
    println!("cargo:rerun-if-env-changed=LFX_LIB_DIR");
}
