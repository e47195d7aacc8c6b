// links libkyber_ed25519_hip.so built by `python __graft_entry__.py build`
fn main() {
    let dir = std::env::var("KYBER_ED25519_HIP_LIB_DIR").unwrap_or_else(|_| "../../".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=kyber_ed25519_hip");
}
