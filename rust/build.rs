// UNTESTED (no Rust toolchain in the build environment).
fn main() {
    let dir = std::env::var("HANAMARU_HIP_DIR").unwrap_or_else(|_| "../hanamaru-renderer_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=hanamaru_hip");
    println!("cargo:rerun-if-env-changed=HANAMARU_HIP_DIR");
}
