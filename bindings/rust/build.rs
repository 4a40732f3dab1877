// build.rs of the reference crate when the GPU prover is enabled (not compiled in this repository:
// the image has no Rust toolchain)
fn main() {
    let dir = std::env::var("ZKGPU_LIB_DIR").expect("set ZKGPU_LIB_DIR to the directory holding libzkgpu.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=zkgpu");
    println!("cargo:rustc-link-search=native=/opt/rocm/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
}
