// Links libmodppl_hip.so (built by `python -m modppl_amd.build`).  MODPPL_HIP_LIB_DIR points at modppl_amd/csrc.
fn main() {
    let dir = std::env::var("MODPPL_HIP_LIB_DIR").expect("set MODPPL_HIP_LIB_DIR to the directory holding libmodppl_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=modppl_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
