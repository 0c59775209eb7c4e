// Link against the in-tree library: HIKARI_HIP_LIB_DIR = <repo>/bevy-hikari_amd (where __graft_entry__.build() puts libhikari_hip.so).
fn main() {
    let dir = std::env::var("HIKARI_HIP_LIB_DIR").expect("set HIKARI_HIP_LIB_DIR to the directory holding libhikari_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=hikari_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=HIKARI_HIP_LIB_DIR");
}
