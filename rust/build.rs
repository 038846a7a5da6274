// Link against the in-tree library: PCDHIP_LIB_DIR=/path/to/repo/pcd_amd (the directory that holds libpcdhip.so).
fn main() {
    if let Ok(dir) = std::env::var("PCDHIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=pcdhip");
    println!("cargo:rerun-if-env-changed=PCDHIP_LIB_DIR");
}
