// build.rs — link librtcore.so (raytracer_amd/lib in the MI355X repository) when the `mi355x` feature is enabled.
fn main() {
    if std::env::var_os("CARGO_FEATURE_MI355X").is_some() {
        let dir = std::env::var("RTCORE_LIB_DIR").expect("set RTCORE_LIB_DIR to the directory holding librtcore.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=rtcore");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rerun-if-env-changed=RTCORE_LIB_DIR");
}
