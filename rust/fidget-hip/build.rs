// Links libfidget_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"` in the fidget-hip repository:
// fidget_amd/csrc/libfidget_hip.so, gfx950 code objects embedded; needs ROCm >= 7's libamdhip64.so at run time).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=FIDGET_HIP_LIB_DIR");
    let dir = env::var("FIDGET_HIP_LIB_DIR")
        .expect("set FIDGET_HIP_LIB_DIR to the directory that holds libfidget_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=fidget_hip");
    // so that `cargo test` finds the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
